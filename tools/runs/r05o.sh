#!/bin/bash
# round 5, lease o: the patch kernels' staging without per-element branches (input activation as a slope, the staging forms
# behind one uniform branch per stage) -- variant library against the shipped one: kernel tests, conv_bench, whole frame
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05o; mkdir -p $O
cd $R
VL=$R/otvm_amd/csrc/build/variants/libotvm_slope.so
OTVM_HIP_LIB=$VL timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or patch or fuzz" > $O/tests.log 2>&1; echo "tests (variant) rc=$?"; tail -3 $O/tests.log
S="--shape 64,64,3,1,1,1088,1920 --shape 80,64,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 80,32,3,1,1,1088,1920 --shape 256,256,3,1,1,272,480"
for i in 1 2; do
python tools/conv_bench.py --iters 30 $S 2>&1 | grep -v amdgpu | sed 's/$/   (shipped)/' | tee -a $O/conv_bench.txt
OTVM_HIP_LIB=$VL python tools/conv_bench.py --iters 30 $S 2>&1 | grep -v amdgpu | sed 's/$/   (slope)/' | tee -a $O/conv_bench.txt
done
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for i in 1 2 3; do
python bench.py --steps 60 --warmup 5 --no-cpu-baseline --layer-report $O/layers_shipped.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shipped', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
OTVM_HIP_LIB=$VL python bench.py --steps 60 --warmup 5 --no-cpu-baseline --layer-report $O/layers_slope.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slope  ', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done
python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05o"
a = {r["layer"]: r for r in json.load(open(O + "/layers_shipped.json"))}
b = {r["layer"]: r for r in json.load(open(O + "/layers_slope.json"))}
rows = sorted(a, key=lambda k: -a[k]["ms_per_frame"])[:40]
for k in rows:
    if k in b and ("refine" in k or "conv_up" in k or "pred" in k or "layer1" in k):
        print("%-44s %.3f -> %.3f ms" % (k, a[k]["ms_per_frame"], b[k]["ms_per_frame"]))
PY
