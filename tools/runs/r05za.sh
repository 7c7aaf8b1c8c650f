#!/bin/bash
# round 5, lease za: the 16x16x32 form of the LDS-DMA tiles as a tile family of its own (64 + t), offered to the plan-time tuner
# next to the 32x32x16 form: kernel tests, then whole-frame A/B by switch (OTVM_IGEMM_M16 = 1 default / 0 never / 2 always)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05za; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or igemm or fuzz or tunable or split or batched" > $O/tests.log 2>&1; echo "kernel tests rc=$?"; tail -3 $O/tests.log
for m in 1 0 2; do
  OTVM_IGEMM_M16=$m OTVM_TUNE_FILE=$O/tune_$m.json python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  OTVM_IGEMM_M16=$m OTVM_TUNE_FILE=$O/tune_$m.json python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
for i in 1 2 3; do for m in 1 0 2; do
  OTVM_IGEMM_M16=$m OTVM_TUNE_FILE=$O/tune_$m.json python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p OTVM_IGEMM_M16=$m', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done; done
for i in 1 2 3; do for m in 1 0 2; do
  OTVM_IGEMM_M16=$m OTVM_TUNE_FILE=$O/tune_$m.json python bench.py --height 480 --width 832 --steps 97 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p OTVM_IGEMM_M16=$m', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done; done
python - <<'PY'
import json, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r05za"
t = json.load(open(O + "/tune_1.json"))
c = collections.Counter()
for k, v in (t.get("cache") or t).items() if isinstance(t, dict) else []:
    code = v if isinstance(v, int) else (v[0] if isinstance(v, list) else 0)
    tile = code // 16 - 1
    c["heuristic" if code == 0 else ("m16" if tile >= 64 else ("glds" if tile >= 32 else "other"))] += 1
print("tuner choices with both forms offered:", dict(c))
PY
