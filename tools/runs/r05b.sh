#!/bin/bash
# round 5, lease b: per-shape A/B of the register-staged (17 / 33) and LDS-DMA (257 / 273) forms of the 256x256 / 256x128 tiles,
# whole-frame bench with the LDS-DMA tiles as heuristic default + tuner candidates vs OTVM_IGEMM_GLDS=0 vs the round-4 tree
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "batched_launch or race_free or every_tunable" > $O/tests_conv.log 2>&1; echo "conv tests rc=$?" | tee -a $O/tests_conv.log
tail -3 $O/tests_conv.log
S="--shape 256,256,3,1,1,272,480 --shape 3072,256,3,1,1,136,240 --shape 512,512,3,1,4,136,240 --shape 1024,256,1,1,1,136,240 --shape 256,1024,1,1,1,136,240 --shape 2048,512,1,1,1,136,240 --shape 512,2048,1,1,1,136,240 --shape 64,256,1,1,1,272,480 --shape 512,128,1,1,1,272,480 --shape 128,512,1,1,1,272,480 --shape 1024,256,3,1,1,136,240"
timeout 600 python tools/conv_bench.py --iters 30 --tune 17,257,33,273 $S > $O/conv_bench_glds.txt 2>&1
cat $O/conv_bench_glds.txt
timeout 600 python tools/conv_bench.py --iters 30 --tune 17,257,33,273 --res 1 --shape 256,1024,1,1,1,136,240 --shape 512,2048,1,1,1,136,240 --shape 128,512,1,1,1,272,480 > $O/conv_bench_glds_res.txt 2>&1
cat $O/conv_bench_glds_res.txt
for i in 1 2; do
timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --tune-report $O/tune_1080p.json > $O/bench_new.json 2> $O/bench_new.err; head -c 200 $O/bench_new.json; echo
OTVM_IGEMM_GLDS=0 timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | head -c 200; echo
(cd _old && timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | head -c 200); echo
done
python - <<'PY'
import json,sys
d=json.load(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r05b/tune_1080p.json"))
rows=d if isinstance(d,list) else d.get("layers",d)
print(str(rows)[:3000])
PY
