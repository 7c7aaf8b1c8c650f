#!/bin/bash
# lease r06d: the planes-128 fused STM bottleneck inside the frame -- frame parity tests, then 1080p / 480p with the block timed
# at plan time (default), always fused (2) and never (0), alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06d; O=gpurun_out/r06d
timeout 1500 python -m pytest tests/test_gpu_frame.py -q -x -m gpu -k "sequence_vs_oracle" > $O/t_frame.log 2>&1; echo "frame rc $?" >> $O/t_frame.log
tail -3 $O/t_frame.log
for rep in 1 2; do for v in 1 0 2; do
  OTVM_FUSE_STM_BLOCK128=$v OTVM_BENCH_LIVE_PMC=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --tune-report $O/tune_1080_$v.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p OTVM_FUSE_STM_BLOCK128=$v', round(d['value'],2), 'frames/s')"
done; done 2>&1 | tee $O/ab_1080p.txt
for rep in 1 2; do for v in 1 0 2; do
  OTVM_FUSE_STM_BLOCK128=$v OTVM_BENCH_LIVE_PMC=0 python bench.py --height 480 --width 832 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --tune-report $O/tune_480_$v.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p OTVM_FUSE_STM_BLOCK128=$v', round(d['value'],2), 'frames/s')"
done; done 2>&1 | tee $O/ab_480p.txt
python - <<'PY'
import json
for f in ("gpurun_out/r06d/tune_1080_1.json", "gpurun_out/r06d/tune_480_1.json"):
    for r in json.load(open(f)):
        if "fused bottleneck" in r["layer"]:
            print(f[-18:], r["layer"][:40], r["shape"]["H"], r["shape"]["W"], "chosen", r["chosen"], r["ms"])
PY
