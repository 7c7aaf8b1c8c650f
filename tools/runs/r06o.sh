#!/bin/bash
# lease r06o: which pixel tile of the fused res3 block serves the FRAME best (the plan-time choice is made on isolated launch times)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06o; O=gpurun_out/r06o
for rep in 1 2; do for v in "1 0" "2 1" "2 2" "2 3" "0 0"; do set -- $v
  OTVM_FUSE_STM_BLOCK128=$1 OTVM_STM128_TILE=$2 OTVM_BENCH_LIVE_PMC=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p FUSE_STM_BLOCK128=$1 TILE=$2', round(d['value'],2), 'frames/s')"
done; done | tee $O/ab_1080p.txt
for rep in 1 2; do for v in "1 0" "2 2" "2 3" "0 0"; do set -- $v
  OTVM_FUSE_STM_BLOCK128=$1 OTVM_STM128_TILE=$2 OTVM_BENCH_LIVE_PMC=0 python bench.py --height 480 --width 832 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p FUSE_STM_BLOCK128=$1 TILE=$2', round(d['value'],2), 'frames/s')"
done; done | tee $O/ab_480p.txt
