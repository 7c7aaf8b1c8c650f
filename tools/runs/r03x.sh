#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r03x
O=gpurun_out/r03x
export OTVM_TUNE_FILE=/tmp/tune_r03x.json
for rep in 1 2; do
for f in 1 0; do
  OTVM_FUSE_STM_BLOCK=$f timeout 900 python bench.py --steps 97 --warmup 3 --no-cpu-baseline > $O/bench_1080p_fuse${f}_$rep.json 2> $O/bench_1080p_fuse${f}_$rep.err
  OTVM_FUSE_STM_BLOCK=$f timeout 600 python bench.py --height 480 --width 832 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_480p_fuse${f}_$rep.json 2> $O/bench_480p_fuse${f}_$rep.err
done
done
timeout 1500 python -m pytest tests -q -m gpu --durations=6 > $O/pytest_gpu.log 2>&1; echo "rc $?" >> $O/pytest_gpu.log
