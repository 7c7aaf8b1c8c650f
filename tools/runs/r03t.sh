#!/bin/bash
# fused STM bottleneck: kernel test, micro-benchmark, frame tests, A/B bench
cd /root/repo
mkdir -p gpurun_out/r03t
O=gpurun_out/r03t
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "stm_bottleneck" > $O/pytest_kernel.log 2>&1; echo "kernel rc $?" >> $O/pytest_kernel.log
timeout 300 python tools/bottleneck_bench.py > $O/bnk_1080p.txt 2>&1
timeout 300 python tools/bottleneck_bench.py --height 120 --width 208 > $O/bnk_480p.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_frame.py -q -x -m gpu > $O/pytest_frame.log 2>&1; echo "frame rc $?" >> $O/pytest_frame.log
export OTVM_TUNE_FILE=/tmp/tune_r03t.json
for f in 1 0; do
  OTVM_FUSE_STM_BLOCK=$f timeout 900 python bench.py --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_1080p_fuse$f.json 2> $O/bench_1080p_fuse$f.err
  OTVM_FUSE_STM_BLOCK=$f timeout 600 python bench.py --height 480 --width 832 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_480p_fuse$f.json 2> $O/bench_480p_fuse$f.err
done
