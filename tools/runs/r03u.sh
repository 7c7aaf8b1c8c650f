#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r03u
O=gpurun_out/r03u
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "stm_bottleneck" > $O/pytest_kernel.log 2>&1; echo "kernel rc $?" >> $O/pytest_kernel.log
timeout 300 python tools/bottleneck_bench.py > $O/bnk_1080p.txt 2>&1
timeout 300 python tools/bottleneck_bench.py --height 120 --width 208 > $O/bnk_480p.txt 2>&1
