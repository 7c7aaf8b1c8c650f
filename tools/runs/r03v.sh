#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r03v
O=gpurun_out/r03v
export OTVM_HIP_LIB=$PWD/otvm_amd/csrc/build/variants/libotvm_bnk_timing.so
timeout 300 python tools/bottleneck_bench.py > $O/bnk_1080p_timing.txt 2>&1
timeout 300 python tools/bottleneck_bench.py --height 120 --width 208 > $O/bnk_480p_timing.txt 2>&1
