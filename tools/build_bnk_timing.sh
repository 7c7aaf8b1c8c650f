#!/bin/bash
# experiment build of the fused bottleneck kernel with per-stage timers (tools/bottleneck_bench.py prints them):
#   OTVM_HIP_LIB=$PWD/otvm_amd/csrc/build/variants/libotvm_bnk_timing.so python tools/bottleneck_bench.py
set -e
cd "$(dirname "$0")/../otvm_amd/csrc"
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -DOTVM_BNK_TIMING -c bottleneck_f16x3.hip -o build/variants/bnk_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libotvm_bnk_timing.so $(ls build/*.o | grep -v bottleneck) build/variants/bnk_timing.o
echo build/variants/libotvm_bnk_timing.so
