#!/bin/bash
# round 5, lease zh: the 16x16x32 timing probe (-DOTVM_PABL_MFMA16=1: two v_mfma_f32_16x16x32_f16 per 32x32x16 MFMA on the same
# fragments; WRONG results) in the patch tiles that still multiply with the 32x32x16 instruction: the 256-channel tile and the
# 32-filter tiles -- what a real 16x16x32 form could buy at most, before building one
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zh; mkdir -p $O
cd $R
VL=$R/otvm_amd/csrc/build/variants/libotvm_pm16.so
S="--shape 256,256,3,1,1,272,480 --shape 512,256,3,1,1,272,480 --shape 2048,256,3,1,1,136,240 --shape 80,32,3,1,1,1088,1920 --shape 64,32,3,1,1,1088,1920 --shape 64,32,3,1,1,480,832"
for i in 1 2 3; do
python tools/conv_bench.py --iters 30 $S 2>&1 | grep -v amdgpu | sed 's/$/   (32x32x16)/' | tee -a $O/conv_bench.txt
OTVM_HIP_LIB=$VL python tools/conv_bench.py --iters 30 $S 2>&1 | grep -v amdgpu | sed 's/$/   (2 x 16x16x32 probe)/' | tee -a $O/conv_bench.txt
done
