#!/bin/bash
# round 5, lease x: the 16x16x32 timing probe (OTVM_PABL_MFMA16: two v_mfma_f32_16x16x32_f16 per 32x32x16 MFMA, same fragments;
# WRONG results) in the patch kernels against the shipped ones
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05x; mkdir -p $O
cd $R
VL=$R/otvm_amd/csrc/build/variants/libotvm_pm16.so
S="--shape 64,64,3,1,1,1088,1920 --shape 80,64,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 80,32,3,1,1,1088,1920 --shape 256,256,3,1,1,272,480 --shape 512,256,3,1,1,272,480 --shape 2048,256,3,1,1,136,240"
for i in 1 2; do
python tools/conv_bench.py --iters 30 $S 2>&1 | grep -v amdgpu | sed 's/$/   (32x32x16)/' | tee -a $O/conv_bench.txt
OTVM_HIP_LIB=$VL python tools/conv_bench.py --iters 30 $S 2>&1 | grep -v amdgpu | sed 's/$/   (2 x 16x16x32 probe)/' | tee -a $O/conv_bench.txt
done
