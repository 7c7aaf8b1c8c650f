#!/bin/bash
# round 5, lease w: v_mfma_f32_16x16x32_f16 sustains ~10 % more than v_mfma_f32_32x32x16_f16 under the power limit in the pure-MFMA
# probe (tools/probes/mfma_variants_probe.hip).  Does it pay inside the implicit-GEMM tiles?  Timing probe OTVM_ABL_MFMA16 (every
# 32x32x16 MFMA replaced by two 16x16x32 on the same fragments: same FLOPs, same traffic, WRONG results) against the shipped tiles
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05w; mkdir -p $O
cd $R
tools/probes/mfma_variants_probe | tee $O/mfma_variants.txt
VL=$R/otvm_amd/csrc/build/variants/libotvm_m16.so
S="--shape 256,256,3,1,1,272,480 --shape 512,512,3,1,4,136,240 --shape 3072,256,3,1,1,136,240 --shape 2048,512,1,1,1,136,240 --shape 512,2048,1,1,1,136,240 --shape 1024,256,1,1,1,136,240"
for i in 1 2; do
python tools/conv_bench.py --iters 30 --tune 529,545 $S 2>&1 | grep -v amdgpu | sed 's/$/   (32x32x16)/' | tee -a $O/conv_bench.txt
OTVM_HIP_LIB=$VL python tools/conv_bench.py --iters 30 --tune 529,545 $S 2>&1 | grep -v amdgpu | sed 's/$/   (2 x 16x16x32 probe)/' | tee -a $O/conv_bench.txt
done
