#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
V=otvm_amd/csrc/build/variants
SH="--shape 1024,256,1,1,1,68,120 --shape 256,1024,1,1,1,68,120 --shape 256,256,3,1,1,68,120 --shape 128,128,3,1,1,136,240 --shape 64,256,1,1,1,120,208 --shape 32,64,1,1,1,30,52 --shape 1024,256,1,1,1,30,52"
for v in default abl_NOMFMA abl_NOLOAD abl_NOLDSRD abl_NOSTAGE abl_NOLOAD_NOSTAGE abl_ALL3; do
  if [ $v = default ]; then unset OTVM_HIP_LIB; else export OTVM_HIP_LIB=$PWD/$V/libotvm_$v.so; fi
  echo "## $v"
  timeout 120 python tools/conv_bench.py --tune 81,65 --iters 50 $SH 2>&1 | grep -v amdgpu | awk '{print $3,$5,$6,$9,$10,$11,$12,$13}'
done
