#!/bin/bash
export TMPDIR=/tmp
V=otvm_amd/csrc/build/variants
SH="--shape 64,64,3,1,1,1088,1920 --shape 80,64,3,1,1,1088,1920 --shape 64,32,3,1,1,1088,1920 --shape 32,16,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960"
for v in default pabl_NOMFMA pabl_NOLOAD pabl_NOCOMMIT pabl_NOLDSRD pabl_NOEPI pabl_LOADCOMMIT pabl_NOMFMA_NOEPI pabl_ONLYMFMA; do
  if [ $v = default ]; then unset OTVM_HIP_LIB; else export OTVM_HIP_LIB=$PWD/$V/libotvm_$v.so; fi
  echo "## $v"
  timeout 120 python tools/conv_bench.py --tune 241 --iters 20 $SH 2>&1 | grep -v amdgpu | awk '{print $2,$4,$8,$9,$11,$12,$13}'
  timeout 120 python tools/conv_bench.py --tune 241 --iters 20 --gn 1 --shape 64,64,3,1,1,1088,1920 2>&1 | grep -v amdgpu | awk '{print "gn:",$2,$4,$8,$9,$11,$12,$13}'
  timeout 120 python tools/conv_bench.py --tune 241 --iters 20 --res 1 --shape 64,64,3,1,1,1088,1920 2>&1 | grep -v amdgpu | awk '{print "res:",$2,$4,$8,$9,$11,$12,$13}'
done
