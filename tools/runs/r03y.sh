#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r03y
O=gpurun_out/r03y
SH="--shape 256,256,3,1,1,272,480 --shape 512,512,3,1,2,136,240 --shape 2048,256,3,1,1,136,240 --shape 1024,256,1,1,1,136,240 --shape 256,1024,1,1,1,136,240 --shape 512,2048,1,1,1,136,240 --shape 256,64,1,1,1,272,480 --shape 64,256,1,1,1,272,480 --shape 1024,512,3,1,1,68,120 --shape 128,128,3,1,1,136,240"
for v in head new head new; do
  if [ $v = new ]; then unset OTVM_HIP_LIB; else export OTVM_HIP_LIB=$PWD/otvm_amd/csrc/build/variants/libotvm_head.so; fi
  echo "## $v" >> $O/conv_ab.txt
  timeout 300 python tools/conv_bench.py --iters 30 $SH 2>&1 | grep -v amdgpu >> $O/conv_ab.txt
  timeout 300 python tools/conv_bench.py --iters 30 --res 1 --relu 1 --shape 64,256,1,1,1,272,480 --shape 256,1024,1,1,1,136,240 2>&1 | grep -v amdgpu >> $O/conv_ab.txt
  timeout 300 python tools/conv_bench.py --iters 30 --gn 1 --shape 256,256,3,1,1,272,480 --shape 2048,256,3,1,1,136,240 2>&1 | grep -v amdgpu >> $O/conv_ab.txt
done
unset OTVM_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu > $O/pytest_kernels.log 2>&1; echo "rc $?" >> $O/pytest_kernels.log
