#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r03z
O=gpurun_out/r03z
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frame.py -q -x -m gpu > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
export OTVM_TUNE_FILE=/tmp/tune_r03z.json
for rep in 1 2; do
for v in new head; do
  if [ $v = new ]; then unset OTVM_HIP_LIB; else export OTVM_HIP_LIB=$PWD/otvm_amd/csrc/build/variants/libotvm_head.so; fi
  timeout 900 python bench.py --steps 97 --warmup 3 --no-cpu-baseline > $O/bench_1080p_${v}_$rep.json 2> $O/bench_1080p_${v}_$rep.err
  timeout 600 python bench.py --height 480 --width 832 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_480p_${v}_$rep.json 2> $O/bench_480p_${v}_$rep.err
done
done
unset OTVM_HIP_LIB
SH="--shape 64,64,3,1,1,1088,1920 --shape 80,64,3,1,1,1088,1920 --shape 64,32,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 256,256,3,1,2,136,240 --shape 512,512,3,1,4,136,240"
for v in head new head new; do
  if [ $v = new ]; then unset OTVM_HIP_LIB; else export OTVM_HIP_LIB=$PWD/otvm_amd/csrc/build/variants/libotvm_head.so; fi
  echo "## $v" >> $O/patch_ab.txt
  timeout 300 python tools/conv_bench.py --iters 20 $SH 2>&1 | grep -v amdgpu >> $O/patch_ab.txt
  timeout 300 python tools/conv_bench.py --iters 20 --gn 1 --shape 64,64,3,1,1,1088,1920 2>&1 | grep -v amdgpu >> $O/patch_ab.txt
  timeout 300 python tools/conv_bench.py --iters 20 --res 1 --shape 64,64,3,1,1,1088,1920 2>&1 | grep -v amdgpu >> $O/patch_ab.txt
  timeout 300 python tools/conv_bench.py --iters 20 --shape 24,64,7,2,1,1088,1920 2>&1 | grep -v amdgpu >> $O/patch_ab.txt
done
