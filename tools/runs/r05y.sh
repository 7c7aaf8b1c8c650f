#!/bin/bash
# round 5, lease y: the LDS-DMA implicit-GEMM tiles on v_mfma_f32_16x16x32_f16 (M16) -- kernel tests (every configuration, race
# screens, fuzz), then per-shape and whole-frame A/B against the same tree built with -DOTVM_IGEMM_M16=0 (32x32x16)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05y; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or igemm or fuzz or tunable or split or batched" > $O/tests.log 2>&1; echo "kernel tests rc=$?"; tail -4 $O/tests.log
timeout 300 python tools/conv_fuzz.py --n 300 --seed 71 2>&1 | grep -v amdgpu | tail -1
VL=$R/otvm_amd/csrc/build/variants/libotvm_m32.so
S="--shape 256,256,3,1,1,272,480 --shape 512,512,3,1,4,136,240 --shape 256,256,3,1,2,136,240 --shape 3072,256,3,1,1,136,240 --shape 2048,512,1,1,1,136,240 --shape 512,2048,1,1,1,136,240 --shape 1024,256,1,1,1,136,240 --shape 256,1024,1,1,1,136,240 --shape 1024,256,1,1,1,60,104 --shape 256,256,3,1,2,60,104"
for i in 1 2; do
python tools/conv_bench.py --iters 30 $S 2>&1 | grep -v amdgpu | sed 's/$/   (16x16x32)/' | tee -a $O/conv_bench.txt
OTVM_HIP_LIB=$VL python tools/conv_bench.py --iters 30 $S 2>&1 | grep -v amdgpu | sed 's/$/   (32x32x16)/' | tee -a $O/conv_bench.txt
done
for v in new m32; do
  lib=""; [ $v = m32 ] && lib=$VL
  OTVM_TUNE_FILE=$O/tune_$v.json OTVM_HIP_LIB=$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
for i in 1 2 3; do
for v in new m32; do
  lib=""; [ $v = m32 ] && lib=$VL
  OTVM_TUNE_FILE=$O/tune_$v.json OTVM_HIP_LIB=$lib python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p $v', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done; done
for i in 1 2; do
for v in new m32; do
  lib=""; [ $v = m32 ] && lib=$VL
  OTVM_HIP_LIB=$lib python bench.py --height 480 --width 832 --steps 97 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p $v', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done; done
