#!/bin/bash
# round 5, lease z: M16 tiles with the conflict-free A-stage swizzle: kernel tests, forced-tile per-shape A/B against the
# 32x32x16 build (variant library), whole frame
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05z; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or igemm or fuzz or tunable or split or batched" > $O/tests.log 2>&1; echo "kernel tests rc=$?"; tail -3 $O/tests.log
VL=$R/otvm_amd/csrc/build/variants/libotvm_m32.so
S="--shape 512,512,3,1,4,136,240 --shape 256,256,3,1,2,136,240 --shape 3072,256,3,1,1,136,240 --shape 2048,512,1,1,1,136,240 --shape 512,2048,1,1,1,136,240 --shape 1024,256,1,1,1,136,240 --shape 256,1024,1,1,1,136,240 --shape 1024,256,1,1,1,60,104 --shape 256,256,3,1,2,60,104"
T="--tune 529,545,561,577,593"
for i in 1 2; do
python tools/conv_bench.py --iters 30 $T $S 2>&1 | grep -v amdgpu | sed 's/$/   (16x16x32)/' | tee -a $O/conv_bench.txt
OTVM_HIP_LIB=$VL python tools/conv_bench.py --iters 30 $T $S 2>&1 | grep -v amdgpu | sed 's/$/   (32x32x16)/' | tee -a $O/conv_bench.txt
done
for v in new m32; do
  lib=""; [ $v = m32 ] && lib=$VL
  OTVM_TUNE_FILE=$O/tune_$v.json OTVM_HIP_LIB=$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  OTVM_TUNE_FILE=$O/tune_$v.json OTVM_HIP_LIB=$lib python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
for i in 1 2 3; do
for v in new m32; do
  lib=""; [ $v = m32 ] && lib=$VL
  OTVM_TUNE_FILE=$O/tune_$v.json OTVM_HIP_LIB=$lib python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p $v', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done; done
for i in 1 2 3; do
for v in new m32; do
  lib=""; [ $v = m32 ] && lib=$VL
  OTVM_TUNE_FILE=$O/tune_$v.json OTVM_HIP_LIB=$lib python bench.py --height 480 --width 832 --steps 97 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p $v', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done; done
