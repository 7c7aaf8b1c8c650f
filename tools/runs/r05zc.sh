#!/bin/bash
# round 5, lease zc: M16 patch tiles with kx-pairs (shared A fragments; default) vs taps paired (t, t + 1) (variant build
# OTVM_PM16_NOKXP) vs the 32x32x16 form (OTVM_PATCH_M16=0): kernel tests, the layers alone, the whole frame, alternating
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zc; mkdir -p $O
cd $R
V=$R/otvm_amd/csrc/build/variants/libotvm_nokxp.so
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $O/tests.log 2>&1; echo "kernel tests rc=$?"; tail -3 $O/tests.log
run() { # name, env...
  local name=$1; shift
  env "$@" python tools/conv_bench.py --iters 30 --shape 64,64,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 64,64,3,1,1,480,832 2>&1 | grep -v amdgpu | sed "s/^/$name  /" | tee -a $O/layers.txt
}
for i in 1 2; do run kxp A=1; run tt1 OTVM_HIP_LIB=$V; run m32 OTVM_PATCH_M16=0; done
for i in 1 2 3; do
  for cfg in "kxp A=1" "tt1 OTVM_HIP_LIB=$V" "m32 OTVM_PATCH_M16=0"; do
    set -- $cfg; name=$1; shift
    env "$@" OTVM_TUNE_FILE=$O/tune_$name.json python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p $name', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
  done
done
for i in 1 2; do
  for cfg in "kxp A=1" "m32 OTVM_PATCH_M16=0"; do
    set -- $cfg; name=$1; shift
    env "$@" python bench.py --height 480 --width 832 --steps 97 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p $name', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
  done
done
