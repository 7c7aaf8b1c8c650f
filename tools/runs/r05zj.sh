#!/bin/bash
# round 5, lease zj: the 256-channel patch tile on v_mfma_f32_16x16x32_f16, five even-tap weight stages per 16 channels
# ({0,3}, {1,4}, {2,5}, {6,7}, {8}); OTVM_PATCH_WIDE_M16 = 1 / 0: kernel tests, the layers alone
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zj; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wide_patch or tunable" > $O/tests.log 2>&1; echo "kernel tests rc=$?"; tail -2 $O/tests.log
for i in 1 2; do for m in 1 0; do
  OTVM_PATCH_WIDE_M16=$m python tools/conv_bench.py --iters 30 --shape 256,256,3,1,1,272,480 --shape 512,256,3,1,1,272,480 --tune 241 2>&1 | grep -v amdgpu | sed "s/^/WIDE_M16=$m  /" | tee -a $O/layers.txt
done; done
