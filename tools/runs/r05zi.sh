#!/bin/bash
# round 5, lease zi: the 256-channel patch tile on v_mfma_f32_16x16x32_f16 with even-tap weight stages ({0,3,1,4}, {2,5,6,7}, {8};
# OTVM_PATCH_WIDE_M16 = 1 default / 0): kernel tests, the layers alone, the whole frame by switch, alternating
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zi; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wide_patch or tunable or conv2d or fused_groupnorm or batched_launch or fuzz_all or input_groupnorm or input_norm" > $O/tests.log 2>&1; echo "kernel tests rc=$?"; tail -4 $O/tests.log
for i in 1 2; do for m in 1 0; do
  OTVM_PATCH_WIDE_M16=$m python tools/conv_bench.py --iters 30 --shape 256,256,3,1,1,272,480 --shape 512,256,3,1,1,272,480 --shape 2048,256,3,1,1,136,240 --tune 0,241 2>&1 | grep -v amdgpu | sed "s/^/WIDE_M16=$m  /" | tee -a $O/layers.txt
done; done
for m in 1 0; do OTVM_PATCH_WIDE_M16=$m OTVM_TUNE_FILE=$O/tune_$m.json python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1; done
for i in 1 2 3; do for m in 1 0; do
  OTVM_PATCH_WIDE_M16=$m OTVM_TUNE_FILE=$O/tune_$m.json python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p OTVM_PATCH_WIDE_M16=$m', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done; done
