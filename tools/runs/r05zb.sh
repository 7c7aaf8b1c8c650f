#!/bin/bash
# round 5, lease zb: the nine-tap 64-filter patch tiles on v_mfma_f32_16x16x32_f16 (OTVM_PATCH_M16 = 1 default / 0): kernel tests,
# the layers alone on the device, then the whole frame by switch, alternating
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zb; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv2d or fused_groupnorm or fuzz or tunable or input_groupnorm or batched_launch or table_of_its_output" > $O/tests.log 2>&1; echo "kernel tests rc=$?"; tail -5 $O/tests.log
for m in 1 0 1 0; do
  echo "OTVM_PATCH_M16=$m" | tee -a $O/layers.txt
  OTVM_PATCH_M16=$m python tools/conv_bench.py --iters 30 --shape 64,64,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 64,64,3,1,1,480,832 --shape 128,64,3,1,2,272,480 2>&1 | grep -v amdgpu | tee -a $O/layers.txt
done
for m in 1 0; do
  OTVM_PATCH_M16=$m OTVM_TUNE_FILE=$O/tune_$m.json python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
for i in 1 2 3; do for m in 1 0; do
  OTVM_PATCH_M16=$m OTVM_TUNE_FILE=$O/tune_$m.json python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p OTVM_PATCH_M16=$m', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done; done
for i in 1 2; do for m in 1 0; do
  OTVM_PATCH_M16=$m python bench.py --height 480 --width 832 --steps 97 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p OTVM_PATCH_M16=$m', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done; done
