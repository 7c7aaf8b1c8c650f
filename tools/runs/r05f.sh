#!/bin/bash
# round 5, lease f: narrow patch tiles (64 / 32 output channels) with LDS-DMA 3-tap weight stages vs the register-staged forms
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or patch or head" > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log; tail -3 $O/tests.log
S="--shape 64,64,3,1,1,1088,1920 --shape 80,64,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 64,32,3,1,1,1088,1920 --shape 80,32,3,1,1,1088,1920 --shape 64,64,3,1,1,272,480 --shape 64,64,3,1,1,480,832"
for v in "OTVM_PATCH64_GLDS=1 OTVM_PATCH32_GLDS=1" "OTVM_PATCH64_GLDS=0 OTVM_PATCH32_GLDS=0"; do
  echo "--- $v" | tee -a $O/patch_glds.txt
  env $v timeout 600 python tools/conv_bench.py --iters 30 --tune 0 $S 2>&1 | grep -v amdgpu | tee -a $O/patch_glds.txt
  env $v timeout 600 python tools/conv_bench.py --iters 30 --tune 0 --res 1 --relu 0 --shape 64,64,3,1,1,1088,1920 2>&1 | grep -v amdgpu | tee -a $O/patch_glds.txt
done
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for v in "OTVM_PATCH64_GLDS=1" "OTVM_PATCH64_GLDS=0" "OTVM_PATCH32_GLDS=0" "OTVM_PATCH64_GLDS=0 OTVM_PATCH32_GLDS=0" "OTVM_PATCH64_GLDS=1"; do
  env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), 'frames/s')" | tee -a $O/ab_patch_glds.txt
done
