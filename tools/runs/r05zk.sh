#!/bin/bash
# round 5, lease zk: where does the 1.26 x conv traffic come from?  FETCH_SIZE / WRITE_SIZE per LAYER of the full-resolution patch
# tiles (tools/conv_bench.py, one grid size per layer) in both tile walks of the grid: OTVM_PATCH_ORDER = 0 (row-major) / 1
# (XCD-aware bands, column-major inside a band); then the layers' times and the frame, alternating
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zk; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or patch" > $O/tests0.log 2>&1; echo "patch kernel tests (order 0) rc=$?"; tail -1 $O/tests0.log
OTVM_PATCH_ORDER=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or patch" > $O/tests1.log 2>&1; echo "patch kernel tests (order 1) rc=$?"; tail -1 $O/tests1.log
SH="--shape 80,32,3,1,1,1088,1920 --shape 64,32,3,1,1,1088,1920 --shape 64,64,3,1,1,1088,1920 --shape 80,64,3,1,1,1088,1920 --shape 256,256,3,1,1,272,480 --shape 512,256,3,1,1,272,480 --shape 320,64,3,1,1,544,960"
for i in 1 2; do for m in 0 1; do
  OTVM_PATCH_ORDER=$m python tools/conv_bench.py --iters 30 $SH --tune 241 2>&1 | grep -v amdgpu | sed "s/^/ORDER=$m  /" | tee -a $O/layers.txt
done; done
for b in 4 16; do OTVM_PATCH_ORDER=1 OTVM_PATCH_BAND=$b python tools/conv_bench.py --iters 30 $SH --tune 241 2>&1 | grep -v amdgpu | sed "s/^/ORDER=1 BAND=$b  /" | tee -a $O/layers.txt; done
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/conv_bench.py --iters 3 $SH --tune 241"
for m in 0 1; do
  OTVM_PATCH_ORDER=$m rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch$m -o f -- $CMD > $O/fetch$m.log 2>&1
  OTVM_PATCH_ORDER=$m rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write$m -o w -- $CMD > $O/write$m.log 2>&1
  echo "== OTVM_PATCH_ORDER=$m" | tee -a $O/layer_traffic.md
  python $R/tools/pmc_layer_traffic.py $O/fetch$m $O/write$m conv_patch | tee -a $O/layer_traffic.md
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
cd $R
for i in 1 2; do for m in 0 1; do
  OTVM_PATCH_ORDER=$m python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ORDER=$m 1080p', d['value'], d['ms_per_step'])" | tee -a $O/frame.txt
done; done
