#!/bin/bash
# round 5, lease s: Gram chunking -- 256-wide blocks for >= 512 channels, up to 512 chunks of >= 128 pixels for the one-block
# 64-channel tensors -- against the previous library: the timing lines of test_gn_predict_matches_accumulated_statistics, a sweep
# of the 256-wide block's workgroup target, GroupNorm-prediction tests, whole-frame A/B
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05s; mkdir -p $O
cd $R
PREV=$R/otvm_amd/csrc/build/variants/libotvm_prev.so
T="python -m pytest tests/test_gpu_kernels.py -q -s -m gpu -k test_gn_predict_matches_accumulated_statistics"
echo "--- previous library" | tee -a $O/gram_timing.txt; OTVM_HIP_LIB=$PREV $T 2>&1 | grep -E "timing|passed|failed" | tee -a $O/gram_timing.txt
echo "--- new defaults" | tee -a $O/gram_timing.txt; $T 2>&1 | grep -E "timing|passed|failed" | tee -a $O/gram_timing.txt
for w in 96 144 288; do echo "--- new, OTVM_GRAM_WGS256=$w" | tee -a $O/gram_timing.txt; OTVM_GRAM_WGS256=$w $T 2>&1 | grep -E "timing|passed|failed" | grep -E "block 256|passed|failed" | tee -a $O/gram_timing.txt; done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frame.py tests/test_gpu_fullsize.py -x -q -m gpu -k "gn_predict or predicted or conditioning or 1080p_two_frames or 1080p_steady" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for i in 1 2 3; do
for v in new prev; do
  lib=""; [ $v = prev ] && lib=$PREV
  OTVM_HIP_LIB=$lib python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p $v', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done; done
