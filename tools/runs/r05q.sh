#!/bin/bash
# round 5, lease q: gn_apply with its loads requested ahead of the statistics (one memory round trip instead of three per launch):
# GroupNorm / frame tests, then whole-frame A/B against the previous library at 832x480 and 1080p
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05q; mkdir -p $O
cd $R
PREV=$R/otvm_amd/csrc/build/variants/libotvm_prev.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frame.py -x -q -m gpu -k "gn or groupnorm or norm or sequence_vs_oracle or head" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for i in 1 2 3; do
for v in new prev; do
  lib=""; [ $v = prev ] && lib=$PREV
  OTVM_HIP_LIB=$lib python bench.py --height 480 --width 832 --steps 97 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p  $v', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done; done
for i in 1 2 3; do
for v in new prev; do
  lib=""; [ $v = prev ] && lib=$PREV
  OTVM_HIP_LIB=$lib python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p $v', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt
done; done
