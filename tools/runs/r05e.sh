#!/bin/bash
# round 5, lease e: the conditioning guard of the predicted GroupNorm statistics (kernel + frame tests), the prediction per
# bottleneck width (which stages pay), frame tests on the LDS-DMA tiles
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s -k "gn_predict" > $O/tests_predict.log 2>&1; echo "predict kernel tests rc=$?" | tee -a $O/tests_predict.log
grep -E "kappa|timing|passed|failed|Error|assert" $O/tests_predict.log | tail -30
timeout 1500 python -m pytest tests/test_gpu_frame.py -x -q -m gpu -s -k "predicted or sequence_vs_oracle" > $O/tests_frame.log 2>&1; echo "frame tests rc=$?" | tee -a $O/tests_frame.log
grep -E "predicted tails|ill-conditioned|interventions|passed|failed|Error|assert|total tie" $O/tests_frame.log | tail -40
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for v in "OTVM_GN_PREDICT_PLANES=64,128,256,512" "OTVM_GN_PREDICT_PLANES=64,256,512" "OTVM_GN_PREDICT_PLANES=256,512" "OTVM_GN_PREDICT_PLANES=512" "OTVM_GN_PREDICT=0" "OTVM_GN_PREDICT_DS=0" "OTVM_GN_PREDICT_PLANES=64,128,256,512"; do
  env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), 'frames/s')" | tee -a $O/ab_gn_predict.txt
done
