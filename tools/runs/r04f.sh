#!/bin/bash
# gram v3 (XCD-aware) / predict v4 timings; glue after the classify / ppm_add fixes; partial-write probe; frame A/B
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s -k "gn_predict or input_norm_with_identity or ppm or trimap" > $O/ktests.log 2>&1
echo "ktests rc $?" >> $O/ktests.log; grep "timing" $O/ktests.log | head -20; tail -3 $O/ktests.log
python tools/glue_bench.py --probe > $O/glue_new.json 2> $O/glue_new.err; cat $O/glue_new.json
python -m pytest tests/test_gpu_frame.py -x -q -m gpu -k "sequence_vs_oracle" > $O/ftests.log 2>&1; tail -3 $O/ftests.log
for v in "OTVM_GN_PREDICT=0" "OTVM_GN_PREDICT=1" "OTVM_GN_PREDICT_DS=0"; do
  env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err
  echo $v; head -c 100 $O/bench_$v.json; echo
done
(cd _old && python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_old.json 2> $O/bench_old.err); head -c 100 $O/bench_old.json; echo
