#!/bin/bash
# predict v5 timings; kernel stats of the frame with and without the prediction (tuned configurations from a file)
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s -k "gn_predict" > $O/ktests.log 2>&1
echo "ktests rc $?" >> $O/ktests.log; grep "timing" $O/ktests.log | head -20; tail -3 $O/ktests.log
R=$PWD
for v in 0 1; do
  export OTVM_GN_PREDICT=$v OTVM_TUNE_FILE=$O/tune$v.json
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_pred$v.json 2> $O/bench_pred$v.err; head -c 100 $O/bench_pred$v.json; echo
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$v -o ks -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/ks$v.log 2>&1
  cd $R
  KS=$(ls $O/ks$v/*kernel_stats.csv $O/ks$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  python tools/kernel_stats_md.py $KS 23 "OTVM_GN_PREDICT=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline" > $O/kernel_stats_pred$v.md
done
find $O -name "*kernel_trace.csv" -delete
head -60 $O/kernel_stats_pred1.md
