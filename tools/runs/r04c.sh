#!/bin/bash
# glue rewrite + predicted GroupNorm statistics: kernel tests, frame tests, A/B bench, kernel stats
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s > $O/ktests.log 2>&1
echo "ktests rc $?" >> $O/ktests.log; grep "gn_predict\|block tail" $O/ktests.log | head -40; tail -4 $O/ktests.log
python -m pytest tests/test_gpu_frame.py -x -q -m gpu -k "sequence_vs_oracle or frame_fuzz or batched" > $O/ftests.log 2>&1
echo "ftests rc $?" >> $O/ftests.log; tail -4 $O/ftests.log
for v in 0 1; do
  OTVM_GN_PREDICT=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_pred$v.json 2> $O/bench_pred$v.err
  head -c 120 $O/bench_pred$v.json; echo
done
OTVM_GN_PREDICT_DS=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_pred1_nods.json 2> $O/bench_pred1_nods.err
head -c 120 $O/bench_pred1_nods.json; echo
OTVM_GN_PREDICT_PASSES=3 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_pred1_p3.json 2> $O/bench_pred1_p3.err
head -c 120 $O/bench_pred1_p3.json; echo
python bench.py --steps 40 --warmup 5 > $O/bench.json 2> $O/bench.err
R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/ks -o ks -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/$O/ks.log 2>&1
cd $R
KS=$(ls $O/ks/*kernel_stats.csv $O/ks/*/*kernel_stats.csv 2>/dev/null | head -1)
python tools/kernel_stats_md.py $KS 23 "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline" > $O/kernel_stats_1080p.md
find $O -name "*kernel_trace.csv" -delete
grep -i "ppm\|edt\|classify\|preprocess\|fba_head\|up4soft\|gram\|gn_predict\|gn_apply" $O/kernel_stats_1080p.md
tail -12 $O/kernel_stats_1080p.md
