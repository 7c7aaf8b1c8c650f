#!/bin/bash
# same-box A/B: round-3 tree (_old) vs this tree; isolated glue timings; gn_predict test; pred variants
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s > $O/ktests.log 2>&1; python -m pytest tests/test_gpu_frame.py -x -q -m gpu -k "sequence_vs_oracle or frame_fuzz or batched" > $O/ftests.log 2>&1; tail -3 $O/ftests.log
echo "ktests rc $?" >> $O/ktests.log; grep "gn_predict \|block tail" $O/ktests.log | head -40; tail -3 $O/ktests.log
(cd _old && python ../tools/glue_bench.py > $O/glue_old.json 2> $O/glue_old.err); python tools/glue_bench.py > $O/glue_new.json 2> $O/glue_new.err
paste $O/glue_old.json $O/glue_new.json
(cd _old && python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_old.json 2> $O/bench_old.err); head -c 100 $O/bench_old.json; echo
for v in "OTVM_GN_PREDICT=0" "OTVM_GN_PREDICT=1" "OTVM_GN_PREDICT_PASSES=3" "OTVM_GN_PREDICT_DS=0"; do
  env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err
  echo $v; head -c 100 $O/bench_$v.json; echo
done
(cd _old && python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_old2.json 2> $O/bench_old2.err); head -c 100 $O/bench_old2.json; echo
export OTVM_TUNE_FILE=$O/tune.json
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/ks.log 2>&1
cd $R
KS=$(ls $O/ks/*kernel_stats.csv $O/ks/*/*kernel_stats.csv 2>/dev/null | head -1)
python tools/kernel_stats_md.py $KS 23 "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline" > $O/kernel_stats_1080p.md
find $O -name "*kernel_trace.csv" -delete
grep -i "gram\|gn_predict\|gn_apply" $O/kernel_stats_1080p.md
tail -12 $O/kernel_stats_1080p.md
