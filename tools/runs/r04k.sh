#!/bin/bash
# 16-wide head kernel: tests + A/B
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04k; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "head" > $O/ktests.log 2>&1
echo "ktests rc $?" >> $O/ktests.log; tail -5 $O/ktests.log
python -m pytest tests/test_gpu_frame.py tests/test_gpu_train.py -x -q -m gpu -k "sequence_vs_oracle or train or batched" > $O/ftests.log 2>&1; tail -3 $O/ftests.log
export OTVM_TUNE_FILE=$O/tune.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for v in "OTVM_HEAD16=1" "OTVM_HEAD16=0" "OTVM_HEAD16=1" "OTVM_HEAD16=0"; do
  env $v python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err
  echo $v; head -c 100 $O/bench_$v.json; echo
done
R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/ks.log 2>&1
cd $R
KS=$(ls $O/ks/*kernel_stats.csv $O/ks/*/*kernel_stats.csv 2>/dev/null | head -1)
python tools/kernel_stats_md.py $KS 23 "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline" > $O/kernel_stats_1080p.md
find $O -name "*kernel_trace.csv" -delete
grep -i "head16\|patch_f16x3_kernel<8, 32" $O/kernel_stats_1080p.md
