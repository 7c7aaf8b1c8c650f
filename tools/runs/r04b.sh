#!/bin/bash
# glue rewrite (trimap encode chain, PPM chain): kernel tests, frame tests, bench, kernel stats
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "trimap or ppm or glue or kernel_fuzz or reference_vectors or maxpool" > $O/ktests.log 2>&1
echo "ktests rc $?" >> $O/ktests.log; tail -4 $O/ktests.log
python -m pytest tests/test_gpu_frame.py -x -q -m gpu -k "sequence_vs_oracle or frame_fuzz" > $O/ftests.log 2>&1
echo "ftests rc $?" >> $O/ftests.log; tail -4 $O/ftests.log
export OTVM_TUNE_FILE=$PWD/$O/tune.json
python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
head -c 300 $O/bench.json; echo
R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/ks -o ks -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/$O/ks.log 2>&1
cd $R
KS=$(ls $O/ks/*kernel_stats.csv $O/ks/*/*kernel_stats.csv 2>/dev/null | head -1)
python tools/kernel_stats_md.py $KS 23 "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline" > $O/kernel_stats_1080p.md
find $O -name "*kernel_trace.csv" -delete
grep -i "ppm\|edt\|classify\|preprocess\|fba_head\|up4soft" $O/kernel_stats_1080p.md
