#!/bin/bash
# the heads read the composite from a dense [P][4] copy instead of D80[67:70]: tests, same-box A/B, head kernel time in the frame
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$PWD/gpurun_out/r04af; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_frame.py tests/test_gpu_train.py tests/test_gpu_fullsize.py -x -q -m gpu -k "sequence or batched or eval_cli or training or 1080p_two or 480p or hazard or reproducible" > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for rep in 1 2 3; do
for v in 0 1; do
  OTVM_HEAD_RGB4=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTVM_HEAD_RGB4=$v', round(d['value'],2), 'frames/s', d['alpha_checksum'])" >> $O/ab.txt
done; done
cat $O/ab.txt
cd /tmp
for v in 0 1; do
  OTVM_HEAD_RGB4=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$v -o ks -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/ks$v.log 2>&1
  KS=$(ls $O/ks$v/*kernel_stats.csv $O/ks$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  (cd $R && python tools/kernel_stats_md.py $KS 23 "OTVM_HEAD_RGB4=$v bench" > $O/kernel_stats_$v.md)
  grep -E "preprocess|head16" $O/kernel_stats_$v.md | cut -c1-120
  find $O/ks$v -name "*kernel_trace.csv" -delete
done
