#!/bin/bash
# rows written whole (ABI 18): kernel + frame + training tests, same-box A/B, kernel stats of both routes
cd "$GRAFT_REPO_ROOT"; R=$PWD; O=$PWD/gpurun_out/r04o; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frame.py tests/test_gpu_train.py tests/test_gpu_fullsize.py -x -q -m gpu \
   -k "rows_written or trimap or head or glue or sequence or batched or eval_cli or training or 1080p or 480p or hazard" > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log; tail -6 $O/tests.log
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for rep in 1 2 3; do
for v in OTVM_ROWS_WHOLE=0 OTVM_ROWS_WHOLE=1; do
  env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), 'frames/s')" >> $O/ab.txt
done; done
cat $O/ab.txt
cd /tmp
for v in 0 1; do
  OTVM_ROWS_WHOLE=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$v -o ks -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/ks$v.log 2>&1
  KS=$(ls $O/ks$v/*kernel_stats.csv $O/ks$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  (cd $R && python tools/kernel_stats_md.py $KS 23 "OTVM_ROWS_WHOLE=$v bench" > $O/kernel_stats_$v.md)
  grep -E "preprocess|edt_rows|head16|gn_apply_kernel" $O/kernel_stats_$v.md
done
