#!/bin/bash
# 16x32-pixel blocks for the 32-filter full-resolution layers: tests + same-box A/B
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04r; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu -k "conv2d or linearity or fuzz or 1080p" > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for rep in 1 2 3; do
for v in 0 1; do
  OTVM_PATCH32_ROWS4=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTVM_PATCH32_ROWS4=$v', round(d['value'],2), 'frames/s')" >> $O/ab.txt
done; done
cat $O/ab.txt
