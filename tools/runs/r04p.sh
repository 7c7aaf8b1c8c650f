#!/bin/bash
# wide patch tiles with the waves as a 4 x 2 grid: conv tests, single-layer timings, same-box A/B
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04p; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu -k "conv or linearity or fuzz" > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
SH="--shape 256,256,3,1,1,272,480 --shape 512,256,3,1,1,272,480 --shape 2048,256,3,1,1,136,240 --shape 320,256,3,1,1,544,960"
for v in 1 2 1 2; do
  echo "== OTVM_PATCH_WIDE_NWN=$v" >> $O/conv.txt
  OTVM_PATCH_WIDE_NWN=$v python tools/conv_bench.py $SH --tune 241 --iters 30 >> $O/conv.txt 2>&1
  OTVM_PATCH_WIDE_NWN=$v python tools/conv_bench.py $SH --tune 241 --iters 30 --gn 1 --bias 1 >> $O/conv.txt 2>&1
done
cat $O/conv.txt
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for rep in 1 2 3; do
for v in 1 2; do
  OTVM_PATCH_WIDE_NWN=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTVM_PATCH_WIDE_NWN=$v', round(d['value'],2), 'frames/s')" >> $O/ab.txt
done; done
cat $O/ab.txt
