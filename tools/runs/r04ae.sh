#!/bin/bash
# implicit-GEMM tiles with transposed accumulators (epilogue without LDS): tests, single-layer timings, same-box A/B
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04ae; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or tunable or fuzz or predict or bottleneck" > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
timeout 900 python tools/conv_fuzz.py --n 500 --seed 21 2>&1 | tail -1
SH="--shape 256,1024,1,1,1,136,240 --shape 512,2048,1,1,1,136,240 --shape 64,256,1,1,1,272,480 --shape 1024,256,1,1,1,136,240 --shape 256,256,3,1,1,68,120 --shape 1024,512,3,1,1,68,120 --shape 512,512,3,1,4,136,240"
for v in 0 1 0 1; do
  echo "== OTVM_CONV_TR=$v" >> $O/conv.txt
  OTVM_CONV_TR=$v python tools/conv_bench.py $SH --iters 30 --bias 1 --res 1 --relu 0 2>/dev/null >> $O/conv.txt
done
cat $O/conv.txt
for rep in 1 2 3; do
for v in 0 1; do
  OTVM_CONV_TR=$v OTVM_TUNE_FILE=$O/tune_cache_$v.json python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p OTVM_CONV_TR=$v', round(d['value'],2), 'frames/s', d['alpha_checksum'])" >> $O/ab.txt
done; done
for rep in 1 2; do
for v in 0 1; do
  OTVM_CONV_TR=$v OTVM_TUNE_FILE=$O/tune_cache480_$v.json python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p OTVM_CONV_TR=$v', round(d['value'],2), 'frames/s', d['alpha_checksum'])" >> $O/ab.txt
done; done
cat $O/ab.txt
