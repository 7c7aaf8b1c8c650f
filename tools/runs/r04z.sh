#!/bin/bash
# 64-filter patch tile: 3-tap weight stages by LDS-DMA + 32-byte swizzled patch rows (three workgroups per CU)
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04z; mkdir -p $O
export TMPDIR=/tmp
OTVM_PATCH64_GLDS=1 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu -k "conv or linearity or fuzz" > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
SH="--shape 64,64,3,1,1,1088,1920 --shape 80,64,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 64,64,3,1,1,480,832 --shape 64,64,3,1,1,272,480"
for v in 0 1 0 1; do
  echo "== OTVM_PATCH64_GLDS=$v" >> $O/conv.txt
  OTVM_PATCH64_GLDS=$v python tools/conv_bench.py $SH --tune 241 --iters 30 --bias 1 --gn 1 2>/dev/null >> $O/conv.txt
done
cat $O/conv.txt
