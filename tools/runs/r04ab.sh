#!/bin/bash
# 256-channel patch tiles of 4 / 5 / 6 rows on the maps where 8-row tiles leave half the chip idle (136x240: 136 tiles)
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04ab; mkdir -p $O
export TMPDIR=/tmp
SH="--shape 2048,256,3,1,1,136,240 --shape 256,256,3,1,1,136,240 --shape 512,256,3,1,1,272,480 --shape 256,256,3,1,1,272,480 --shape 256,256,3,1,1,120,208"
for v in 0 6 5 4 0 6 5 4; do
  echo "== OTVM_PATCH_WIDE_TH=$v" >> $O/conv.txt
  OTVM_PATCH_WIDE_TH=$v python tools/conv_bench.py $SH --tune 241 --iters 30 --bias 1 --gn 1 2>/dev/null >> $O/conv.txt
done
cat $O/conv.txt
for v in 6 5 4; do
OTVM_PATCH_WIDE_TH=$v timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "every_tunable or race_free or fuzz_all" 2>&1 | tail -1
done
