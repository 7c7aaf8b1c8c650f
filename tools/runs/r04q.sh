#!/bin/bash
# 32-channel patch tiles with four rows per wave (half the A-fragment reads per MFMA): single-layer timings
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04q; mkdir -p $O
export TMPDIR=/tmp
SH="--shape 80,32,3,1,1,1088,1920 --shape 64,32,3,1,1,1088,1920 --shape 64,32,3,1,1,480,832"
for v in 0 1 2 0 1 2; do
  echo "== OTVM_PATCH32_VARIANT=$v" >> $O/conv.txt
  OTVM_PATCH32_VARIANT=$v python tools/conv_bench.py $SH --tune 241 --iters 30 --bias 1 2>/dev/null >> $O/conv.txt
done
cat $O/conv.txt
