#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04y; mkdir -p $O
export TMPDIR=/tmp
SH="--shape 64,64,3,1,1,1088,1920 --shape 80,64,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 64,64,3,1,1,480,832 --shape 80,32,3,1,1,1088,1920 --shape 64,32,3,1,1,1088,1920 --shape 64,32,3,1,1,480,832"
for v in 0 1 0 1; do
  echo "== OTVM_PATCH64_GLDS=$v OTVM_PATCH32_GLDS=$v" >> $O/conv.txt
  OTVM_PATCH64_GLDS=$v OTVM_PATCH32_GLDS=$v python tools/conv_bench.py $SH --tune 241 --iters 30 --bias 1 --gn 1 2>/dev/null >> $O/conv.txt
done
cat $O/conv.txt
