#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04v; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/conv_fuzz.py --n 800 --seed 4 > $O/conv_fuzz.log 2>&1; echo "rc $?" >> $O/conv_fuzz.log
tail -3 $O/conv_fuzz.log
timeout 1500 python tools/tune_verify.py --diag > $O/tune_verify_1080p.log 2>&1; echo "rc $?" >> $O/tune_verify_1080p.log
grep DIAG $O/tune_verify_1080p.log | head -20; grep -c MISMATCH $O/tune_verify_1080p.log; tail -2 $O/tune_verify_1080p.log
SH="--shape 256,256,3,1,1,272,480 --shape 512,256,3,1,1,272,480 --shape 2048,256,3,1,1,136,240 --shape 320,256,3,1,1,544,960"
for v in 0 1 0 1; do
  echo "== OTVM_PATCH_WIDE_GLDS=$v" >> $O/conv.txt
  OTVM_PATCH_WIDE_GLDS=$v python tools/conv_bench.py $SH --tune 241 --iters 30 --gn 1 --bias 1 2>/dev/null >> $O/conv.txt
done
cat $O/conv.txt
