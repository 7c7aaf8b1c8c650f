#!/bin/bash
# wider randomised sweep (other seeds) of the final tree
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04fz; mkdir -p $O
export TMPDIR=/tmp
for seed in 11 12 13 14 15; do
  timeout 900 python tools/conv_fuzz.py --n 600 --seed $seed --verbose > $O/conv_fuzz_$seed.log 2>&1; echo "conv_fuzz seed $seed rc $? : $(grep -v '^case' $O/conv_fuzz_$seed.log | tail -1)" >> $O/summary.txt
done
for seed in 21 22 23; do
  timeout 900 python tools/kernel_fuzz.py --n 120 --seed $seed > $O/kernel_fuzz_$seed.log 2>&1; echo "kernel_fuzz seed $seed rc $? : $(tail -1 $O/kernel_fuzz_$seed.log)" >> $O/summary.txt
done
for seed in 31 32; do
  timeout 1500 python tools/frame_fuzz.py --n 24 --seed $seed > $O/frame_fuzz_$seed.log 2>&1; echo "frame_fuzz seed $seed rc $? : $(tail -1 $O/frame_fuzz_$seed.log)" >> $O/summary.txt
done
cat $O/summary.txt
