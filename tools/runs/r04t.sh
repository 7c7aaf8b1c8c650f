#!/bin/bash
# randomised / stress checks of the final round-4 tree
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04t; mkdir -p $O
export TMPDIR=/tmp
{
echo "# randomised checks of the final round-4 tree (GPU box)"
timeout 900 python tools/conv_fuzz.py --n 800 --seed 4 2>&1 | tail -1
timeout 900 python tools/kernel_fuzz.py --n 150 --seed 4 2>&1 | tail -1
timeout 1500 python tools/frame_fuzz.py --n 30 --seed 4 2>&1 | tail -1
timeout 1500 python tools/tune_verify.py 2>&1 | tail -1
timeout 1500 python tools/tune_verify.py --height 480 --width 832 2>&1 | tail -1
timeout 900 python tools/race_stress.py --frames 14 --reps 20 2>&1 | tail -2
timeout 900 python tools/gn_tail_stress.py --reps 2000 2>&1 | tail -2
} > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
