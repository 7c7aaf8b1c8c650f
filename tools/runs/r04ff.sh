#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04ff; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/frame_fuzz.py --n 15 --seed 31 --trace --keep-going > $O/f16x3.log 2>&1
timeout 1500 python tools/frame_fuzz.py --n 15 --seed 31 --trace --keep-going --precision f32 > $O/f32.log 2>&1
(cd _old && timeout 1500 python ../tools/frame_fuzz.py --n 15 --seed 31 --trace --keep-going > $O/round3_tree.log 2>&1)
grep "39x98" $O/f16x3.log; echo; grep "39x98" $O/f32.log; echo; grep "39x98" $O/round3_tree.log; tail -1 $O/f16x3.log $O/f32.log $O/round3_tree.log
