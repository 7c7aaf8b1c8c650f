#!/bin/bash
# round 3: determinism stress of the batched path (direct launches and graphs), whole suite under OTVM_GRAPHS=1
export TMPDIR=/tmp
O=gpurun_out/r03k; mkdir -p $O
OTVM_GRAPHS=0 timeout 600 python tools/race_stress.py --height 480 --width 832 --frames 14 --reps 25 --batch 3 2>&1 | tail -3 | tee $O/race_480p_b3_direct.txt
OTVM_GRAPHS=1 timeout 600 python tools/race_stress.py --height 480 --width 832 --frames 14 --reps 25 --batch 3 2>&1 | tail -3 | tee $O/race_480p_b3_graphs.txt
timeout 600 python tools/race_stress.py --height 1080 --width 1920 --frames 14 --reps 10 --batch 2 2>&1 | tail -3 | tee $O/race_1080p_b2.txt
timeout 600 python tools/race_stress.py --height 1080 --width 1920 --frames 14 --reps 15 2>&1 | tail -3 | tee $O/race_1080p_b1.txt
( time OTVM_GRAPHS=1 timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py::test_4k_growing_bank_frame_vs_oracle ) > $O/pytest_graphs.log 2>&1; echo "rc $?" >> $O/pytest_graphs.log; tail -8 $O/pytest_graphs.log
