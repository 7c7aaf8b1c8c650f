#!/bin/bash
cd /root/repo
O=gpurun_out/r03am; mkdir -p $O
timeout 900 python tools/gn_tail_stress.py --reps 1500 > $O/gn_tail_stress.txt 2>&1; tail -7 $O/gn_tail_stress.txt
timeout 600 python tools/race_stress.py --reps 8 > $O/race_1080p.txt 2>&1; tail -1 $O/race_1080p.txt
timeout 600 python tools/race_stress.py --height 480 --width 832 --batch 3 --reps 12 > $O/race_480p_b3.txt 2>&1; tail -1 $O/race_480p_b3.txt
OTVM_GRAPHS=1 timeout 600 python tools/race_stress.py --height 480 --width 832 --reps 12 > $O/race_480p_graphs.txt 2>&1; tail -1 $O/race_480p_graphs.txt
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_multirank.py -q -x -m gpu > $O/pytest2.log 2>&1; tail -2 $O/pytest2.log
