#!/bin/bash
# final validation of the round-3 tree: whole GPU suite, the frame / multirank / train tests under hipGraphs, determinism stress
export TMPDIR=/tmp
cd /root/repo
O=gpurun_out/r03aj; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 -rP ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |real" $O/pytest.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
OTVM_GRAPHS=1 timeout 900 python -m pytest tests/test_gpu_frame.py tests/test_gpu_train.py tests/test_gpu_multirank.py -m gpu -x -q > $O/pytest_graphs.log 2>&1; echo "graphs rc $?" >> $O/pytest_graphs.log; tail -3 $O/pytest_graphs.log
timeout 600 python tools/race_stress.py --reps 6 > $O/race_1080p.txt 2>&1; tail -1 $O/race_1080p.txt
timeout 600 python tools/race_stress.py --height 480 --width 832 --batch 3 --reps 10 > $O/race_480p_b3.txt 2>&1; tail -1 $O/race_480p_b3.txt
OTVM_GRAPHS=1 timeout 600 python tools/race_stress.py --height 480 --width 832 --reps 10 > $O/race_480p_graphs.txt 2>&1; tail -1 $O/race_480p_graphs.txt
timeout 600 python tools/frame_fuzz.py --n 20 > $O/frame_fuzz.txt 2>&1; tail -2 $O/frame_fuzz.txt
