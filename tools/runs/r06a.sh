#!/bin/bash
# lease r06a: the round's first GPU check -- changed tests, the driver's bench command on the T = 100 clip
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06a; O=gpurun_out/r06a
timeout 1500 python -m pytest tests/test_gpu_frame.py -q -x -m gpu -k "conditioning_guard or ill_conditioned" -s > $O/t_guard.log 2>&1; echo "guard rc $?" >> $O/t_guard.log
timeout 900 python -m pytest tests/test_gpu_multirank.py -q -x -m gpu -k "live" -s > $O/t_live.log 2>&1; echo "live rc $?" >> $O/t_live.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "memory_read or memread" > $O/t_mr.log 2>&1; echo "mr rc $?" >> $O/t_mr.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc $?"
tail -3 $O/t_guard.log; tail -3 $O/t_live.log; tail -3 $O/t_mr.log; cut -c1-1500 $O/bench_driver.json
