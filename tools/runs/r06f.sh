#!/bin/bash
# lease r06f: the whole GPU suite on the final tree (with its printed parity margins), smoke(), the driver's bench command
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06f; O=gpurun_out/r06f
( time timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=15 ) > $O/gpu_suite.log 2>&1; echo "suite rc $?" >> $O/gpu_suite.log
tail -25 $O/gpu_suite.log
{ echo "# every parity margin the GPU suite prints (pytest tests -m gpu -s), final round-6 tree; lease tools/runs/r06f.sh"; grep -a "max-abs\|alpha=\|tie-break\|worst \|vs fp64\|vs oracle" $O/gpu_suite.log | cut -c1-260; grep -a "passed\|failed" $O/gpu_suite.log | tail -2; } > $O/parity_margins.txt
wc -l $O/parity_margins.txt
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc $?"; cut -c1-400 $O/bench_driver.json
