#!/bin/bash
# round 5, lease g: the whole -m gpu suite with per-test durations (the driver's limit is 1200 s)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=45 ) > $O/tests.log 2>&1; echo "suite rc=$?" | tee -a $O/tests.log
tail -70 $O/tests.log
