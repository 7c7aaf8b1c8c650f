#!/bin/bash
# round 5, lease zs: last check of the committed tree -- the whole GPU suite, smoke(), bench.py with its default flags
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zs; mkdir -p $O
cd $R
timeout 1100 python -m pytest tests -q -m gpu > $O/tests.log 2>&1; echo "gpu suite rc=$?"; tail -2 $O/tests.log
bash tools/runs/r05zp.sh
