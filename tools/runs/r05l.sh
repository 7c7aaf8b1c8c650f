#!/bin/bash
# round 5, lease l: the whole GPU suite (durations) and then tools/profile_r05.sh, one box, one tree
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05l; mkdir -p $O
cd $R
timeout 1100 python -m pytest tests -q -m gpu --durations=40 > $O/tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
bash tools/profile_r05.sh > $O/profile.log 2>&1; tail -60 $O/profile.log
