#!/bin/bash
# the whole -m gpu suite, then the round-4 profiles from the same tree
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu -s > $O/gputests.log 2>&1
echo "gpu tests rc $?" >> $O/gputests.log; tail -4 $O/gputests.log
bash tools/profile_r04.sh > $O/profile.log 2>&1; tail -40 $O/profile.log
