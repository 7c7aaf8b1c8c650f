#!/bin/bash
# the whole -m gpu suite (with durations), then the round-4 profiles from the same tree
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu --durations=30 > $O/gputests.log 2>&1
echo "gpu tests rc $?" >> $O/gputests.log; tail -45 $O/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/profile_r04.sh > $O/profile.log 2>&1; tail -12 $O/profile.log
