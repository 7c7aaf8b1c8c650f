#!/bin/bash
# round 3 final: whole GPU suite (timed; -rP for the parity margins), smoke, the profile set of the final tree
export TMPDIR=/tmp
O=gpurun_out/r03ab; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 -rP ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "passed|failed|rc |real" $O/pytest.log | tail -5
grep -E "steady state .*frame 21|4K growing bank .*frame 3|margin under" $O/pytest.log | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/profile_r03.sh > $O/profile.log 2>&1; tail -70 $O/profile.log
