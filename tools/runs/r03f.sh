#!/bin/bash
# round 3: whole GPU suite with timing (driver limit: 1200 s) + smoke + the round's profile set
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -22 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/profile_r03.sh > $O/profile.log 2>&1; tail -80 $O/profile.log
