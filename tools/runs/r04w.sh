#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04w; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/tune_verify.py --diag > $O/tune_verify_1080p.log 2>&1; echo "rc $?" >> $O/tune_verify_1080p.log
grep DIAG $O/tune_verify_1080p.log | cut -c1-300 | head -10; tail -2 $O/tune_verify_1080p.log
timeout 1500 python tools/tune_verify.py --height 480 --width 832 > $O/tune_verify_480p.log 2>&1; echo "rc $?" >> $O/tune_verify_480p.log
grep MISMATCH $O/tune_verify_480p.log | awk '{print $2}' | sort | uniq -c | head; tail -2 $O/tune_verify_480p.log
