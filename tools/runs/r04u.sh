#!/bin/bash
# full logs of the two checks that complained in r04t
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04u; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/conv_fuzz.py --n 800 --seed 4 --verbose > $O/conv_fuzz.log 2>&1; echo "rc $?" >> $O/conv_fuzz.log
tail -6 $O/conv_fuzz.log
OTVM_PATCH_WIDE_GLDS=0 timeout 900 python tools/conv_fuzz.py --n 800 --seed 4 --verbose > $O/conv_fuzz_noglds.log 2>&1; echo "rc $?" >> $O/conv_fuzz_noglds.log
tail -4 $O/conv_fuzz_noglds.log
timeout 1500 python tools/tune_verify.py > $O/tune_verify_1080p.log 2>&1; echo "rc $?" >> $O/tune_verify_1080p.log
grep -c MISMATCH $O/tune_verify_1080p.log; grep MISMATCH $O/tune_verify_1080p.log | awk '{print $2}' | sort | uniq -c | sort -rn | head -30
tail -3 $O/tune_verify_1080p.log
