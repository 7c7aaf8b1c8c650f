#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04aa; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "race_free or every_tunable" > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log; tail -5 $O/tests.log
