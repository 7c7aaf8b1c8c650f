#!/bin/bash
# round 5, lease zd: the kernel tests after the candidate-list assertions follow OTVM_IGEMM_M16 = 2
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zd; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu > $O/tests.log 2>&1; echo "kernel tests rc=$?"; tail -8 $O/tests.log
