#!/bin/bash
# round 5, lease zu: the live-traffic test and its neighbours
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu --durations=5 2>&1 | grep -v amdgpu | tail -12
