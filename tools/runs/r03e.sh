#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q -rP > $O/pytest_train.log 2>&1; echo "rc $?" >> $O/pytest_train.log; grep -v "amdgpu.ids\|^$\|Captured\|^---" $O/pytest_train.log | tail -40
