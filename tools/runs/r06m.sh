#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "stm_bottleneck" 2>&1 | tail -3
