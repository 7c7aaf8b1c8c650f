#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06h; O=gpurun_out/r06h
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "beyond_2_gib or nine_tap" -s 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -m gpu -k "4k_growing" -s 2>&1 | grep -a "4K growing\|passed\|failed" | tail -5
