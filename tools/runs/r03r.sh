#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_frame.py -m gpu -x -q -k "range_guard" 2>&1 | tail -15
