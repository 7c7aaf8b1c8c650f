#!/bin/bash
cd /root/repo
O=gpurun_out/r03ag; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frame.py tests/test_gpu_train.py tests/test_gpu_multirank.py -q -x -m gpu > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 900 python tools/tune_verify.py > $O/tune_verify.txt 2>&1; tail -3 $O/tune_verify.txt
timeout 600 python tools/conv_fuzz.py --n 400 > $O/conv_fuzz.txt 2>&1; tail -2 $O/conv_fuzz.txt
