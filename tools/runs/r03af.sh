#!/bin/bash
cd /root/repo
O=gpurun_out/r03af; mkdir -p $O
SH="--shape 256,256,3,1,1,272,480 --shape 512,512,3,1,2,136,240 --shape 2048,256,3,1,1,136,240 --shape 512,2048,1,1,1,136,240 --shape 1024,512,3,1,1,136,240 --shape 1024,256,1,1,1,136,240 --shape 256,1024,1,1,1,136,240"
timeout 600 python tools/conv_bench.py --iters 30 --tune 17,225,33,129,145 $SH 2>&1 | grep -v amdgpu > $O/w4.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "tunable or batched_launch or fuzz or candidates" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
cat $O/w4.txt; tail -3 $O/pytest.log
