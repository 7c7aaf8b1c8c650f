#!/bin/bash
# round 5, lease u: is a conv launch's time set by power?  The same launches with all-zero inputs / weights / both (the matrix
# cores draw far less on zero operands: 2283 vs 1554 TFLOP/s in the pure-MFMA probe) -- if the time follows, the launch is bound
# by the power limit; if it stays, by its structure
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05u; mkdir -p $O
cd $R
S="--shape 64,64,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 256,256,3,1,1,272,480 --shape 512,512,3,1,4,136,240 --shape 256,1024,1,1,1,136,240 --shape 1024,256,1,1,1,136,240 --shape 2048,512,1,1,1,136,240"
for z in 0 1 2 3 0; do echo "--- --zero $z" | tee -a $O/zero.txt; python tools/conv_bench.py --iters 30 --zero $z $S 2>&1 | grep -v amdgpu | tee -a $O/zero.txt; done
