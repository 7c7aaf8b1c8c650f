#!/bin/bash
# round 5, lease k: stem kernel with up-front buffer loads vs the round-4 tree; kernel tests of the stem
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05k; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "every_tunable or batched_launch or conv2d" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
S="--shape 24,64,7,2,1,1088,1920 --shape 12,64,7,2,1,1088,1920 --shape 24,64,7,2,1,480,832"
for i in 1 2; do
python tools/conv_bench.py --iters 30 --tune 209 $S 2>&1 | grep -v amdgpu | tee -a $O/stem.txt
(cd _old && python tools/conv_bench.py --iters 30 --tune 209 $S 2>&1 | grep -v amdgpu | sed 's/$/   (round-4 tree)/' | tee -a $O/stem.txt)
done
