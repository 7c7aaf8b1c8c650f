#!/bin/bash
# lease r06g: per-workgroup fixed cost of the 64-filter nine-tap patch tile -- time against the number of 16-channel stages
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06g; O=gpurun_out/r06g
python tools/conv_bench.py --iters 30 --shape 16,64,3,1,1,1088,1920 --shape 32,64,3,1,1,1088,1920 --shape 64,64,3,1,1,1088,1920 --shape 128,64,3,1,1,1088,1920 --shape 256,64,3,1,1,1088,1920 2>&1 | grep -v amdgpu.ids | tee $O/patch64_stages.txt
python tools/conv_bench.py --iters 30 --zero 3 --shape 16,64,3,1,1,1088,1920 --shape 64,64,3,1,1,1088,1920 --shape 256,64,3,1,1,1088,1920 2>&1 | grep -v amdgpu.ids | sed 's/^/zero operands: /' | tee -a $O/patch64_stages.txt
