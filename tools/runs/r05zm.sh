#!/bin/bash
# round 5, lease zm: the XCD-aware tile walk per kernel family, each alone on the device (head conv, stems, fused bottleneck),
# walk 0 / 1 alternating; band heights for the head conv
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zm; mkdir -p $O
cd $R
for i in 1 2 3; do for m in 0 4; do
  OTVM_TILE_WALK=$m python tools/head_bench.py --iters 50 --size 1088,1920 2>&1 | grep -v amdgpu | sed "s/^/WALK=$m  /" | tee -a $O/head.txt
done; done
for b in 2 4 16; do OTVM_TILE_WALK=4 OTVM_TILE_BAND=$b python tools/head_bench.py --iters 50 --size 1088,1920 2>&1 | grep -v amdgpu | sed "s/^/WALK=4 BAND=$b  /" | tee -a $O/head.txt; done
for i in 1 2; do for m in 0 2; do
  OTVM_TILE_WALK=$m python tools/conv_bench.py --iters 30 --shape 4,64,7,2,1,1088,1920 --shape 11,64,7,2,1,1088,1920 --shape 24,64,7,2,1,1088,1920 --tune 209 2>&1 | grep -v amdgpu | sed "s/^/WALK=$m  /" | tee -a $O/stem.txt
done; done
for i in 1 2; do for m in 0 8; do
  OTVM_TILE_WALK=$m python tools/bottleneck_bench.py 2>&1 | grep -v amdgpu | sed "s/^/WALK=$m  /" | tee -a $O/bottleneck.txt
done; done
