#!/bin/bash
# lease r06c: planes-128 fused bottleneck -- kernel test, timing with and without the per-stage timers (experiment build)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06c; O=gpurun_out/r06c
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "stm_bottleneck128" > $O/t_bnk.log 2>&1; echo "bnk rc $?" >> $O/t_bnk.log
tail -3 $O/t_bnk.log
for hw in "136 240" "60 104"; do set -- $hw
timeout 600 python tools/bottleneck_bench.py --planes128 --height $1 --width $2 2>&1 | grep -v amdgpu.ids
OTVM_HIP_LIB=$PWD/otvm_amd/variants/libotvm_b128t.so timeout 600 python tools/bottleneck_bench.py --planes128 --height $1 --width $2 2>&1 | grep "per workgroup"
done > $O/stages.txt 2>&1
cat $O/stages.txt
