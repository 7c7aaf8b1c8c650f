#!/bin/bash
# lease r06b: the planes-128 fused STM bottleneck -- kernel test, then fused vs three launches at the 1080p / 480p map sizes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06b; O=gpurun_out/r06b
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "stm_bottleneck" -s > $O/t_bnk.log 2>&1; echo "bnk rc $?" >> $O/t_bnk.log
tail -5 $O/t_bnk.log
timeout 600 python tools/bottleneck_bench.py --planes128 --height 136 --width 240 > $O/bench128.txt 2>&1
timeout 600 python tools/bottleneck_bench.py --planes128 --height 60 --width 104 >> $O/bench128.txt 2>&1
cat $O/bench128.txt
