#!/bin/bash
# round 5, FINAL tree (XCD-aware tile walk in the patch / stem / bottleneck kernels): the whole GPU suite (durations),
# tools/profile_r05.sh, the eval.py-shaped command line with its IO, a 2000-frame soak, the randomised / stress checks -- one box
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zo; mkdir -p $O
cd $R
timeout 1100 python -m pytest tests -q -m gpu --durations=40 > $O/tests.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
bash tools/profile_r05.sh > $O/profile.log 2>&1; tail -12 $O/profile.log | cut -c1-300
timeout 600 python tools/eval_cli_bench.py 2>&1 | grep -v amdgpu | tee $O/eval_cli_bench.txt | tail -6
timeout 600 python tools/soak.py 2>&1 | grep -v amdgpu | tee $O/soak_1080p.txt | tail -3
F=$O/fuzz.txt
for s in 41 42 43; do timeout 600 python tools/conv_fuzz.py --n 500 --seed $s 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/conv_fuzz seed $s: /" | tee -a $F; done
for s in 44 45; do timeout 600 python tools/conv_fuzz.py --n 300 --seed $s --patch64 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/conv_fuzz --patch64 seed $s: /" | tee -a $F; done
for s in 51 52; do timeout 600 python tools/kernel_fuzz.py --n 100 --seed $s 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/kernel_fuzz seed $s: /" | tee -a $F; done
for s in 61 62; do timeout 900 python tools/frame_fuzz.py --n 20 --seed $s --keep-going 2>&1 | grep -v amdgpu | tail -2 | sed "s/^/frame_fuzz seed $s: /" | tee -a $F; done
timeout 600 python tools/race_stress.py 2>&1 | grep -v amdgpu | tail -1 | tee -a $F
timeout 600 python tools/gn_tail_stress.py 2>&1 | grep -v amdgpu | tail -2 | tee -a $F
timeout 900 python tools/tune_verify.py 2>&1 | grep -v amdgpu | tail -1 | tee -a $F
timeout 900 python tools/tune_verify.py --height 480 --width 832 2>&1 | grep -v amdgpu | tail -1 | tee -a $F
