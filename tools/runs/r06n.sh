#!/bin/bash
# lease r06n: randomised / stress checks, 2000-frame soak and the eval.py-shaped command line on the final round-6 tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06n; O=gpurun_out/r06n
F=$O/fuzz_and_tune_verify.txt
echo "# randomised / stress checks of the final round-6 tree (GPU box; tools/runs/r06n.sh)" > $F
for s in 41 42 43; do echo "conv_fuzz seed $s: $(timeout 900 python tools/conv_fuzz.py --n 500 --seed $s 2>&1 | tail -1)" >> $F; done
for s in 44 45; do echo "conv_fuzz --patch64 seed $s: $(timeout 900 python tools/conv_fuzz.py --n 300 --seed $s --patch64 2>&1 | tail -1)" >> $F; done
for s in 51 52; do echo "kernel_fuzz seed $s: $(timeout 900 python tools/kernel_fuzz.py --n 100 --seed $s 2>&1 | tail -1)" >> $F; done
for s in 61 62; do timeout 1500 python tools/frame_fuzz.py --n 20 --seed $s 2>&1 | tail -2 | sed "s/^/frame_fuzz seed $s: /" >> $F; done
timeout 900 python tools/race_stress.py --reps 20 2>&1 | tail -2 | sed 's/^/race_stress 1080p: /' >> $F
timeout 900 python tools/race_stress.py --height 480 --width 832 --reps 20 2>&1 | tail -2 | sed 's/^/race_stress 480p: /' >> $F
timeout 900 python tools/gn_tail_stress.py --reps 2000 2>&1 | tail -2 | sed 's/^/gn_tail_stress: /' >> $F
timeout 1500 python tools/tune_verify.py 2>&1 | tail -3 | sed 's/^/tune_verify 1080p: /' >> $F
timeout 1500 python tools/tune_verify.py --height 480 --width 832 2>&1 | tail -3 | sed 's/^/tune_verify 480p: /' >> $F
cat $F | cut -c1-250
timeout 900 python tools/soak.py --frames 2000 > $O/soak.txt 2>&1; tail -14 $O/soak.txt | cut -c1-200
timeout 1500 python tools/eval_cli_bench.py > $O/eval_cli_bench.txt 2>&1; tail -6 $O/eval_cli_bench.txt | cut -c1-250
