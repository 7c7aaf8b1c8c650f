#!/bin/bash
# batch benches + the default bench line on the final tree (traffic from the committed PMC passes of this tree)
cd /root/repo
O=gpurun_out/r03ac; mkdir -p $O
export OTVM_TUNE_FILE=/tmp/tune_r03ac.json
timeout 900 python bench.py > $O/bench_1080p.json 2> $O/bench_1080p.err
timeout 900 python bench.py --batch 2 --no-cpu-baseline > $O/bench_1080p_b2.json 2> $O/bench_1080p_b2.err
for b in 1 2 4; do
  timeout 600 python bench.py --height 480 --width 832 --steps 47 --warmup 3 --batch $b --no-cpu-baseline > $O/bench_480p_b$b.json 2> $O/bench_480p_b$b.err
done
