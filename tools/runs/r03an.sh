#!/bin/bash
cd /root/repo
O=gpurun_out/r03an; mkdir -p $O
export OTVM_TUNE_FILE=/tmp/tune_r03an.json
for rep in 1 2; do
for v in 2048 512 256; do
  export OTVM_GN_STATS_BLOCKS=$v
  timeout 600 python bench.py --height 480 --width 832 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_480p_${v}_$rep.json 2> $O/bench_480p_${v}_$rep.err
done
done
for v in 2048 256; do
  export OTVM_GN_STATS_BLOCKS=$v
  timeout 900 python bench.py --steps 97 --warmup 3 --no-cpu-baseline > $O/bench_1080p_${v}.json 2> $O/bench_1080p_${v}.err
done
