#!/bin/bash
# copy the summaries tools/profile_r06.sh left under gpurun_out/prof_r06 into profiles/ (tracked)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/prof_r06; P=$R/profiles; T=r06
set -e
cp $O/bench_1080p.json $P/${T}_bench_1gpu.json; cp $O/bench_480p.json $P/${T}_bench_1gpu_480p.json
cp $O/bench_f16_1080p.json $P/${T}_bench_f16_1080p.json
cp $O/bench_1080p_batch2.json $P/${T}_bench_1080p_batch2.json; cp $O/bench_480p_batch4.json $P/${T}_bench_480p_batch4.json
cp $O/kernel_stats_1080p.csv $P/${T}_kernel_stats_f16x3_1080p.csv; cp $O/kernel_stats_1080p.md $P/${T}_kernel_stats_f16x3_1080p.md
cp $O/kernel_stats_480p.csv $P/${T}_kernel_stats_f16x3_480p.csv; cp $O/kernel_stats_480p.md $P/${T}_kernel_stats_f16x3_480p.md
cp $O/mfma_busy_1080p.md $P/${T}_mfma_busy_f16x3_1080p.md; cp $O/conv_traffic_1080p.json $P/${T}_conv_traffic_f16x3_1920x1080.json
cp $O/kernel_traffic_gbps_1080p.md $P/${T}_kernel_traffic_gbps_1080p.md
cp $O/tune_1080p.json $P/${T}_autotune_1080p.json; cp $O/tune_480p.json $P/${T}_autotune_480p.json
cp $O/layer_roofline_1080p.md $P/${T}_layer_roofline_1080p.md; cp $O/layer_roofline_480p.md $P/${T}_layer_roofline_480p.md
cp $O/ab_1080p.txt $P/${T}_ab_1080p.txt
[ -s $O/bench_4k_T200_growing.json ] && cp $O/bench_4k_T200_growing.json $P/${T}_bench_4k_T200_growing.json
git -C $R status --short profiles | head -40
