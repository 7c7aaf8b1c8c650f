#!/bin/bash
# copy the summaries tools/profile_r02.sh left under gpurun_out/prof_r02 into profiles/ (tracked):  tools/publish_profiles.sh [r02]
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/prof_r02; P=$R/profiles; T=${1:-r02}
set -e
cp $O/bench_1080p.json $P/${T}_bench_1gpu.json; cp $O/bench_480p.json $P/${T}_bench_1gpu_480p.json
cp $O/ks1080/ks_kernel_stats.csv $P/${T}_kernel_stats_f16x3_1080p.csv; cp $O/kernel_stats_1080p.md $P/${T}_kernel_stats_f16x3_1080p.md
cp $O/ks480/ks_kernel_stats.csv $P/${T}_kernel_stats_f16x3_480p.csv; cp $O/kernel_stats_480p.md $P/${T}_kernel_stats_f16x3_480p.md
cp $O/mfma_busy_1080p.md $P/${T}_mfma_busy_f16x3_1080p.md; cp $O/conv_traffic_1080p.json $P/${T}_conv_traffic_f16x3_1920x1080.json
cp $O/timeline_1080p.md $P/${T}_frame_timeline_1080p.md; cp $O/frame_trace_1080p.txt $P/${T}_frame_trace_1080p.txt; cp $O/frame_trace_480p.txt $P/${T}_frame_trace_480p.txt
cp $O/tune_1080p.json $P/${T}_autotune_1080p.json; cp $O/tune_480p.json $P/${T}_autotune_480p.json
cp $O/layer_roofline_1080p.md $P/${T}_layer_roofline_1080p.md; cp $O/layer_roofline_480p.md $P/${T}_layer_roofline_480p.md
cp $O/mfma_power_ceiling.txt $P/${T}_mfma_power_ceiling.txt; cp $O/memread_bench.txt $P/${T}_memread_bench.txt
ls -la $P | grep ${T}_
