#!/bin/bash
# round 5, lease d: where the frame's time is after the LDS-DMA tiles: kernel stats (rocprofv3 --kernel-trace --stats), per-layer
# roofline, MFMA-busy PMC pass -- interim snapshot (the round's final profiles come from tools/profile_r05.sh)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05d; mkdir -p $O
export OTVM_TUNE_FILE=$O/tune_cache.json
cd $R
python bench.py --layer-report $O/layers_1080p.json --tune-report $O/tune_1080p.json --no-cpu-baseline > $O/bench_1080p.json 2> $O/bench_1080p.err
head -c 300 $O/bench_1080p.json; echo
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1080 -o ks -- $CMD > $O/ks1080.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -o m -- $CMD --steps 8 > $O/mfma.log 2>&1
cd $R
KS1080=$(ls $O/ks1080/*kernel_stats.csv $O/ks1080/*/*kernel_stats.csv 2>/dev/null | head -1)
python tools/kernel_stats_md.py $KS1080 23 "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline" > $O/kernel_stats_1080p.md
python tools/pmc_mfma.py $O/mfma > $O/mfma_busy_1080p.md 2>&1
python tools/layer_roofline_md.py $O/layers_1080p.json "1920x1080 (bench.py default run)" > $O/layer_roofline_1080p.md 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
head -60 $O/kernel_stats_1080p.md | cut -c1-200; head -40 $O/mfma_busy_1080p.md | cut -c1-200
