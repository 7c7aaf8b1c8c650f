#!/bin/bash
# round-2 profiles (GPU box): kernel stats (1080p, 480p), MFMA-busy PMC pass, FETCH / WRITE PMC passes (separate runs, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; --kernel-trace only beside --pmc).  The tuned configurations are
# timed once in an un-profiled run and read from OTVM_TUNE_FILE by the profiled ones.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_r02; mkdir -p $O
export OTVM_TUNE_FILE=$O/tune_cache.json
cd $R
python bench.py --no-cpu-baseline --layer-report $O/layers_1080p.json --tune-report $O/tune_1080p.json > $O/bench_1080p.json 2> $O/bench_1080p.err
python bench.py --no-cpu-baseline --height 480 --width 832 --steps 47 --warmup 3 --layer-report $O/layers_480p.json --tune-report $O/tune_480p.json > $O/bench_480p.json 2> $O/bench_480p.err
python tools/conv_bench.py --tune all --bias 1 --shape 4,64,7,2,1,1088,1920 --shape 12,64,7,2,1,1088,1920 --shape 24,64,7,2,1,1088,1920 > $O/stem_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1080 -o ks -- $CMD > $O/ks1080.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks480 -o ks -- $CMD --height 480 --width 832 --steps 47 > $O/ks480.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -o m -- $CMD --steps 8 > $O/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $CMD --steps 8 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $CMD --steps 8 > $O/write.log 2>&1
cd $R
# keep the merged output small: drop the per-dispatch traces of the counter passes once summarised
python tools/kernel_stats_md.py $O/ks1080/ks_kernel_stats.csv 23 "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline" > $O/kernel_stats_1080p.md
python tools/kernel_stats_md.py $O/ks480/ks_kernel_stats.csv 50 "python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline" > $O/kernel_stats_480p.md
python tools/pmc_mfma.py $O/mfma > $O/mfma_busy_1080p.md 2>&1
python tools/pmc_traffic.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv 11 > $O/conv_traffic_1080p.json 2>$O/traffic.err
python tools/frame_timeline.py $O/ks1080/ks_kernel_trace.csv > $O/timeline_1080p.md 2>&1
python tools/layer_roofline_md.py $O/layers_1080p.json "1920x1080 (bench.py default run)" > $O/layer_roofline_1080p.md 2>&1
python tools/layer_roofline_md.py $O/layers_480p.json "832x480" > $O/layer_roofline_480p.md 2>&1
python tools/frame_trace_dump.py $O/ks1080/ks_kernel_trace.csv 13 60 > $O/frame_trace_1080p.txt 2>&1
python tools/frame_trace_dump.py $O/ks480/ks_kernel_trace.csv 30 25 > $O/frame_trace_480p.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_probe tools/probes/mfma_probe.hip && /tmp/mfma_probe > $O/mfma_power_ceiling.txt 2>&1
python tools/memread_bench.py > $O/memread_bench.txt 2>&1
rm -f $O/mfma/*kernel_trace.csv $O/fetch/*kernel_trace.csv $O/write/*kernel_trace.csv $O/ks1080/*kernel_trace.csv $O/ks480/*kernel_trace.csv
ls -la $O $O/mfma $O/fetch | head -40
cat $O/bench_1080p.json | head -c 600; echo; cat $O/bench_480p.json | head -c 300; echo
grep -v amdgpu $O/stem_bench.log
head -30 $O/kernel_stats_1080p.md; tail -9 $O/kernel_stats_1080p.md; cat $O/mfma_busy_1080p.md | head -20; cat $O/conv_traffic_1080p.json; cat $O/timeline_1080p.md
