#!/bin/bash
# round-6 profiles (GPU box), all from ONE tree: bench JSONs (1080p, 480p, f16 mode, batches, 4K stress), per-layer roofline, kernel
# stats, MFMA-busy PMC pass, FETCH / WRITE PMC passes (separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes)
# -> conv traffic per launch and GB/s of the glue kernels, same-box A/Bs against the round-4 tree (_old/) and the round's switches.
# The tuned configurations are timed once in an un-profiled run and read from OTVM_TUNE_FILE by the profiled ones.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_r06; mkdir -p $O
export OTVM_TUNE_FILE=$O/tune_cache.json
cd $R
python bench.py --layer-report $O/layers_1080p.json --tune-report $O/tune_1080p.json > $O/bench_1080p.json 2> $O/bench_1080p.err
python bench.py --height 480 --width 832 --steps 47 --warmup 3 --layer-report $O/layers_480p.json --tune-report $O/tune_480p.json > $O/bench_480p.json 2> $O/bench_480p.err
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 20 --warmup 3 --clip-frames 23 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1080 -o ks -- $CMD > $O/ks1080.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks480 -o ks -- $CMD --height 480 --width 832 --steps 47 --clip-frames 50 > $O/ks480.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -o m -- $CMD --steps 8 --clip-frames 11 > $O/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $CMD --steps 8 --clip-frames 11 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $CMD --steps 8 --clip-frames 11 > $O/write.log 2>&1
cd $R
KS1080=$(ls $O/ks1080/*kernel_stats.csv $O/ks1080/*/*kernel_stats.csv 2>/dev/null | head -1)
KS480=$(ls $O/ks480/*kernel_stats.csv $O/ks480/*/*kernel_stats.csv 2>/dev/null | head -1)
python tools/kernel_stats_md.py $KS1080 23 "python bench.py --steps 20 --warmup 3 --clip-frames 23 --no-cpu-baseline --no-roofline" > $O/kernel_stats_1080p.md
python tools/kernel_stats_md.py $KS480 50 "python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline" > $O/kernel_stats_480p.md
cp $KS1080 $O/kernel_stats_1080p.csv; cp $KS480 $O/kernel_stats_480p.csv
python tools/pmc_mfma.py $O/mfma > $O/mfma_busy_1080p.md 2>&1
FC=$(ls $O/fetch/*counter_collection.csv $O/fetch/*/*counter_collection.csv 2>/dev/null | head -1)
WC=$(ls $O/write/*counter_collection.csv $O/write/*/*counter_collection.csv 2>/dev/null | head -1)
python tools/pmc_traffic.py $FC $WC 11 > $O/conv_traffic_1080p.json 2>$O/traffic.err
python tools/pmc_glue_traffic.py $O/fetch $O/write 11 > $O/kernel_traffic_gbps_1080p.md 2>>$O/traffic.err
python tools/layer_roofline_md.py $O/layers_1080p.json "1920x1080 (bench.py default run)" > $O/layer_roofline_1080p.md 2>&1
python tools/layer_roofline_md.py $O/layers_480p.json "832x480" > $O/layer_roofline_480p.md 2>&1
python bench.py --precision f16 --no-cpu-baseline --layer-report $O/layers_f16_1080p.json > $O/bench_f16_1080p.json 2> $O/bench_f16.err
python bench.py --batch 2 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_1080p_batch2.json 2> $O/bench_b2.err
python bench.py --height 480 --width 832 --batch 4 --steps 47 --warmup 3 --no-cpu-baseline > $O/bench_480p_batch4.json 2> $O/bench_480b4.err
rm -f $O/ab_1080p.txt
PL=$R/otvm_amd/libotvm_hip_probes.so
for v in "OTVM_FUSE_STM_BLOCK128=1" "OTVM_FUSE_STM_BLOCK128=0" "OTVM_GN_PREDICT=0" "OTVM_FUSE_HEAD=0" "OTVM_FUSE_STM_BLOCK128=1" "OTVM_FUSE_STM_BLOCK128=0"; do
  env $v OTVM_BENCH_LIVE_PMC=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), 'frames/s')" >> $O/ab_1080p.txt
done
if [ -f $PL ]; then for v in "OTVM_TILE_WALK=11" "OTVM_TILE_WALK=0" "OTVM_IGEMM_M16=0" "OTVM_PATCH_M16=0" "OTVM_IGEMM_GLDS=0" "OTVM_TILE_WALK=11"; do
  env OTVM_HIP_LIB=$PL $v OTVM_BENCH_LIVE_PMC=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('probes library, $v', round(d['value'],2), 'frames/s')" >> $O/ab_1080p.txt
done; fi
[ -f $R/otvm_amd/variants/libotvm_nolean.so ] && for rep in 1 2; do
  OTVM_BENCH_LIVE_PMC=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lean staging of the nine-tap tiles (default)', round(d['value'],2), 'frames/s')" >> $O/ab_1080p.txt
  OTVM_HIP_LIB=$R/otvm_amd/variants/libotvm_nolean.so OTVM_BENCH_LIVE_PMC=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-5 staging (-DOTVM_PATCH_NO_LEAN=1)', round(d['value'],2), 'frames/s')" >> $O/ab_1080p.txt
done
(cd _prev 2>/dev/null && unset OTVM_TUNE_FILE && OTVM_BENCH_LIVE_PMC=0 python bench.py --steps 40 --warmup 60 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-5 tree (5b5c5c4), same box, T = 100 clip', round(d['value'],2), 'frames/s')" >> $O/ab_1080p.txt)
(cd _prev 2>/dev/null && unset OTVM_TUNE_FILE && OTVM_BENCH_LIVE_PMC=0 python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round-5 tree (5b5c5c4), same box, 832x480', round(d['value'],2), 'frames/s')" >> $O/ab_1080p.txt)
cat $O/ab_1080p.txt
# BASELINE configs[4]: 3840x2160, T=200, every frame memorised, nothing evicted
(unset OTVM_TUNE_FILE; OTVM_BENCH_LIVE_PMC=1 python bench.py --height 2160 --width 3840 --steps 197 --warmup 3 --stress-bank --no-cpu-baseline > $O/bench_4k_T200_growing.json 2> $O/bench_4k.err)
head -c 300 $O/bench_4k_T200_growing.json; echo
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/bench_1080p.json | head -c 700; echo; cat $O/bench_480p.json | head -c 300; echo; head -c 400 $O/bench_f16_1080p.json; echo
head -24 $O/kernel_stats_1080p.md | cut -c1-180; cat $O/mfma_busy_1080p.md | head -16 | cut -c1-180; cat $O/conv_traffic_1080p.json; head -30 $O/kernel_traffic_gbps_1080p.md | cut -c1-180
