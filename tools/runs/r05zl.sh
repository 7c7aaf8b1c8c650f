#!/bin/bash
# round 5, lease zl: the XCD-aware tile walk (common.h) in all four spatially tiled kernel families -- kernel tests in both walks,
# FETCH_SIZE / WRITE_SIZE of the whole frame in both walks (the measurement behind roofline.traffic), frame rates alternating
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zl; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or patch or head or stem or bottleneck" > $O/tests1.log 2>&1; echo "kernel tests (walk 1) rc=$?"; tail -1 $O/tests1.log
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_tune.json 2> $O/bench_tune.err
for i in 1 2 3; do for m in 0 15; do
  OTVM_TILE_WALK=$m python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTVM_TILE_WALK=$m 1080p', round(d['value'],2), 'frames/s')" | tee -a $O/frame.txt
done; done
for i in 1 2; do for m in 0 15; do
  OTVM_TILE_WALK=$m python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTVM_TILE_WALK=$m 832x480', round(d['value'],2), 'frames/s')" | tee -a $O/frame.txt
done; done
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline"
for m in 0 15; do
  OTVM_TILE_WALK=$m rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch$m -o f -- $CMD > $O/fetch$m.log 2>&1
  OTVM_TILE_WALK=$m rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write$m -o w -- $CMD > $O/write$m.log 2>&1
  FC=$(ls $O/fetch$m/*counter_collection.csv $O/fetch$m/*/*counter_collection.csv 2>/dev/null | head -1)
  WC=$(ls $O/write$m/*counter_collection.csv $O/write$m/*/*counter_collection.csv 2>/dev/null | head -1)
  python $R/tools/pmc_traffic.py $FC $WC 11 > $O/conv_traffic_walk$m.json 2>$O/traffic$m.err
  python $R/tools/pmc_glue_traffic.py $O/fetch$m $O/write$m 11 conv_ stm_ > $O/kernel_traffic_walk$m.md 2>>$O/traffic$m.err
  echo "== OTVM_TILE_WALK=$m"; grep -E "traffic_bytes_per_launch|traffic_bytes_per_frame" $O/conv_traffic_walk$m.json
  grep -E "conv_patch|conv_stem|conv_head|stm_bott" $O/kernel_traffic_walk$m.md | cut -c1-150
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
