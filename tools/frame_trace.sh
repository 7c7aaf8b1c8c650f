#!/bin/bash
# kernel-trace timeline of a few steady-state frames: tools/frame_trace.sh [extra bench.py arguments, e.g. --height 480 --width 832]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/frame_trace; mkdir -p $O
export OTVM_TUNE_FILE=$O/tune_cache.json
cd $R
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline "$@" 2>$O/bench.err | tee $O/bench.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $O/kt.log 2>&1
cd $R
python tools/frame_trace_dump.py $O/kt/kt_kernel_trace.csv 13 ${MIN_US:-60} > $O/frame13.txt 2>&1
python tools/frame_trace_dump.py $O/kt/kt_kernel_trace.csv 15 ${MIN_US:-60} > $O/frame15.txt 2>&1
python tools/frame_timeline.py $O/kt/kt_kernel_trace.csv > $O/timeline.md 2>&1
rm -f $O/kt/kt_kernel_trace.csv
cat $O/timeline.md; head -5 $O/frame13.txt; tail -2 $O/frame13.txt
