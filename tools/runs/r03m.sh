#!/bin/bash
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r03m; mkdir -p $O
export OTVM_TUNE_FILE=$O/tune.json
python bench.py --steps 17 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python $R/bench.py --steps 17 --warmup 3 --no-cpu-baseline --no-roofline > $O/ks.log 2>&1
cd $R
KS=$(ls $O/ks/*kernel_stats.csv $O/ks/*/*kernel_stats.csv 2>/dev/null | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$KS")))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('ppm','gn_stats','upsample_bilinear_kernel')): print(n[:70], r['Calls'], r['TotalDurationNs'], float(r['TotalDurationNs'])/int(r['Calls'])/1e3,'us')
PY
find $O -name "*kernel_trace.csv" -delete
