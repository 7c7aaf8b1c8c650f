#!/bin/bash
# round 5, lease zt: bench.py with its live PMC leg (roofline.traffic measured by two child runs under rocprofv3) -- the driver's command
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zt; mkdir -p $O
cd $R
S0=$SECONDS; python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$? wall $((SECONDS-S0)) s"
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'traffic MB', r['traffic']/1e6, 'x', r['traffic']/r['algorithmic_bytes_per_launch']); print(r['traffic_source'])"
OTVM_BENCH_LIVE_PMC=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read())['roofline']; print('LIVE_PMC=0:', r['traffic']/1e6, r['traffic_source'])"
