#!/bin/bash
# round 5, lease zp: the driver's round-end sequence on the final tree -- smoke(), then bench.py with its default flags
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -3
python bench.py 2>/dev/null | tail -1 > gpurun_out/final_bench.json; python -c "
import json; d=json.load(open('gpurun_out/final_bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'traffic MB', r['traffic']/1e6, 'x', r['traffic']/r['algorithmic_bytes_per_launch'], 'cpu', d['cpu_baseline']['value'])"
