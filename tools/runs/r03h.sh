#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
SH="--shape 1024,256,1,1,1,68,120 --shape 256,1024,1,1,1,68,120 --shape 256,256,3,1,1,68,120 --shape 512,128,1,1,1,136,240 --shape 128,512,1,1,1,136,240 --shape 1024,256,1,1,1,136,240 --shape 256,1024,1,1,1,136,240 --shape 2048,512,1,1,1,136,240 --shape 512,2048,1,1,1,136,240 --shape 512,512,3,1,4,136,240 --shape 256,256,3,1,1,272,480 --shape 1024,256,1,1,1,30,52"
for pad in 0 32 48 96; do
  echo "## pad $pad"
  timeout 300 python tools/conv_bench.py --tune all --iters 30 --pad-ld $pad $SH > $O/convbench_pad$pad.txt 2>&1
  python - <<PY
import re
res={}
for line in open('$O/convbench_pad$pad.txt'):
    m=re.match(r'Cin\s+(\d+) Cout\s+(\d+) k(\d) s(\d) d(\d)\s+(\d+)x(\d+)\s+(\S+)\s*:\s+([\d.]+) ms',line)
    if m: res.setdefault(tuple(m.groups()[:7]),{})[m.group(8)]=float(m.group(9))
for k,d in res.items():
    w={c:v for c,v in d.items() if c.startswith('wave')}; o={c:v for c,v in d.items() if not c.startswith('wave')}
    bw=min(w,key=w.get) if w else None; bo=min(o,key=o.get)
    print(k, 'heur %.4f best other %s %.4f | best wave %s %s' % (d['heuristic'],bo,o[bo],bw,('%.4f'%w[bw]) if bw else '-'))
PY
done
