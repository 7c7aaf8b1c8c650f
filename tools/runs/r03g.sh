#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "tunable or batched_launch or fuzz or conv2d or split_k" > $O/pytest_k.log 2>&1; echo "rc $?" >> $O/pytest_k.log; tail -5 $O/pytest_k.log
SH="--shape 256,256,3,1,1,30,52 --shape 1024,256,1,1,1,30,52 --shape 256,1024,1,1,1,30,52 --shape 128,128,3,1,1,60,104 --shape 512,128,1,1,1,60,104 --shape 128,512,1,1,1,60,104 --shape 256,256,3,1,2,60,104 --shape 1024,256,1,1,1,60,104 --shape 256,1024,1,1,1,60,104 --shape 512,512,3,1,4,60,104 --shape 64,64,3,1,1,120,208 --shape 256,64,1,1,1,120,208 --shape 64,256,1,1,1,120,208 --shape 256,256,3,1,1,68,120 --shape 1024,256,1,1,1,68,120 --shape 256,1024,1,1,1,68,120 --shape 128,128,3,1,1,136,240 --shape 512,128,1,1,1,136,240 --shape 128,512,1,1,1,136,240 --shape 1024,256,1,1,1,136,240 --shape 256,1024,1,1,1,136,240 --shape 256,256,3,1,2,136,240"
timeout 300 python tools/conv_bench.py --tune all --iters 30 $SH > $O/convbench.txt 2>&1
python - <<'PY'
import re
res={}
for line in open('gpurun_out/r03g/convbench.txt'):
    m=re.match(r'Cin\s+(\d+) Cout\s+(\d+) k(\d) s(\d) d(\d)\s+(\d+)x(\d+)\s+(\S+)\s*:\s+([\d.]+) ms',line)
    if m: res.setdefault(tuple(m.groups()[:7]),{})[m.group(8)]=float(m.group(9))
for k,d in res.items():
    w={c:v for c,v in d.items() if c.startswith('wave')}; o={c:v for c,v in d.items() if not c.startswith('wave')}
    bw=min(w,key=w.get) if w else None; bo=min(o,key=o.get)
    print(k, 'best other %s %.4f | best wave %s %s' % (bo,o[bo],bw,('%.4f'%w[bw]) if bw else '-'))
PY
for v in 1 0; do
  OTVM_WAVE_TILE=$v timeout 300 python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline > $O/bench_480p_w$v.json 2> $O/bench_480p_w$v.err
  OTVM_WAVE_TILE=$v timeout 300 python bench.py --steps 47 --warmup 3 --no-cpu-baseline > $O/bench_1080p_w$v.json 2> $O/bench_1080p_w$v.err
  python -c "import json;a=json.load(open('$O/bench_480p_w$v.json'));b=json.load(open('$O/bench_1080p_w$v.json'));print('wave=$v: 480p %.1f fps conv frac %.3f | 1080p %.2f fps conv frac %.3f' % (a['value'],a['roofline']['frac'],b['value'],b['roofline']['frac']))"
done
