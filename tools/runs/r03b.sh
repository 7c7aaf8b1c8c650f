#!/bin/bash
# round 3, ring-prefetch A/B: small-map conv shapes per library variant, then whole-frame benches
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
V=otvm_amd/csrc/build/variants
SH="--shape 256,256,3,1,1,30,52 --shape 1024,256,1,1,1,30,52 --shape 256,1024,1,1,1,30,52 --shape 128,128,3,1,1,60,104 --shape 512,128,1,1,1,60,104 --shape 128,512,1,1,1,60,104 --shape 256,256,3,1,2,60,104 --shape 1024,256,1,1,1,60,104 --shape 256,1024,1,1,1,60,104 --shape 512,512,3,1,4,60,104 --shape 64,64,3,1,1,120,208 --shape 256,64,1,1,1,120,208 --shape 64,256,1,1,1,120,208 --shape 256,256,3,1,1,68,120 --shape 1024,256,1,1,1,68,120 --shape 256,1024,1,1,1,68,120 --shape 128,128,3,1,1,136,240 --shape 512,128,1,1,1,136,240 --shape 128,512,1,1,1,136,240"
for v in nopf pf3 default pf8; do
  if [ $v = default ]; then unset OTVM_HIP_LIB; else export OTVM_HIP_LIB=$PWD/$V/libotvm_$v.so; fi
  timeout 300 python tools/conv_bench.py --tune all --iters 30 $SH > $O/convbench_$v.txt 2>&1
  timeout 300 python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline > $O/bench_480p_$v.json 2> $O/bench_480p_$v.err
  timeout 300 python bench.py --steps 47 --warmup 3 --no-cpu-baseline > $O/bench_1080p_$v.json 2> $O/bench_1080p_$v.err
  echo "$v: $(python -c "import json;a=json.load(open('$O/bench_480p_$v.json'));b=json.load(open('$O/bench_1080p_$v.json'));print('480p %.1f fps conv frac %.3f | 1080p %.2f fps conv frac %.3f' % (a['value'],a['roofline']['frac'],b['value'],b['roofline']['frac']))")"
done
unset OTVM_HIP_LIB
OTVM_GRAPHS=0 timeout 300 python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_480p_default_nographs.json 2>/dev/null
python -c "import json;a=json.load(open('$O/bench_480p_default_nographs.json'));print('480p direct launches %.1f fps host %.2f ms' % (a['value'], a['host_issue_ms_per_frame']))"
