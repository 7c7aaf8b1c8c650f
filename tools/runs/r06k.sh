#!/bin/bash
# lease r06k: s_setprio around the MFMA section of the nine-tap 16x16x32 patch tiles (A/B builds) -- per-launch times, then the frame
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06k; O=gpurun_out/r06k; R=$PWD
SH="--shape 64,64,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 80,64,3,1,1,1088,1920"
for rep in 1 2 3; do
  python tools/conv_bench.py --iters 40 $SH 2>&1 | grep -v amdgpu | sed 's/^/default  /'
  OTVM_HIP_LIB=$R/otvm_amd/variants/libotvm_prio1.so python tools/conv_bench.py --iters 40 $SH 2>&1 | grep -v amdgpu | sed 's/^/setprio 1/'
  OTVM_HIP_LIB=$R/otvm_amd/variants/libotvm_prio2.so python tools/conv_bench.py --iters 40 $SH 2>&1 | grep -v amdgpu | sed 's/^/setprio 2/'
done | tee $O/times.txt
for rep in 1 2; do for v in "" prio1; do
  L=""; [ -n "$v" ] && L=$R/otvm_amd/variants/libotvm_$v.so
  OTVM_HIP_LIB=$L OTVM_BENCH_LIVE_PMC=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p ${v:-default}', round(d['value'],2), 'frames/s')"
done; done | tee $O/frame.txt
