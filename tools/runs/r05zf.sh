#!/bin/bash
# round 5, lease zf: the memory read's P.V on v_mfma_f32_16x16x32_f16 over the M16 value layout (OTVM_MEMREAD_M16 = 1 default / 0):
# kernel + frame tests that touch the read, the read alone on the device, the whole frame by switch (1080p, 4K with a growing bank)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zf; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "memory_read or kernel_fuzz" > $O/tests.log 2>&1; echo "kernel tests rc=$?"; tail -3 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_frame.py -x -q -m gpu -k "sequence_vs_oracle" > $O/tests_frame.log 2>&1; echo "frame tests rc=$?"; tail -3 $O/tests_frame.log
for i in 1 2; do for m in 1 0; do
  OTVM_MEMREAD_M16=$m python tools/memread_bench.py --iters 30 --case 5,68,120 --case 5,30,52 --case 20,136,240 --case 1,68,120 2>&1 | grep -v amdgpu | sed "s/^/M16=$m  /" | tee -a $O/read_alone.txt
done; done
for m in 1 0; do OTVM_MEMREAD_M16=$m OTVM_TUNE_FILE=$O/tune.json python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1; done
for i in 1 2 3; do for m in 1 0; do
  OTVM_MEMREAD_M16=$m OTVM_TUNE_FILE=$O/tune.json python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p OTVM_MEMREAD_M16=$m', round(d['value'],2), 'frames/s; read', round(d['memory_read']['ms_per_launch'],4), 'ms per launch, frac', round(d['memory_read']['frac'],4))" | tee -a $O/ab.txt
done; done
for m in 1 0; do
  OTVM_MEMREAD_M16=$m python bench.py --height 2160 --width 3840 --steps 197 --warmup 3 --stress-bank --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4K T=200 growing bank OTVM_MEMREAD_M16=$m', round(d['value'],3), 'frames/s')" | tee -a $O/ab.txt
done
