#!/bin/bash
# lease r06e: lean staging of the nine-tap 16x16x32 patch tiles -- kernel tests, then the frame with and without (A/B build)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06e; O=gpurun_out/r06e
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "patch or nine_tap or input_norm or fused_gn or gn_apply" > $O/t_patch.log 2>&1; echo "patch rc $?" >> $O/t_patch.log
tail -3 $O/t_patch.log
timeout 600 python tools/conv_fuzz.py --n 200 --seed 21 --patch64 2>&1 | tail -2
NL=$PWD/otvm_amd/variants/libotvm_nolean.so
for rep in 1 2 3; do
  OTVM_BENCH_LIVE_PMC=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p lean staging     ', round(d['value'],2), 'frames/s')"
  OTVM_HIP_LIB=$NL OTVM_BENCH_LIVE_PMC=0 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p round-5 staging  ', round(d['value'],2), 'frames/s')"
done 2>&1 | tee $O/ab_1080p.txt
for rep in 1 2; do
  OTVM_BENCH_LIVE_PMC=0 python bench.py --height 480 --width 832 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p lean staging     ', round(d['value'],2), 'frames/s')"
  OTVM_HIP_LIB=$NL OTVM_BENCH_LIVE_PMC=0 python bench.py --height 480 --width 832 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p round-5 staging  ', round(d['value'],2), 'frames/s')"
done 2>&1 | tee $O/ab_480p.txt
