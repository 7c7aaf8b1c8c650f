#!/bin/bash
# weight stages of the 256-channel patch tiles by LDS-DMA into two buffers: tests, race screen, same-box A/B
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04x; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu -k "conv or linearity or fuzz or reproducible or 1080p_two" > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
timeout 900 python tools/conv_fuzz.py --n 400 --seed 7 2>&1 | tail -1
timeout 900 python tools/race_stress.py --frames 14 --reps 20 2>&1 | tail -1
timeout 900 python tools/tune_verify.py 2>&1 | tail -1
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for rep in 1 2 3; do
for v in 0 1; do
  OTVM_PATCH_WIDE_GLDS=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTVM_PATCH_WIDE_GLDS=$v', round(d['value'],2), 'frames/s', d['alpha_checksum'])" >> $O/ab.txt
done; done
cat $O/ab.txt
