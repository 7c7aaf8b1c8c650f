#!/bin/bash
# round 5, lease zn: upsample2x with four input rows per workgroup (the lower row of a cell carried in registers) against one
# (OTVM_UPSAMPLE2X_ROWS=1): kernel tests, the kernel alone, the frame alternating; the tile-walk bit-identity test
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zn; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "upsample or kernel_fuzz or tile_walk or resample" > $O/tests.log 2>&1; echo "kernel tests rc=$?"; tail -2 $O/tests.log
for i in 1 2; do for m in 1 0; do
  OTVM_UPSAMPLE2X_ROWS=$m python tools/upsample_bench.py 2>&1 | grep -v amdgpu | sed "s/^/ROWS=$m  /" | tee -a $O/upsample.txt
done; done
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_tune.json 2> $O/bench_tune.err
for i in 1 2 3; do for m in 1 0; do
  OTVM_UPSAMPLE2X_ROWS=$m python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTVM_UPSAMPLE2X_ROWS=$m 1080p', round(d['value'],2), 'frames/s')" | tee -a $O/frame.txt
done; done
