#!/bin/bash
# round 5, lease zr: XCD-aware (query tile, chunk) walk of the memory-read grid (OTVM_MEMREAD_WALK = 1 / 0): kernel tests, the read
# alone (tools/memread_bench.py), FETCH_SIZE of the read in both walks, 1080p frame alternating, 4K T = 200 growing bank
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zr; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "memory" > $O/tests.log 2>&1; echo "kernel tests (walk 1) rc=$?"; tail -1 $O/tests.log
CASES="--case 5,68,120 --case 1,68,120 --case 4,68,120 --case 5,30,52 --case 16,68,120 --case 8,136,240 --case 40,136,240"
for i in 1 2; do for m in 0 1; do
  OTVM_MEMREAD_WALK=$m python tools/memread_bench.py --iters 20 $CASES 2>&1 | grep -v amdgpu | sed "s/^/WALK=$m  /" | tee -a $O/memread.txt
done; done
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  OTVM_MEMREAD_WALK=$m rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch$m -o f -- python $R/tools/memread_bench.py --iters 3 --warm 0 $CASES > $O/fetch$m.log 2>&1
  echo "== OTVM_MEMREAD_WALK=$m (written column = the fetch pass again: ignore)" | tee -a $O/fetch.md
  python $R/tools/pmc_layer_traffic.py $O/fetch$m $O/fetch$m memory_read_f16x3_kernel | tee -a $O/fetch.md
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*counter_collection.csv" -delete
cd $R
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for i in 1 2 3; do for m in 0 1; do
  OTVM_MEMREAD_WALK=$m python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTVM_MEMREAD_WALK=$m 1080p', round(d['value'],2), 'frames/s', d.get('alpha_checksum'))" | tee -a $O/frame.txt
done; done
unset OTVM_TUNE_FILE
for m in 0 1; do
  OTVM_MEMREAD_WALK=$m python bench.py --height 2160 --width 3840 --steps 197 --warmup 3 --stress-bank --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTVM_MEMREAD_WALK=$m 4K T=200 growing bank', round(d['value'],3), 'frames/s', d.get('alpha_checksum'))" | tee -a $O/frame.txt
done
