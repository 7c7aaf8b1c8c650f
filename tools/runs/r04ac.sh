#!/bin/bash
# block heights of the 256-channel patch tile as autotuner candidates: tests, tune_verify, same-box A/B at 1080p and 480p
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04ac; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frame.py -x -q -m gpu -k "conv or tunable or fuzz or race_free or autotune or sequence_vs_oracle" > $O/tests.log 2>&1
echo "tests rc $?" >> $O/tests.log; tail -4 $O/tests.log
timeout 900 python tools/tune_verify.py 2>&1 | tail -1
timeout 900 python tools/tune_verify.py --height 480 --width 832 2>&1 | tail -1
for rep in 1 2 3; do
for v in 0 1; do
  OTVM_PATCH_HEIGHTS=$v OTVM_TUNE_FILE=$O/tune_cache_$v.json python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p OTVM_PATCH_HEIGHTS=$v', round(d['value'],2), 'frames/s')" >> $O/ab.txt
done; done
for rep in 1 2 3; do
for v in 0 1; do
  OTVM_PATCH_HEIGHTS=$v OTVM_TUNE_FILE=$O/tune_cache480_$v.json python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p OTVM_PATCH_HEIGHTS=$v', round(d['value'],2), 'frames/s')" >> $O/ab.txt
done; done
cat $O/ab.txt
python - <<'PY'
import json
for f in ('tune_cache_1.json','tune_cache480_1.json'):
    try:
        d=json.load(open('gpurun_out/r04ac/'+f))
        n=sum(1 for k,v in d.items() if isinstance(v,(int,)) and v//16-1==14 and v&15>1)
        print(f, 'entries', len(d), 'patch block heights chosen (tune codes 242..244):', [v for v in d.values() if isinstance(v,int) and v//16-1==14 and (v&15)>1])
    except Exception as e: print(f, e)
PY
