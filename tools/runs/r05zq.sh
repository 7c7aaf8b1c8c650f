#!/bin/bash
# round 5, lease zq: split-K with the finish inside the conv launch (last-ticket workgroup adds the partials; OTVM_SPLITK_FUSED=1,
# experiment): kernel tests, fuzz, tuner cross-check and race screen with it on; 480p / 1080p frame rates alternating (own tune files)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zq; mkdir -p $O
cd $R
OTVM_SPLITK_FUSED=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "split or conv" > $O/tests.log 2>&1; echo "kernel tests (fused) rc=$?"; tail -2 $O/tests.log
for s in 71 72; do OTVM_SPLITK_FUSED=1 timeout 600 python tools/conv_fuzz.py --n 300 --seed $s 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/FUSED=1 conv_fuzz seed $s: /" | tee -a $O/fuzz.txt; done
OTVM_SPLITK_FUSED=1 timeout 900 python tools/tune_verify.py --height 480 --width 832 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/FUSED=1 /" | tee -a $O/fuzz.txt
for i in 1 2 3; do for m in 0 1; do
  OTVM_TUNE_FILE=$O/tune480_$m.json OTVM_SPLITK_FUSED=$m python bench.py --height 480 --width 832 --steps 97 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTVM_SPLITK_FUSED=$m 832x480', round(d['value'],2), 'frames/s', 'checksum', d.get('alpha_checksum'))" | tee -a $O/frame.txt
done; done
for i in 1 2; do for m in 0 1; do
  OTVM_TUNE_FILE=$O/tune1080_$m.json OTVM_SPLITK_FUSED=$m python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTVM_SPLITK_FUSED=$m 1080p', round(d['value'],2), 'frames/s', 'checksum', d.get('alpha_checksum'))" | tee -a $O/frame.txt
done; done
OTVM_SPLITK_FUSED=1 timeout 600 python tools/race_stress.py --height 480 --width 832 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/FUSED=1 /" | tee -a $O/fuzz.txt
