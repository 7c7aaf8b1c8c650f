#!/bin/bash
# does the tuner's preference for K splits (timed alone on the device) hold inside the frame?
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04ad; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
i=0
for v in "OTVM_TUNE_SPLITK_MARGIN=0" "OTVM_TUNE_SPLITK_MARGIN=0.10" "OTVM_TUNE_SPLITK_MARGIN=0.10 OTVM_SPLITK=0" "OTVM_TUNE_SPLITK_MARGIN=0.5 OTVM_SPLITK=0" "OTVM_TUNE_SPLITK_MARGIN=9 OTVM_SPLITK=0"; do
  i=$((i+1))
  env $v OTVM_TUNE_FILE=$O/tune_cache_$i.json python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1080p $v', round(d['value'],2), 'frames/s')" >> $O/ab.txt
done; done
for rep in 1 2; do
i=0
for v in "OTVM_TUNE_SPLITK_MARGIN=0" "OTVM_TUNE_SPLITK_MARGIN=0.10" "OTVM_TUNE_SPLITK_MARGIN=0.10 OTVM_SPLITK=0" "OTVM_TUNE_SPLITK_MARGIN=0.5 OTVM_SPLITK=0"; do
  i=$((i+1))
  env $v OTVM_TUNE_FILE=$O/tune_cache480_$i.json python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p $v', round(d['value'],2), 'frames/s')" >> $O/ab.txt
done; done
sort $O/ab.txt
