#!/bin/bash
# round 5, lease t: the pixel threshold of the predicted GroupNorm tail at 832x480
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05t; mkdir -p $O
for i in 1 2; do for m in 16384 4096 0; do OTVM_GN_PREDICT_MIN_PIXELS=$m python bench.py --height 480 --width 832 --steps 97 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('480p OTVM_GN_PREDICT_MIN_PIXELS=$m', round(d['value'],2), 'frames/s')" | tee -a $O/ab.txt; done; done
