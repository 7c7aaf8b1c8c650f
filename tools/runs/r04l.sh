#!/bin/bash
# PPM chain on an auxiliary stream beside conv_up1.0: frame tests (graphs, hazards, determinism) + A/B
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04l; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_frame.py -x -q -m gpu > $O/ftests.log 2>&1; tail -3 $O/ftests.log
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "reproducible or 480p_sequence or two_frames" > $O/fstests.log 2>&1; tail -3 $O/fstests.log
export OTVM_TUNE_FILE=$O/tune.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for v in "OTVM_PPM_FORK=1" "OTVM_PPM_FORK=0" "OTVM_PPM_FORK=1" "OTVM_PPM_FORK=0"; do
  env $v python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err
  echo $v; head -c 100 $O/bench_$v.json; echo
done
python tools/race_stress.py --reps 6 > $O/race.log 2>&1; tail -4 $O/race.log
