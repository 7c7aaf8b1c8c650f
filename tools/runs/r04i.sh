#!/bin/bash
# fused head conv: kernel tests, frame tests (incl. training forward), A/B bench
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "head or glue or conv_input" > $O/ktests.log 2>&1
echo "ktests rc $?" >> $O/ktests.log; tail -3 $O/ktests.log
python -m pytest tests/test_gpu_frame.py tests/test_gpu_train.py -x -q -m gpu > $O/ftests.log 2>&1; tail -3 $O/ftests.log
for v in "OTVM_FUSE_HEAD=1" "OTVM_FUSE_HEAD=0" "OTVM_FUSE_HEAD=1"; do
  env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err
  echo $v; head -c 100 $O/bench_$v.json; echo
done
(cd _old && python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_old.json 2> $O/bench_old.err); head -c 100 $O/bench_old.json; echo
python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_480.json 2>$O/bench_480.err; head -c 100 $O/bench_480.json; echo
(cd _old && python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_old480.json 2> $O/bench_old480.err); head -c 100 $O/bench_old480.json; echo
