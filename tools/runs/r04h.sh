#!/bin/bash
# state check: all kernel + frame tests, isolated glue, bench (prediction / refine tail off by default) vs the round-3 tree, late ev_dec A/B
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s > $O/ktests.log 2>&1
echo "ktests rc $?" >> $O/ktests.log; grep "timing" $O/ktests.log | head -12; tail -3 $O/ktests.log
python tools/glue_bench.py > $O/glue_new.json 2> $O/glue_new.err; cat $O/glue_new.json
python -m pytest tests/test_gpu_frame.py -x -q -m gpu > $O/ftests.log 2>&1; tail -3 $O/ftests.log
for v in "OTVM_EVDEC_LATE=1" "OTVM_EVDEC_LATE=0" "OTVM_GN_PREDICT=1"; do
  env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err
  echo $v; head -c 100 $O/bench_$v.json; echo
done
(cd _old && python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_old.json 2> $O/bench_old.err); head -c 100 $O/bench_old.json; echo
env OTVM_EVDEC_LATE=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_late2.json 2>/dev/null; head -c 100 $O/bench_late2.json; echo
