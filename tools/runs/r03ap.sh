#!/bin/bash
cd /root/repo
O=gpurun_out/r03ap; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_frame.py -q -x -m gpu > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
export OTVM_TUNE_FILE=/tmp/tune_r03ap.json
for rep in 1 2; do
for v in new kxk; do
  if [ $v = new ]; then unset OTVM_FUSE_GN_APPLY_IGEMM_KXK; else export OTVM_FUSE_GN_APPLY_IGEMM_KXK=1; fi
  timeout 900 python bench.py --steps 97 --warmup 3 --no-cpu-baseline > $O/bench_1080p_${v}_$rep.json 2> $O/bench_1080p_${v}_$rep.err
  timeout 600 python bench.py --height 480 --width 832 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_480p_${v}_$rep.json 2> $O/bench_480p_${v}_$rep.err
done
done
