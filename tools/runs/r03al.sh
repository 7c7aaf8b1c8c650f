#!/bin/bash
cd /root/repo
O=gpurun_out/r03al; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frame.py -q -x -m gpu > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
export OTVM_TUNE_FILE=/tmp/tune_r03al.json
for rep in 1 2; do
for v in new head; do
  if [ $v = new ]; then unset OTVM_FUSE_GN_TABLE; else export OTVM_FUSE_GN_TABLE=0; fi
  timeout 900 python bench.py --steps 97 --warmup 3 --no-cpu-baseline > $O/bench_1080p_${v}_$rep.json 2> $O/bench_1080p_${v}_$rep.err
  timeout 600 python bench.py --height 480 --width 832 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_480p_${v}_$rep.json 2> $O/bench_480p_${v}_$rep.err
done
done
