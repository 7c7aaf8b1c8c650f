#!/bin/bash
# gram v2 / predict v3 timings per shape; EDT probes; isolated per-kernel glue times; frame A/B
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s -k "gn_predict or input_norm_with_identity" > $O/ktests.log 2>&1
echo "ktests rc $?" >> $O/ktests.log; grep "gn_predict \|timing" $O/ktests.log | grep -A1 "passes 1" | head -40; tail -3 $O/ktests.log
for v in "OTVM_GRAM_BS=128" "OTVM_GRAM_WGS=96" "OTVM_GRAM_WGS=384"; do
  echo "== $v"; env $v python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s -k "gn_predict and 1" 2>&1 | grep "timing"
done
for d in 0 1 2 3 4 8 12 15 16 48 64; do
  echo "== OTVM_EDT_DBG=$d"; OTVM_EDT_DBG=$d python tools/glue_bench.py 2>/dev/null | grep trimap_encode
done
R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gl -o gl -- python $R/tools/glue_bench.py > $O/gl.log 2>&1
cd $R
KS=$(ls $O/gl/*kernel_stats.csv $O/gl/*/*kernel_stats.csv 2>/dev/null | head -1)
python tools/kernel_stats_md.py $KS 1 "python tools/glue_bench.py" > $O/glue_kernel_stats.md
find $O -name "*kernel_trace.csv" -delete
head -30 $O/glue_kernel_stats.md
python -m pytest tests/test_gpu_frame.py -x -q -m gpu -k "sequence_vs_oracle" > $O/ftests.log 2>&1; tail -3 $O/ftests.log
for v in "OTVM_GN_PREDICT=0" "OTVM_GN_PREDICT=1" "OTVM_FUSE_REFINE_TAIL=0"; do
  env $v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_$v.json 2> $O/bench_$v.err
  echo $v; head -c 100 $O/bench_$v.json; echo
done
(cd _old && python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_old.json 2> $O/bench_old.err); head -c 100 $O/bench_old.json; echo
