#!/bin/bash
# round 5, lease n: the 16-wide head-carrying conv with its epilogue's global loads hoisted (head weights through LDS, RGB and
# weights requested at the top of the kernel): kernel tests, then tools/head_bench.py (shipped + probes); Gram / gn_predict
# launch times by chunking (the timing lines of test_gn_predict_matches_accumulated_statistics)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05n; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "head or fba_fusion or reference_vectors" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
V=otvm_amd/csrc/build/variants
for n in "" nohead nopatch noall ""; do
  lib=""; [ -n "$n" ] && lib=$R/$V/libotvm_h16_$n.so
  echo "--- variant: ${n:-shipped}" | tee -a $O/head_bench.txt
  OTVM_HIP_LIB=$lib python tools/head_bench.py --iters 30 2>&1 | grep -v amdgpu | tee -a $O/head_bench.txt
done
for e in "X=0" "OTVM_GRAM_WGS=192" "OTVM_GRAM_WGS=96" "OTVM_GRAM_WGS=768" "OTVM_GRAM_BS=256"; do
  echo "--- $e" | tee -a $O/gram_timing.txt
  env $e python -m pytest tests/test_gpu_kernels.py -q -s -m gpu -k "test_gn_predict_matches_accumulated_statistics" 2>&1 | grep -E "timing|passed|failed" | tee -a $O/gram_timing.txt
done
