#!/bin/bash
# round 5, lease m: ablation of the 16-wide head-carrying conv alone on the device (tools/head_bench.py, variants of
# conv_head16_f16x3.hip built by tools/build_variant.sh with the OTVM_H16_* probes); Gram / gn_predict launch times by chunking
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05m; mkdir -p $O
cd $R
V=otvm_amd/csrc/build/variants
for n in "" nopatch now nomfma nohead nostore noall wgs2 wgs4 ""; do
  lib=""; [ -n "$n" ] && lib=$R/$V/libotvm_h16_$n.so
  echo "--- variant: ${n:-shipped}" | tee -a $O/head_bench.txt
  OTVM_HIP_LIB=$lib python tools/head_bench.py --iters 30 2>&1 | grep -v amdgpu | tee -a $O/head_bench.txt
done
for e in "X=0" "OTVM_GRAM_WGS=192" "OTVM_GRAM_WGS=96" "OTVM_GRAM_WGS=768" "OTVM_GRAM_BS=256"; do
  echo "--- $e" | tee -a $O/gram_timing.txt
  env $e python -m pytest tests/test_gpu_kernels.py -q -s -m gpu -k "test_gn_predict_matches_accumulated_statistics and 1-" 2>&1 | grep -E "timing|passed|failed" | tee -a $O/gram_timing.txt
done
