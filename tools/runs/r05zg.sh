#!/bin/bash
# round 5, lease zg: the 64-filter patch tile's wave cycles in its two matrix-core forms (OTVM_PATCH_M16 = 1 / 0): the SQ counters of
# lease r05p on tools/conv_bench.py for the 64 -> 64 full-resolution layer and the 320 -> 64 layer
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05zg; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/conv_bench.py --iters 5 --shape 64,64,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960"
run() { m=$1; n=$2; shift; shift; OTVM_PATCH_M16=$m rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n$m -o p -- $CMD > $O/$n$m.log 2>&1; echo "OTVM_PATCH_M16=$m"; python $R/tools/pmc_table.py $O/$n$m --top 4 | grep -E "kernel|---|conv_patch" | cut -c1-260; }
for m in 1 0; do
  run $m a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
  run $m b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
  run $m c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA
  run $m d SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16
  run $m e SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
