#!/bin/bash
# round 5, lease p: where do the waves of the 64-filter full-resolution patch tile spend their cycles?  SQ counters (+ instruction
# cache, if the part exposes it) on tools/conv_bench.py for the 64 -> 64 layer and, for comparison, the 320 -> 64 and 256 -> 256 layers
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_all.txt 2>&1
grep -i -o -E "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_CACHE|DCACHE)[A-Z0-9_]*)\b" $O/counters_all.txt | sort -u | tr '\n' ' ' | tee $O/icache_counters.txt; echo
CMD="python $R/tools/conv_bench.py --iters 5 --shape 64,64,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 256,256,3,1,1,272,480"
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -o p -- $CMD > $O/$n.log 2>&1; python $R/tools/pmc_table.py $O/$n --top 6 | grep -E "kernel|---|conv_" | cut -c1-260; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
run c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run d SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run e SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_IFETCH
IC=$(cat $O/icache_counters.txt | tr ' ' '\n' | grep -E "ICACHE" | head -4 | tr '\n' ' ')
[ -n "$IC" ] && run f $IC
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
