#!/bin/bash
# which SQ / LDS counters exist on this chip, and a first pass of wait / LDS-conflict counters over the bench
cd /root/repo
O=$PWD/gpurun_out/r03ad; mkdir -p $O
export OTVM_TUNE_FILE=$O/tune.json
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > $O/warm.json 2> $O/warm.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
CMD="python /root/repo/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $O/lds -o l -- $CMD > $O/lds.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/wait -o w -- $CMD > $O/wait.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/insts -o i -- $CMD > $O/insts.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/active -o a -- $CMD > $O/active.log 2>&1
find $O -name "*kernel_trace.csv" -delete
ls -la $O/*/ | head -40
tail -3 $O/lds.log $O/wait.log $O/insts.log $O/active.log
