#!/bin/bash
# rocprofv3 PMC passes over otvm_memory_read_f16x3 on a 4K-sized bank (hw = 136 x 240 = 32 640) at 50 / 100 / 200 slots
# (BASELINE configs[4]: the growing bank; north_star: "coalesced HBM reads of the growing bank, evidenced by rocprof HBM GB/s
# and MFMA-busy").  Output: one markdown table on stdout.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/memread_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
READS=2            # 1 warm-up read is excluded below by running --iters (READS) with 0 extra warm-ups: see --warm 0
echo "# otvm_memory_read_f16x3 on a 2176x3840 frame's bank (hw = 32 640 queries, slot = 83.6 MB as split fp16 in MFMA fragment order)"
echo "# rocprofv3 --pmc <counters> --kernel-trace -- python tools/memread_bench.py --case T,136,240 --iters $READS --warm 0 ; FETCH_SIZE x2 (gfx950), KiB -> bytes; sums over the memory_read_f16x3_kernel launches of one read (8 slots per launch)"
echo "| slots | bank GB | launches / read | ms / read (kernel time) | TFLOP/s (1280 T hw^2) | clock GHz | MFMA busy % of elapsed | MFMA busy % at 2.4 GHz | fetched GB / read | fetched / bank | written GB / read | L2-fabric GB/s |"
echo "|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"
for T in 50 100 200; do
  D=$O/T$T; rm -rf $D; mkdir -p $D
  CMD="python $R/tools/memread_bench.py --case $T,136,240 --iters $READS --warm 0"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D/mfma -o m -- $CMD > $D/mfma.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -o f -- $CMD > $D/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -o w -- $CMD > $D/write.log 2>&1
  python $R/tools/memread_pmc.py $D $T 136 240 $READS
  rm -rf $D/mfma/*/*kernel_trace.csv $D/fetch $D/write
done
