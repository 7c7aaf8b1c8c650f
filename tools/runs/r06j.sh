#!/bin/bash
# lease r06j: SQ counters and per-launch time of the 64-filter nine-tap patch tile, lean staging (default) against the round-5 staging
# (A/B build -DOTVM_PATCH_NO_LEAN=1), alone on the device: rocprofv3 --pmc <4 counters> --kernel-trace, one pass per counter group
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06j; O=$PWD/gpurun_out/r06j; R=$PWD
SH="--shape 64,64,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960"
NL=$R/otvm_amd/variants/libotvm_nolean.so
{ for rep in 1 2 3; do
  python tools/conv_bench.py --iters 40 $SH 2>&1 | grep -v amdgpu | sed 's/^/lean staging      /'
  OTVM_HIP_LIB=$NL python tools/conv_bench.py --iters 40 $SH 2>&1 | grep -v amdgpu | sed 's/^/round-5 staging   /'
  python tools/conv_bench.py --iters 40 --gn 1 --res 1 --shape 64,64,3,1,1,1088,1920 2>&1 | grep -v amdgpu | sed 's/^/lean staging, fused GN sums + residual      /'
  OTVM_HIP_LIB=$NL python tools/conv_bench.py --iters 40 --gn 1 --res 1 --shape 64,64,3,1,1,1088,1920 2>&1 | grep -v amdgpu | sed 's/^/round-5 staging, fused GN sums + residual   /'
done; } > $O/times.txt
cat $O/times.txt
cd /tmp; export TMPDIR=/tmp
G1="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES"; G2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
G3="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"; G4="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16"
G5="SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
for form in lean nolean; do
  [ $form = nolean ] && export OTVM_HIP_LIB=$NL
  echo "## $form" >> $O/sq.md
  i=0; for g in "$G1" "$G2" "$G3" "$G4" "$G5"; do i=$((i+1))
    rocprofv3 --pmc $g --kernel-trace --output-format csv -d $O/pmc_${form}_$i -o p -- python $R/tools/conv_bench.py --iters 5 $SH > $O/pmc_${form}_$i.log 2>&1
    python $R/tools/pmc_table.py $O/pmc_${form}_$i --top 1 | grep -v "^|---" >> $O/sq.md
  done
done
unset OTVM_HIP_LIB
find $O -name "*.csv" -delete; find $O -type d -empty -delete
cat $O/sq.md | cut -c1-200
