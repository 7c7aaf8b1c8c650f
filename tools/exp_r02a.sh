#!/bin/bash
# round-2 experiment batch A (GPU box): full gpu tests, hl8 staging hack timing, short-K 1x1 tile overrides, kernel stats
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/r02_tests2.log
BIG="--shape 2048,512,1,1,1,136,240 --shape 1024,256,1,1,1,136,240 --shape 512,512,3,1,4,136,240 --shape 256,256,3,1,2,136,240 --shape 3072,256,3,1,1,136,240 --shape 256,256,3,1,1,68,120 --shape 1024,512,3,1,1,68,120"
SHORT="--shape 64,256,1,1,1,272,480 --shape 128,512,1,1,1,136,240 --shape 256,1024,1,1,1,136,240 --shape 256,1024,1,1,1,68,120 --shape 512,2048,1,1,1,136,240"
V=otvm_amd/csrc/build/variants/libotvm_hl8hack.so
{
echo "== base big"; python tools/conv_bench.py $BIG
echo "== hl8hack big"; OTVM_HIP_LIB=$V python tools/conv_bench.py $BIG
echo "== base short res"; python tools/conv_bench.py --res 1 --relu 0 $SHORT
echo "== hl8hack short res"; OTVM_HIP_LIB=$V python tools/conv_bench.py --res 1 $SHORT
echo "== base short gn"; python tools/conv_bench.py --gn 1 $SHORT
echo "== no256x256 short res"; OTVM_T_HUGE=100000000 python tools/conv_bench.py --res 1 $SHORT
echo "== 128x128 short res"; OTVM_T_HUGE=100000000 OTVM_T_BIG=100000000 python tools/conv_bench.py --res 1 $SHORT
echo "== 128x128 short gn"; OTVM_T_HUGE=100000000 OTVM_T_BIG=100000000 python tools/conv_bench.py --gn 1 $SHORT
echo "== 128x64 short res"; OTVM_T_HUGE=100000000 OTVM_T_BIG=100000000 OTVM_T_MID=100000000 python tools/conv_bench.py --res 1 $SHORT
} > gpurun_out/exp_r02a.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02a -o r02a -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_r02a.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof_r02a | head
tail -8 gpurun_out/r02_tests2.log
cat gpurun_out/exp_r02a.log
