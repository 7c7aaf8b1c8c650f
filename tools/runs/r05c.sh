#!/bin/bash
# round 5, lease c: LDS-DMA weight stages on ALL implicit-GEMM tiles (tile 32 + t) + branch-free buffer loads in the patch kernels.
# kernel tests, per-shape A/B (staged t vs 32 + t), whole-frame bench 1080p / 480p against OTVM_IGEMM_GLDS=0 and the round-4 tree
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or igemm or patch or race or bottleneck" > $O/tests_conv.log 2>&1; echo "conv tests rc=$?" | tee -a $O/tests_conv.log
tail -3 $O/tests_conv.log
code() { echo $(( ($1 + 1) * 16 + ${2:-1} )); }
T="$(code 2),$(code 34),$(code 3),$(code 35),$(code 4),$(code 36),$(code 7),$(code 39),$(code 8),$(code 40),$(code 10),$(code 42),$(code 11),$(code 43)"
S="--shape 256,1024,1,1,1,136,240 --shape 1024,256,1,1,1,136,240 --shape 128,512,1,1,1,136,240 --shape 512,128,1,1,1,136,240 --shape 256,256,3,1,1,136,240 --shape 128,128,3,1,1,136,240 --shape 256,1024,1,1,1,68,120 --shape 1024,256,1,1,1,68,120 --shape 256,256,3,1,1,68,120 --shape 64,256,1,1,1,272,480 --shape 256,128,1,1,1,272,480"
timeout 900 python tools/conv_bench.py --iters 30 --tune $T $S > $O/conv_bench_glds_small.txt 2>&1
cat $O/conv_bench_glds_small.txt
timeout 600 python tools/conv_bench.py --iters 30 --tune 0 --shape 64,64,3,1,1,1088,1920 --shape 80,32,3,1,1,1088,1920 --shape 64,32,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 256,256,3,1,1,272,480 --shape 512,256,3,1,1,272,480 > $O/conv_bench_patch.txt 2>&1
(cd _old && timeout 600 python tools/conv_bench.py --iters 30 --tune 0 --shape 64,64,3,1,1,1088,1920 --shape 80,32,3,1,1,1088,1920 --shape 64,32,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 --shape 256,256,3,1,1,272,480 --shape 512,256,3,1,1,272,480 > $O/conv_bench_patch_r04tree.txt 2>&1)
paste -d'\n' $O/conv_bench_patch.txt $O/conv_bench_patch_r04tree.txt
for i in 1 2; do
timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --tune-report $O/tune_1080p.json > $O/bench_new.json 2> $O/bench_new.err; head -c 200 $O/bench_new.json; echo
OTVM_IGEMM_GLDS=0 timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | head -c 200; echo
(cd _old && timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | head -c 200); echo
done
timeout 600 python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --tune-report $O/tune_480p.json > $O/bench_480p.json 2>$O/bench_480p.err; head -c 200 $O/bench_480p.json; echo
(cd _old && timeout 600 python bench.py --height 480 --width 832 --steps 47 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | head -c 200); echo
