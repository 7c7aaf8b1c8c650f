#!/bin/bash
# round-2 experiment batch C (GPU box): HL8 correctness after the relu fix + same-box A/B against the pre-HL8 tree (_old/)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(python -m pytest tests/test_gpu_kernels.py -q 2>&1 | tail -30) > gpurun_out/r02c_kernels.log
(python -m pytest tests/test_gpu_frame.py -q 2>&1 | tail -30) > gpurun_out/r02c_frame.log
SH="--shape 2048,512,1,1,1,136,240 --shape 512,512,3,1,4,136,240 --shape 3072,256,3,1,1,136,240 --shape 256,256,3,1,1,272,480 --shape 64,64,3,1,1,1088,1920 --shape 512,2048,1,1,1,136,240 --shape 64,256,1,1,1,272,480 --shape 256,256,3,1,1,68,120"
{
echo "== OLD tree"; (cd _old && python tools/conv_bench.py $SH)
echo "== NEW fp32 views"; python tools/conv_bench.py $SH
echo "== NEW hl8 in"; python tools/conv_bench.py --hl8 1,0,0 $SH
echo "== NEW hl8 in+out"; python tools/conv_bench.py --hl8 1,0,1 $SH
echo "== OLD tree res"; (cd _old && python tools/conv_bench.py --res 1 $SH)
echo "== NEW fp32 res"; python tools/conv_bench.py --res 1 $SH
echo "== NEW hl8 all res"; python tools/conv_bench.py --res 1 --hl8 1,1,1 $SH
echo "== OLD tree gn"; (cd _old && python tools/conv_bench.py --gn 1 $SH)
echo "== NEW fp32 gn"; python tools/conv_bench.py --gn 1 $SH
echo "== NEW hl8 in gn"; python tools/conv_bench.py --gn 1 --hl8 1,0,0 $SH
} > gpurun_out/exp_r02c.log 2>&1
(cd _old && python bench.py --no-cpu-baseline --no-roofline) > gpurun_out/bench_r02c_old.json 2>/dev/null
OTVM_HL8=0 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_r02c_f32.json 2>/dev/null
OTVM_HL8=1 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_r02c_hl8.json 2>/dev/null
tail -12 gpurun_out/r02c_kernels.log; tail -12 gpurun_out/r02c_frame.log
grep -v amdgpu gpurun_out/exp_r02c.log
python - <<'PY'
import json
for f in ("old","f32","hl8"):
    try:
        d=json.load(open("gpurun_out/bench_r02c_%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["alpha_checksum"])
    except Exception as e: print(f, "failed", e)
PY
