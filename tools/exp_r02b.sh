#!/bin/bash
# round-2 experiment batch B (GPU box): HL8 correctness + timing
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(python -m pytest tests/test_gpu_kernels.py -q 2>&1 | tail -40) > gpurun_out/r02b_kernels.log
(python -m pytest tests/test_gpu_frame.py -q 2>&1 | tail -40) > gpurun_out/r02b_frame.log
OTVM_HL8=0 python bench.py --no-cpu-baseline --layer-report gpurun_out/layers_r02b_f32.json > gpurun_out/bench_r02b_f32.json 2> gpurun_out/bench_r02b_f32.err
OTVM_HL8=1 python bench.py --no-cpu-baseline --layer-report gpurun_out/layers_r02b_hl8.json > gpurun_out/bench_r02b_hl8.json 2> gpurun_out/bench_r02b_hl8.err
BIG="--shape 2048,512,1,1,1,136,240 --shape 512,512,3,1,4,136,240 --shape 256,256,3,1,1,272,480 --shape 64,64,3,1,1,1088,1920 --shape 64,256,1,1,1,272,480 --shape 256,64,1,1,1,272,480"
{
echo "== fp32 views"; python tools/conv_bench.py $BIG
echo "== hl8 in"; python tools/conv_bench.py --hl8 1,0,0 $BIG
echo "== hl8 in+out"; python tools/conv_bench.py --hl8 1,0,1 $BIG
echo "== fp32 res"; python tools/conv_bench.py --res 1 $BIG
echo "== hl8 in+res+out"; python tools/conv_bench.py --res 1 --hl8 1,1,1 $BIG
} > gpurun_out/exp_r02b.log 2>&1
tail -15 gpurun_out/r02b_kernels.log; tail -15 gpurun_out/r02b_frame.log
grep -v amdgpu gpurun_out/exp_r02b.log
python - <<'PY'
import json
for f in ("f32","hl8"):
    try:
        d=json.load(open("gpurun_out/bench_r02b_%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["conv_ms_per_frame"], d["alpha_checksum"])
    except Exception as e: print(f, "failed", e)
PY
tail -5 gpurun_out/bench_r02b_hl8.err
