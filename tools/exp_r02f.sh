#!/bin/bash
# round-2 batch F (GPU box): row-reuse patch variant, folded GroupNorm tables (upsample input, refine residual)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(python -m pytest tests/test_gpu_kernels.py -q 2>&1 | tail -25) > gpurun_out/r02f_kernels.log
(python -m pytest tests/test_gpu_frame.py -q 2>&1 | tail -25) > gpurun_out/r02f_frame.log
python tools/conv_bench.py --tune all --shape 64,64,3,1,1,1088,1920 --shape 80,64,3,1,1,1088,1920 --shape 64,32,3,1,1,1088,1920 --shape 32,16,3,1,1,1088,1920 --shape 320,64,3,1,1,544,960 > gpurun_out/exp_r02f.log 2>&1
(cd _old && python bench.py --no-cpu-baseline --no-roofline) > gpurun_out/bench_r02f_old.json 2>/dev/null
OTVM_FUSE_GN_APPLY=0 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_r02f_nofuse.json 2>/dev/null
python bench.py --no-cpu-baseline --layer-report gpurun_out/layers_r02f.json --tune-report gpurun_out/tune_r02f.json > gpurun_out/bench_r02f.json 2>gpurun_out/bench_r02f.err
tail -6 gpurun_out/r02f_kernels.log; tail -6 gpurun_out/r02f_frame.log
grep -v amdgpu gpurun_out/exp_r02f.log
python - <<'PY'
import json
for f in ("_old","_nofuse",""):
    try:
        d=json.load(open("gpurun_out/bench_r02f%s.json"%f)); print(f or "tuned", round(d["value"],2), round(d["ms_per_step"],3), d["alpha_checksum"], d.get("roofline",{}).get("frac"))
    except Exception as e: print(f, "failed", e)
PY
tail -3 gpurun_out/bench_r02f.err
