#!/bin/bash
# round-2 batch G (GPU box): stem kernel, GN folds at full size, checksums
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(python -m pytest tests/test_gpu_kernels.py -q 2>&1 | tail -25) > gpurun_out/r02g_kernels.log
(python -m pytest tests/test_gpu_fullsize.py -q -s -k "two_frames or steady or 480p" 2>&1 | grep -E "frame|passed|failed|Error|assert" | tail -30) > gpurun_out/r02g_fullsize.log
python tools/conv_bench.py --tune all --bias 1 --shape 4,64,7,2,1,1088,1920 --shape 12,64,7,2,1,1088,1920 --shape 24,64,7,2,1,1088,1920 > gpurun_out/exp_r02g.log 2>&1
OTVM_FUSE_GN_APPLY=0 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_r02g_nofuse.json 2>/dev/null
python bench.py --layer-report gpurun_out/layers_r02g.json --tune-report gpurun_out/tune_r02g.json > gpurun_out/bench_r02g.json 2>gpurun_out/bench_r02g.err
tail -6 gpurun_out/r02g_kernels.log; cat gpurun_out/r02g_fullsize.log
grep -v amdgpu gpurun_out/exp_r02g.log
python - <<'PY'
import json
for f in ("_nofuse",""):
    try:
        d=json.load(open("gpurun_out/bench_r02g%s.json"%f)); print(f or "tuned", round(d["value"],2), round(d["ms_per_step"],3), d["alpha_checksum"], d.get("roofline",{}).get("frac"), d.get("cpu_baseline",{}).get("alpha_maxabs_hip_vs_cpu_same_frame"))
    except Exception as e: print(f, "failed", e)
PY
tail -3 gpurun_out/bench_r02g.err
