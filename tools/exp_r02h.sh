#!/bin/bash
# round-2 batch H (GPU box): gn_apply restored, folds fixed, early query encoder; vs the round-1 tree (_old/ = 8cc7866)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(python -m pytest tests/test_gpu_kernels.py tests/test_gpu_frame.py -q 2>&1 | tail -25) > gpurun_out/r02h_tests.log
(python -m pytest tests/test_gpu_fullsize.py -q -s -k "two_frames or steady or 480p" 2>&1 | grep -E "frame|passed|failed|Error|assert" | tail -30) > gpurun_out/r02h_fullsize.log
(cd _old && python bench.py --no-cpu-baseline --no-roofline) > gpurun_out/bench_r02h_old.json 2>/dev/null
OTVM_SIDE_STREAM=0 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_r02h_noside.json 2>/dev/null
python bench.py --layer-report gpurun_out/layers_r02h.json --tune-report gpurun_out/tune_r02h.json > gpurun_out/bench_r02h.json 2>gpurun_out/bench_r02h.err
(cd _old && python bench.py --no-cpu-baseline --no-roofline --height 480 --width 832 --steps 47 --warmup 3) > gpurun_out/bench_r02h_old480.json 2>/dev/null
python bench.py --no-cpu-baseline --no-roofline --height 480 --width 832 --steps 47 --warmup 3 > gpurun_out/bench_r02h_480.json 2>/dev/null
tail -8 gpurun_out/r02h_tests.log; cat gpurun_out/r02h_fullsize.log
python - <<'PY'
import json
for f in ("_old","_noside","","_old480","_480"):
    try:
        d=json.load(open("gpurun_out/bench_r02h%s.json"%f)); print(f or "new", round(d["value"],2), round(d["ms_per_step"],3), d["alpha_checksum"], d.get("roofline",{}).get("frac"), d.get("cpu_baseline",{}).get("alpha_maxabs_hip_vs_cpu_same_frame"))
    except Exception as e: print(f, "failed", e)
PY
tail -3 gpurun_out/bench_r02h.err
