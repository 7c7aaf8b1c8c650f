#!/bin/bash
# round-2 experiment batch D (GPU box): fp32 fast paths restored, autotuner on/off, HL8 opt-in, vs the pre-round tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(python -m pytest tests/test_gpu_kernels.py -q 2>&1 | tail -15) > gpurun_out/r02d_kernels.log
(python -m pytest tests/test_gpu_frame.py -q 2>&1 | tail -15) > gpurun_out/r02d_frame.log
(OTVM_HL8=1 python -m pytest tests/test_gpu_frame.py -q -k "sequence or fuzz" 2>&1 | tail -8) > gpurun_out/r02d_frame_hl8.log
(cd _old && python bench.py --no-cpu-baseline --no-roofline) > gpurun_out/bench_r02d_old.json 2>/dev/null
OTVM_AUTOTUNE=0 python bench.py --no-cpu-baseline --layer-report gpurun_out/layers_r02d_notune.json > gpurun_out/bench_r02d_notune.json 2>/dev/null
OTVM_AUTOTUNE=1 python bench.py --no-cpu-baseline --layer-report gpurun_out/layers_r02d_tune.json --tune-report gpurun_out/tune_r02d_1080.json > gpurun_out/bench_r02d_tune.json 2>gpurun_out/bench_r02d_tune.err
OTVM_AUTOTUNE=1 OTVM_HL8=1 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_r02d_tune_hl8.json 2>/dev/null
(cd _old && python bench.py --no-cpu-baseline --no-roofline --height 480 --width 832 --steps 47 --warmup 3) > gpurun_out/bench_r02d_old480.json 2>/dev/null
OTVM_AUTOTUNE=0 python bench.py --no-cpu-baseline --no-roofline --height 480 --width 832 --steps 47 --warmup 3 > gpurun_out/bench_r02d_notune480.json 2>/dev/null
OTVM_AUTOTUNE=1 python bench.py --no-cpu-baseline --no-roofline --height 480 --width 832 --steps 47 --warmup 3 --tune-report gpurun_out/tune_r02d_480.json > gpurun_out/bench_r02d_tune480.json 2>gpurun_out/bench_r02d_tune480.err
tail -4 gpurun_out/r02d_kernels.log; tail -4 gpurun_out/r02d_frame.log; tail -4 gpurun_out/r02d_frame_hl8.log
python - <<'PY'
import json
for f in ("old","notune","tune","tune_hl8","old480","notune480","tune480"):
    try:
        d=json.load(open("gpurun_out/bench_r02d_%s.json"%f)); print(f, round(d["value"],2), round(d["ms_per_step"],3), d["alpha_checksum"])
    except Exception as e: print(f, "failed", e)
PY
tail -3 gpurun_out/bench_r02d_tune.err
