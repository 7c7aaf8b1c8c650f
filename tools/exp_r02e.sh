#!/bin/bash
# round-2 batch E (GPU box): full gpu test suite on the cleaned tree + bench vs the pre-round tree (_old/)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/r02e_tests.log
(cd _old && python bench.py --no-cpu-baseline --no-roofline) > gpurun_out/bench_r02e_old.json 2>/dev/null
OTVM_AUTOTUNE=0 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_r02e_notune.json 2>/dev/null
python bench.py --no-cpu-baseline --layer-report gpurun_out/layers_r02e.json > gpurun_out/bench_r02e.json 2>gpurun_out/bench_r02e.err
tail -8 gpurun_out/r02e_tests.log
python - <<'PY'
import json
for f in ("_old","_notune",""):
    try:
        d=json.load(open("gpurun_out/bench_r02e%s.json"%f)); print(f or "tuned", round(d["value"],2), round(d["ms_per_step"],3), d["alpha_checksum"], d.get("roofline",{}).get("frac"))
    except Exception as e: print(f, "failed", e)
PY
