#!/bin/bash
# round 5, lease h: the labelled single-pass "f16" mode (kernel + frame tests, bench with its error block), the 16-wide head tile
# at three workgroups per CU with buffer loads
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -s -k "single_pass or head_epilogue or every_tunable" > $O/tests_k.log 2>&1; echo "kernel tests rc=$?" | tee -a $O/tests_k.log
grep -E "f16x3 .* f16 |passed|failed|Error|assert" $O/tests_k.log | tail -20
timeout 900 python -m pytest tests/test_gpu_frame.py -x -q -m gpu -s -k "f16_mode" > $O/tests_f.log 2>&1; echo "frame tests rc=$?" | tee -a $O/tests_f.log
grep -E "f16 vs|passed|failed|Error|assert" $O/tests_f.log | tail -12
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --layer-report $O/layers_f16x3.json > $O/bench_f16x3.json 2> $O/bench_f16x3.err; head -c 200 $O/bench_f16x3.json; echo
python bench.py --precision f16 --steps 40 --warmup 5 --no-cpu-baseline --layer-report $O/layers_f16.json > $O/bench_f16.json 2> $O/bench_f16.err; tail -3 $O/bench_f16.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05h/bench_f16.json"))
print({k: d[k] for k in ("value","ms_per_step","dtype")}); print(d.get("alpha_error_vs_f16x3")); r=d.get("roofline",{}); print({k:r.get(k) for k in ("achieved","peak","frac","conv_ms_per_frame")})
PY
python - <<'PY'
import json
for f in ("layers_f16x3.json", "layers_f16.json"):
    rows = json.load(open("gpurun_out/r05h/" + f))
    print(f, [(r["layer"][-28:], round(r["ms_per_frame"], 3)) for r in rows if "head" in r["layer"] or r["layer"].endswith("conv_up1.0.main") or r["layer"].endswith("layer1.conv1") or r["layer"].endswith("RF2.convFS")])
PY
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | head -c 150; echo
