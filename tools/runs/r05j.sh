#!/bin/bash
# round 5, lease j: the memory read with ONE 512-thread workgroup per CU whose two wave groups alternate explicitly
# (memory_read_f16x3_alt_kernel, OTVM_MEMREAD_ALT=1) against the two-independent-workgroups kernel
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05j; mkdir -p $O
cd $R
OTVM_MEMREAD_ALT=1 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu -k "memory_read" > $O/tests_alt.log 2>&1; echo "memory-read tests (alt) rc=$?" | tee -a $O/tests_alt.log; tail -3 $O/tests_alt.log
python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/bitwise.txt
import ctypes as C, os, subprocess, sys, torch
code = r"""
import ctypes as C, torch, sys
from otvm_amd import lib as L
lib = L.load(); dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(5)
res = []
for (T, hw) in ((5, 8160), (3, 8160), (1, 1560), (7, 2000), (11, 8160)):
    q = torch.randn(hw, 128, device=dev, generator=g) * 0.8
    slots = []
    for t in range(T):
        k = torch.randn(hw, 128, device=dev, generator=g) * 0.8; v = torch.randn(hw, 512, device=dev, generator=g)
        sl = torch.zeros(int(lib.otvm_bank_slot_bytes_f16x3(hw)), dtype=torch.uint8, device=dev)
        L.check(lib.otvm_bank_pack_f16x3(k.data_ptr(), v.data_ptr(), hw, sl.data_ptr(), st)); slots.append(sl)
    sp = (C.c_void_p * T)(*[s.data_ptr() for s in slots])
    out = torch.empty(hw, 512, device=dev)
    ws = torch.empty(int(lib.otvm_memory_read_ws_bytes(hw, T)), dtype=torch.uint8, device=dev)
    L.check(lib.otvm_memory_read_f16x3(q.data_ptr(), 128, sp, T, hw, out.data_ptr(), 512, ws.data_ptr(), st)); torch.cuda.synchronize()
    res.append(out.cpu())
torch.save(res, sys.argv[1])
"""
for alt in ("0", "1"):
    env = dict(os.environ, OTVM_MEMREAD_ALT=alt)
    subprocess.run([sys.executable, "-c", code, "/tmp/mr%s.pt" % alt], env=env, check=True)
a, b = torch.load("/tmp/mr0.pt"), torch.load("/tmp/mr1.pt")
print("bit-identical outputs (two kernels, five bank shapes):", [bool(torch.equal(x, y)) for x, y in zip(a, b)], [float((x - y).abs().max()) for x, y in zip(a, b)])
PY
C="--case 5,68,120 --case 3,68,120 --case 1,68,120 --case 5,30,52 --case 20,68,120 --case 3,136,240 --case 8,136,240"
for alt in 0 1 0 1; do echo "--- OTVM_MEMREAD_ALT=$alt" | tee -a $O/memread_bench.txt; OTVM_MEMREAD_ALT=$alt python tools/memread_bench.py --iters 30 $C 2>&1 | grep -v amdgpu | tee -a $O/memread_bench.txt; done
export OTVM_TUNE_FILE=$O/tune_cache.json
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for alt in 0 1 0 1; do OTVM_MEMREAD_ALT=$alt python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OTVM_MEMREAD_ALT=$alt', round(d['value'],2), 'frames/s; memory read', {k: round(v, 4) if isinstance(v, float) else v for k, v in d.get('memory_read', {}).items() if k in ('ms_per_launch','frac','achieved')})" | tee -a $O/ab_memread.txt; done
unset OTVM_TUNE_FILE
for alt in 0 1; do OTVM_MEMREAD_ALT=$alt python bench.py --height 2160 --width 3840 --steps 57 --warmup 3 --stress-bank --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4K growing bank T=60, OTVM_MEMREAD_ALT=$alt', round(d['value'],3), 'frames/s')" | tee -a $O/ab_memread.txt; done
