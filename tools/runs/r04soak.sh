#!/bin/bash
# soak (round 4, final tree): 2000-frame 1080p clip (memory every 5, max 5): throughput stays flat, device memory does not grow, the range guard stays quiet
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04soak; mkdir -p $O
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r04soak/soak_1080p.txt
import time, torch, sys
sys.path.insert(0, '.')
import bench
from otvm_amd.synth_data import disc_trimap
dev = torch.device('cuda', 0)
model, _ = bench.build_model(dev)
H, W, T, N = 1080, 1920, 50, 2000
frames = bench.device_clip(H, W, T, seed=5, dev=dev)
tri = torch.from_numpy(disc_trimap(H, W))[None, None].to(dev)
a = torch.ones(1, 1, 1, H, W, device=dev)
t0 = time.perf_counter(); last = t0; chk = []
for i in range(N):
    f = frames[i % T if (i // T) % 2 == 0 else T - 1 - (i % T)]          # the clip played forwards and backwards
    out = model(a, f, f, tri=None, tri_gt=tri, large_input=False, _inputs_ready=True, first_frame=(i == 0), last_frame=(i == N - 1),
                memorize=(i % 5 == 0), max_memory_num=5)
    if (i + 1) % 250 == 0:
        torch.cuda.synchronize()
        now = time.perf_counter()
        print('frames %4d..%4d: %.2f frames/s, device memory allocated %.2f GB reserved %.2f GB, alpha mean %.6f, finite %s'
              % (i - 249, i, 250 / (now - last), torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
                 float(out[3].mean()), bool(torch.isfinite(out[3]).all())), flush=True)
        last = now
torch.cuda.synchronize()
print('soak 1080p: %d frames in %.1f s = %.2f frames/s; bank %s' % (N, time.perf_counter() - t0, N / (time.perf_counter() - t0), model.memories['frames']))
PY
