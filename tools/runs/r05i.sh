#!/bin/bash
# round 5, lease i: how many host threads the CPU oracle should use on the GPU box (128 hardware threads: the default = all of them
# spends a third of its time in the kernel), + lease h (f16 mode, head16)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/r05i; mkdir -p $O
cd $R
python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/oracle_threads.txt
import os, time, torch, numpy as np
from otvm_amd.synth_weights import synthetic_state_dict
from otvm_amd.synth_data import synthetic_clip
from oracle.otvm_oracle import OtvmOracle
print("cpus", os.cpu_count(), "torch default threads", torch.get_num_threads())
sd = synthetic_state_dict(0)
H, W = 1080, 1920
frames, tri = synthetic_clip(H, W, 1, seed=3)
fg = torch.from_numpy(frames[0].astype(np.float32)).permute(2, 0, 1)[None, None].contiguous()
a, tg = torch.ones(1, 1, 1, H, W), torch.from_numpy(tri)[None, None]
for n in (128, 64, 32, 16):
    torch.set_num_threads(n)
    orc = OtvmOracle(sd, dilate_kernel=12, threads=n)
    t0 = time.perf_counter()
    orc.frame(a, fg, fg.clone(), tri_gt=tg, frame_id=0, first_frame=True, last_frame=True, memorize=False, max_memory_num=5)
    print("threads %3d: first frame (no memorize) %.1f s" % (n, time.perf_counter() - t0))
PY
bash tools/runs/r05h.sh
