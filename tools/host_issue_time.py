"""Host-side cost of issuing one frame (python + ctypes launches, no synchronisation) next to the GPU frame time.

    python tools/host_issue_time.py [--height 480 --width 832] [--frames 60]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--frames", type=int, default=60)
    args = ap.parse_args()
    from otvm_amd.synth_data import disc_trimap
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    H, W, T = args.height, args.width, args.frames
    frames = bench.device_clip(H, W, T, seed=1, dev=dev)
    tri = torch.from_numpy(disc_trimap(H, W))[None, None].to(dev)
    a = torch.ones(1, 1, 1, H, W, device=dev)
    for t in range(5):
        model(a, frames[t], frames[t], tri_gt=tri, **bench.frame_kwargs(t, T, 5, 5))
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for t in range(5, T):
        c0 = time.perf_counter()
        model(a, frames[t], frames[t], tri_gt=tri, **bench.frame_kwargs(t, T, 5, 5))
        host += time.perf_counter() - c0
    issued = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    n = T - 5
    print("%dx%d: host issue %.2f ms/frame (loop returned after %.2f ms/frame), GPU-inclusive %.2f ms/frame"
          % (W, H, 1e3 * host / n, 1e3 * issued / n, 1e3 * total / n))


if __name__ == "__main__":
    main()
