"""Soak run of the frame path (GPU only): a long 1080p clip (memory every 5, max 5 slots) -- throughput stays flat, device memory
does not grow, alpha stays finite, the range guard stays quiet.

    python tools/soak.py [--frames 2000] [--height 1080] [--width 1920]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                        # noqa: E402
from otvm_amd.synth_data import disc_trimap         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    H, W, T, N = args.height, args.width, 50, args.frames
    frames = bench.device_clip(H, W, T, seed=5, dev=dev)
    tri = torch.from_numpy(disc_trimap(H, W))[None, None].to(dev)
    a = torch.ones(1, 1, 1, H, W, device=dev)
    t0 = time.perf_counter()
    last = t0
    for i in range(N):
        f = frames[i % T if (i // T) % 2 == 0 else T - 1 - (i % T)]          # the clip played forwards and backwards
        out = model(a, f, f, tri=None, tri_gt=tri, large_input=False, _inputs_ready=True, first_frame=(i == 0),
                    last_frame=(i == N - 1), memorize=(i % 5 == 0), max_memory_num=5)
        if (i + 1) % 250 == 0:
            torch.cuda.synchronize()
            now = time.perf_counter()
            print("frames %4d..%4d: %.2f frames/s, device memory allocated %.2f GB reserved %.2f GB, alpha mean %.6f, finite %s"
                  % (i - 249, i, 250 / (now - last), torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
                     float(out[3].mean()), bool(torch.isfinite(out[3]).all())), flush=True)
            last = now
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("soak %dx%d: %d frames in %.1f s = %.2f frames/s; bank %s" % (W, H, N, dt, N / dt, model.memories["frames"]))


if __name__ == "__main__":
    main()
