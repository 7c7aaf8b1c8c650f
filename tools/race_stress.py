"""Determinism stress: the same clip matted over and over in one process (fixed kernel configurations), host running ahead of
the device, with and without the input-ready hint -- every repetition must reproduce the first one bit for bit.  A stream
hazard (a buffer read on one stream while another stream rewrites it) shows up as a mismatch.

    python tools/race_stress.py [--height 1080 --width 1920] [--frames 14] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="sequences stepped in lock-step (EvalModel.forward_batch)")
    ap.add_argument("--idle-seconds", type=float, default=0.0, help="host sleeps this long before the --flush-at frame (device idle, clocks down), as bench.py's CPU oracle does")
    ap.add_argument("--flush-at", type=int, default=-1, help="bench.py's cross-check flow: engine.flush() + device sync before this frame")
    args = ap.parse_args()
    from otvm_amd.synth_data import disc_trimap
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    H, W, T = args.height, args.width, args.frames
    frames = bench.device_clip(H, W, T, seed=3, dev=dev)
    NB = max(1, args.batch)
    clips = [frames] + [bench.device_clip(H, W, T, seed=3 + 10 * b, dev=dev) for b in range(1, NB)]
    tri = torch.from_numpy(disc_trimap(H, W))[None, None].to(dev)
    a = torch.ones(1, 1, 1, H, W, device=dev)

    def clip(ready, sync):
        outs = []
        for t in range(T):
            if t == args.flush_at:
                model._engine.flush()
                torch.cuda.synchronize()
                if args.idle_seconds > 0 and not sync:
                    import time
                    time.sleep(args.idle_seconds)
            if NB == 1:
                o = model(a, frames[t], frames[t], tri=None, tri_gt=tri, large_input=False,
                          _inputs_ready=(None if t == args.flush_at else ready), **bench.frame_kwargs(t, T, 5, 5))
                outs.append((o[3], o[1]))
            else:
                fr = [c[t] for c in clips]
                ob = model.forward_batch([a] * NB, fr, fr, [tri] * NB, large_input=False,
                                         _inputs_ready=(None if t == args.flush_at else ready), **bench.frame_kwargs(t, T, 5, 5))
                outs.append((torch.stack([o[3] for o in ob]), torch.stack([o[1] for o in ob])))
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return outs
    ref = clip(None, True)                                   # device sync after every frame: no overlap across frames
    bad = 0
    for r in range(args.reps):
        for ready in (True, None):
            got = clip(ready, False)
            for t in range(T):
                if not (torch.equal(got[t][0], ref[t][0]) and torch.equal(got[t][1], ref[t][1])):
                    d = float((got[t][0] - ref[t][0]).abs().max())
                    print("MISMATCH rep %d ready %s frame %d: alpha max-abs %.3e" % (r, ready, t, d), flush=True)
                    bad += 1
    print("race_stress %dx%d batch %d (graphs %s): %d repetitions x 2 modes x %d frames, %d mismatching frames" % (W, H, NB, os.environ.get("OTVM_GRAPHS", "auto"), args.reps, T, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
