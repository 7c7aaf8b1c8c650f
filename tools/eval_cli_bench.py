"""Frames per second of the eval.py-shaped command line on a demo-layout 1080p clip (JPEG frames + one trimap PNG on disk,
alpha PNGs written), pipelined IO vs --sync-io (the reference's blocking pattern, eval.py:209-217), next to io_bench's
figure for the same pipeline driven directly (eval_cli within 5 % of io_bench).

    python tools/eval_cli_bench.py [--frames 100] [--skip 5]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--skip", type=int, default=5)
    args = ap.parse_args()
    from PIL import Image
    from otvm_amd import eval_cli
    from otvm_amd.synth_data import synthetic_clip
    H, W, T = 1080, 1920, args.frames
    frames_bgr, tri = synthetic_clip(H, W, T, seed=31)
    root = tempfile.mkdtemp(prefix="otvm_demo_")
    os.makedirs(os.path.join(root, "clip", "frames")); os.makedirs(os.path.join(root, "clip", "trimap"))
    for t in range(T):
        Image.fromarray(frames_bgr[t][..., ::-1].copy()).save(os.path.join(root, "clip", "frames", "%05d.jpg" % t), quality=92)
    g = (np.asarray(tri)[1] * 128 + np.asarray(tri)[2] * 254).astype(np.uint8)
    Image.fromarray(g).save(os.path.join(root, "clip", "trimap", "00000.png"))
    base = ["--demo", "--data", root, "--synthetic-weights", "--skip", str(args.skip)]
    eval_cli.main(base + ["--out", os.path.join(root, "warm"), "--max-frames", "6"])          # plan build, tuner, module load
    for label, extra in (("pipelined IO", []), ("--sync-io", ["--sync-io"])):
        s = eval_cli.main(base + ["--out", os.path.join(root, "out_" + label.strip("-").replace(" ", "_"))] + extra)
        print("eval_cli %-13s: %d frames, %.1f frames/s" % (label, s["frames"], s["fps"]), flush=True)


if __name__ == "__main__":
    main()
