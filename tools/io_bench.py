"""End-to-end throughput of run_video_matte_io at 1080p: JPEG decode + upload + matte + PNG encode.

    python tools/io_bench.py [--frames 60] [--decode-workers 8] [--encode-workers 8]
"""
import argparse
import io
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--decode-workers", type=int, default=8)
    ap.add_argument("--encode-workers", type=int, default=8)
    args = ap.parse_args()
    from PIL import Image
    from bench import build_model, device_clip
    from otvm_amd.io_pipeline import run_video_matte_io
    from otvm_amd.synth_data import disc_trimap
    from otvm_amd.video import run_video_matte
    dev = torch.device("cuda:0")
    model, _ = build_model(dev)
    H, W, T = args.height, args.width, args.frames
    clip = device_clip(H, W, T, 77, dev)
    enc = []
    for t in range(T):
        rgb = clip[t][0, 0].flip(0).permute(1, 2, 0).byte().cpu().numpy()
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, format="JPEG", quality=92)
        enc.append(buf.getvalue())
    tri = disc_trimap(H, W)
    out = tempfile.mkdtemp(prefix="otvm_io_")
    run_video_matte_io(model, enc[:4], tri, skip=5, max_num=5, outdir=out, decode_workers=args.decode_workers,
                       encode_workers=args.encode_workers)                     # warm-up (plan build)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_video_matte_io(model, enc, tri, skip=5, max_num=5, outdir=out, decode_workers=args.decode_workers,
                       encode_workers=args.encode_workers)
    torch.cuda.synchronize()
    t_io = time.perf_counter() - t0
    frames = [clip[t][0, 0].flip(0).permute(1, 2, 0).contiguous() for t in range(T)]
    t0 = time.perf_counter()
    run_video_matte(model, frames, trimap=tri, skip=5, max_num=5, frames_are_rgb=True, keep_on_device=True)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    kb = sum(len(b) for b in enc) / T / 1024
    print("%dx%d, %d frames, JPEG %.0f KiB/frame: with IO (decode x%d, PNG encode x%d) %.1f frames/s; frames resident in HBM %.1f frames/s"
          % (W, H, T, kb, args.decode_workers, args.encode_workers, T / t_io, T / t_gpu))


if __name__ == "__main__":
    main()
