"""Shared helpers for the parity tests (CPU oracle side)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_sequences_meta():
    return json.load(open(os.path.join(GOLDEN, "sequences.json")))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, "seq_%s.npz" % name))


def clip_inputs(meta):
    """Rebuild the seeded inputs of a golden sequence: list of per-frame (a, fg, bg, tri_gt) tensors."""
    from otvm_amd.synth_data import synthetic_clip, soft_alpha
    H, W, T = meta["H"], meta["W"], meta["T"]
    frames, tri = synthetic_clip(H, W, T, meta["clip_seed"])
    out = []
    for t in range(T):
        fg = torch.from_numpy(frames[t].astype(np.float32)).permute(2, 0, 1)[None, None].contiguous()
        if meta["style"] == "demo":
            a = torch.ones(1, 1, 1, H, W)
            tri_gt = torch.from_numpy(tri)[None, None]
        else:
            a = torch.from_numpy(soft_alpha(H, W, t))[None, None, None]
            tri_gt = None
        out.append((a, fg, fg.clone(), tri_gt))
    return out


def frame_flags(meta, t):
    """eval.py:178-189 for frame t."""
    skip = meta["skip"]
    return dict(first_frame=(t == 0), last_frame=(t == meta["T"] - 1),
                memorize=((t % skip) == 0) if skip > 2 else False, max_memory_num=meta["max_num"])
