"""Golden fixture of the TRAINING-mode forward, produced by running the REFERENCE itself.

    python -m tests.golden.make_train_golden        (development container only: imports /root/reference)

Builds the reference's training classes -- models.alpha.model.FullModel over models.trimap.model.FullModel, stage 4
(helpers.get_model_alpha(cfg, model_trimap, mode='Train')) -- through tools/ref_import.py, loads the synthetic checkpoint,
freezes the BatchNorms as train.py:311-319 does (m.eval() on every nn.BatchNorm2d) and runs FullModel.forward
(models/alpha/model.py:189-312) on seeded B x sample_length clips.  Stored: the inputs as seeds, the four loss scalars and
the visualisation tensors the function returns.  Data only.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tools.ref_import import load_reference  # noqa: E402
from otvm_amd.synth_weights import synthetic_state_dict  # noqa: E402
from otvm_amd.synth_data import train_batch  # noqa: E402

CASES = [
    # name, B, S, H, W, seed
    ("b2_s3_64x64", 2, 3, 64, 64, 11),
    ("b1_s4_64x96", 1, 4, 64, 96, 12),
]


def build_train_model():
    helpers = load_reference()
    cfg = types.SimpleNamespace(TRAIN=types.SimpleNamespace(STAGE=4))
    with contextlib.redirect_stdout(io.StringIO()):
        mt = helpers.get_model_trimap(cfg, "Train", None)
        m = helpers.get_model_alpha(cfg, mt, "Train", None)
    m.train()
    for mod in m.modules():                               # train.py:311-319: BatchNorm stays in eval mode
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eval()
    return m


def run_case(B, S, H, W, seed):
    m = build_train_model()
    missing = m.load_state_dict(synthetic_state_dict(0), strict=True)
    a, fg, bg, tri = train_batch(B, S, H, W, seed)
    with torch.no_grad():
        out = m(torch.from_numpy(a), torch.from_numpy(fg), torch.from_numpy(bg), tri=torch.from_numpy(tri))
    names = ("loss1", "loss2", "loss3", "loss_trimap", "scaled_imgs", "tris_vis", "alphas", "comps", "scaled_gts", "Fs", "Bs",
             "preds_trimap")
    return {k: np.asarray(v.detach().numpy(), dtype=np.float32) for k, v in zip(names, out)}


def main():
    for name, B, S, H, W, seed in CASES:
        r = run_case(B, S, H, W, seed)
        np.savez_compressed(os.path.join(HERE, "train_%s.npz" % name), B=B, S=S, H=H, W=W, seed=seed, **r)
        print(name, {k: (float(v) if v.size == 1 else v.shape) for k, v in r.items()})


if __name__ == "__main__":
    main()
