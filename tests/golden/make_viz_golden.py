"""Generate the `--viz` composite fixture by running the REFERENCE's own write_image (development container only).

    python -m tests.golden.make_viz_golden

eval.py cannot be imported as it stands (cv2 IO, yacs, the training-time dependencies of dataset.py), so its module is
loaded with the shims of tools/ref_import.py plus inert stand-ins for `config` and `dataset` (write_image touches
neither), and torchvision.utils.save_image is replaced by a recorder: the fixture holds the tensor and `nrow` the
reference hands to save_image (eval.py:96-115) for seeded inputs, plus the green-screen composite of eval.py:199-203.
What torchvision then does with them (make_grid padding 2, x*255+0.5 clamp, uint8) is third-party behaviour restated in
otvm_amd/viz.py and pinned by tests/test_host_logic.py::test_viz_grid_layout_and_rounding.  Data only travels.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tools.ref_import import REF, load_reference  # noqa: E402


def load_reference_eval():
    load_reference()                                        # cv2 / torchvision / stty shims, sys.path
    rec = {}

    def save_image(tensor, fp, nrow=8, **kw):
        rec["imgs"], rec["nrow"], rec["path"] = tensor.detach().clone(), int(nrow), fp
    sys.modules["torchvision.utils"].save_image = save_image
    cfgm = types.ModuleType("config")
    cfgm.get_cfg_defaults = lambda: None
    sys.modules["config"] = cfgm
    dsm = types.ModuleType("dataset")
    dsm.EvalDataset = dsm.VideoMatting108_Test = dsm.Demo_Test = object
    sys.modules["dataset"] = dsm
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_eval", os.path.join(REF, "eval.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, rec


def main():
    ev, rec = load_reference_eval()
    out = {}
    for i, (h, w) in enumerate([(22, 30), (33, 47)]):       # even and odd sizes (the panels are h//2 x w//2)
        g = torch.Generator().manual_seed(100 + i)
        scaled_imgs = torch.rand(1, 1, 3, h, w, generator=g)
        tri_pred = torch.softmax(torch.randn(1, 1, 3, h, w, generator=g) * 2, 2)
        tri_gt = torch.nn.functional.one_hot(torch.randint(0, 3, (1, 1, h, w), generator=g), 3).permute(0, 1, 4, 2, 3).float()
        alphas = torch.rand(1, 1, 1, h, w, generator=g)
        scaled_gts = torch.rand(1, 1, 1, h, w, generator=g)
        green_bg = torch.zeros_like(scaled_imgs)
        green_bg[:, :, 1] = 1.
        comps = scaled_imgs * alphas + green_bg * (1. - alphas)          # as eval.py:199-203 builds its 6th element
        ev.write_image("%s", (scaled_imgs, tri_pred, tri_gt, alphas, scaled_gts, comps), "f0.jpg")
        out.update({"in%d_imgs" % i: scaled_imgs.numpy(), "in%d_tri_pred" % i: tri_pred.numpy(),
                    "in%d_tri_gt" % i: tri_gt.numpy(), "in%d_alpha" % i: alphas.numpy(), "in%d_gt" % i: scaled_gts.numpy(),
                    "out%d_imgs" % i: rec["imgs"].numpy(), "out%d_nrow" % i: np.int64(rec["nrow"])})
    np.savez_compressed(os.path.join(HERE, "viz.npz"), **out)
    print("viz.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
