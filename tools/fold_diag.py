"""Compare the plan buffers of a first frame with the GroupNorm-apply folds on and off (debugging aid, GPU only)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from otvm_amd import engine, helpers
    from otvm_amd.synth_data import synthetic_clip
    from otvm_amd.synth_weights import synthetic_state_dict
    H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 480, int(sys.argv[2]) if len(sys.argv) > 2 else 832
    sd = synthetic_state_dict(0)
    frames, tri = synthetic_clip(H, W, 1, seed=22)
    fg = torch.from_numpy(frames[0].astype(np.float32)).permute(2, 0, 1)[None, None].contiguous().cuda()
    a = torch.ones(1, 1, 1, H, W).cuda()
    tg = torch.from_numpy(tri)[None, None].cuda()
    snaps = []
    for fuse in (False, True):
        engine.FUSE_GN_APPLY = fuse
        cfg = helpers.default_cfg()
        m = helpers.get_model_alpha(cfg, helpers.get_model_trimap(cfg, "Test", 12), "Test", 12)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        out = m(a, fg, fg.clone(), tri_gt=tg, first_frame=True, last_frame=True, memorize=True, max_memory_num=5)
        torch.cuda.synchronize()
        pl = m._engine.last_plan
        snap = {"alpha": out[3].clone()}
        for name, act in (("ppmcat_l4", pl.PPMCAT.ch(0, 2048)), ("ppm0", pl.PPMCAT.ch(2048, 256)), ("ppm3", pl.PPMCAT.ch(2816, 256)),
                          ("U2up", pl.U2.ch(0, 256)), ("U3up", pl.U3.ch(0, 256)), ("D80up", pl.D80.ch(0, 64)),
                          ("D80rest", pl.D80.ch(64, 16)), ("h32", pl.buf("h32", pl.Hp, pl.Wp, 32)),
                          ("r_layer1", pl.buf("r_layer1", pl.Hp, pl.Wp, 64)), ("r_layer2", pl.buf("r_layer2", pl.Hp, pl.Wp, 64))):
            snap[name] = act.torch().clone()
        snaps.append(snap)
        print("fuse", fuse, "tune log entries", len(engine.TUNE_LOG))
    for k in snaps[0]:
        d = (snaps[0][k] - snaps[1][k]).abs()
        print("%-10s max-abs diff %.3e (max |ref| %.3e) nan %d" % (k, float(d.max()), float(snaps[0][k].abs().max()),
                                                                  int(torch.isnan(snaps[1][k]).sum())))


if __name__ == "__main__":
    main()
