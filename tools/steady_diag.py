"""Diagnose the 1080p steady-state parity frame (tests/test_gpu_fullsize.py::test_1080p_steady_state_frame_vs_oracle):
HIP f16x3 and HIP exact-fp32 against the oracle, the oracle's own summation-order noise (oneDNN on/off), per-stage diffs.

    python tools/steady_diag.py [--t 21] [--seed 23]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--t", type=int, default=21)
    ap.add_argument("--seed", type=int, default=23)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--ready", type=int, default=0, help="pass _inputs_ready=True to the free-running frames")
    args = ap.parse_args()
    from oracle.otvm_oracle import OtvmOracle
    from otvm_amd import helpers
    from otvm_amd.synth_data import synthetic_clip
    from otvm_amd.synth_weights import synthetic_state_dict
    from tests.test_gpu_frame import fmt, stage_report
    sd = synthetic_state_dict(0)
    H, W, T, t_s = args.height, args.width, args.t + 3, args.t
    frames, tri = synthetic_clip(H, W, t_s + 1, seed=args.seed)
    flags = lambda t: dict(first_frame=(t == 0), last_frame=(t == T - 1), memorize=(t % 5 == 0), max_memory_num=5)

    def tensors(t):
        fg = torch.from_numpy(frames[t].astype(np.float32)).permute(2, 0, 1)[None, None].contiguous()
        return torch.ones(1, 1, 1, H, W), fg, torch.from_numpy(tri)[None, None]
    res = {}
    for prec in ("f16x3", "f32"):
        cfg = helpers.default_cfg()
        m = helpers.get_model_alpha(cfg, helpers.get_model_trimap(cfg, "Test", 12), "Test", 12)
        m.load_state_dict(sd, strict=True)
        m.precision = prec
        m = m.cuda().eval()
        for t in range(t_s):
            a, fg, tg = tensors(t)
            m(a.cuda(), fg.cuda(), fg.cuda(), tri_gt=tg.cuda(), _frame_id=t, _inputs_ready=(True if args.ready else None), **flags(t))
        eng = m._engine
        eng.flush()
        torch.cuda.synchronize()
        pl = eng.last_plan
        hw, h16, w16 = pl.hw, pl.Hp // 16, pl.Wp // 16
        bank = [(s["k"].t.reshape(hw, 128).t().reshape(128, h16, w16).cpu().contiguous(),
                 s["v"].t.reshape(hw, 512).t().reshape(512, h16, w16).cpu().contiguous(), s["frame"]) for s in eng.bank]
        a, fg, tg = tensors(t_s)
        out = m(a.cuda(), fg.cuda(), fg.cuda(), tri_gt=tg.cuda(), _frame_id=t_s, **flags(t_s))
        torch.cuda.synchronize()
        orc = OtvmOracle(sd, dilate_kernel=12)
        orc.bank = list(bank)
        cap = {}
        ref = orc.frame(a, fg, fg.clone(), tri_gt=tg, frame_id=t_s, capture=cap, **flags(t_s))
        cls_h = pl.CLS.reshape(pl.Hp, pl.Wp).cpu().long()
        d = (out[3].cpu() - ref[3]).abs()
        print("%s: alpha max-abs %.3e (mean %.3e, #>5e-4: %d), class flips %d | %s" %
              (prec, float(d.max()), float(d.mean()), int((d > 5e-4).sum()), int((cls_h != cap["cls"]).sum()),
               fmt(stage_report(pl, cap, False))), flush=True)
        if prec == "f16x3":
            # the oracle's own reorder noise on this very frame: same bank, oneDNN off
            orc2 = OtvmOracle(sd, dilate_kernel=12)
            orc2.bank = list(bank)
            with torch.backends.mkldnn.flags(enabled=False):
                ref2 = orc2.frame(a, fg, fg.clone(), tri_gt=tg, frame_id=t_s, **flags(t_s))
            d2 = (ref2[3] - ref[3]).abs()
            print("oracle self-noise (oneDNN on vs off): alpha max-abs %.3e (mean %.3e), trimap %.3e" %
                  (float(d2.max()), float(d2.mean()), float((ref2[1] - ref[1]).abs().max())), flush=True)
        res[prec] = out[3].cpu()
        del m
        torch.cuda.empty_cache()
    print("f16x3 vs f32 HIP: %.3e" % float((res["f16x3"] - res["f32"]).abs().max()))


if __name__ == "__main__":
    main()
