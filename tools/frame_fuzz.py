"""Randomised end-to-end check: short clips at random resolutions / schedules through EvalModel.forward against the CPU
oracle (GPU only; the oracle is test infrastructure, this is a test tool).

    python tools/frame_fuzz.py [--n 12] [--seed 0] [--max-side 220]

Per case: random H, W (odd sizes, not multiples of 32), demo flow (first-frame trimap) or V108 flow (trimap from the
ground-truth alpha, separate backgrounds), random memory period / bank size / dilation, uint8 or fp32 frames.  Alpha must
stay within 1e-3 of the oracle on every frame; a frame whose class map differs from the oracle's only at near-ties is
re-run with the tie-breaks imposed (tests/test_gpu_frame.py explains the protocol).
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=12)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-side", type=int, default=220)
    ap.add_argument("--precision", default=None, choices=["f32", "f16x3"])
    ap.add_argument("--trace", action="store_true", help="print the distance of every frame")
    ap.add_argument("--keep-going", action="store_true", help="count frames over the tolerance instead of stopping at the first")
    args = ap.parse_args()
    from oracle.otvm_oracle import OtvmOracle
    from otvm_amd import helpers
    from otvm_amd.synth_data import soft_alpha, synthetic_clip
    from otvm_amd.synth_weights import synthetic_state_dict
    from otvm_amd.video import memory_schedule
    rng = random.Random(args.seed)
    sd = synthetic_state_dict(0)
    cfg = helpers.default_cfg()
    models = {}
    worst, ties_total, failed = 0.0, 0, 0
    for it in range(args.n):
        H, W = rng.randint(33, args.max_side), rng.randint(33, args.max_side)
        T = rng.randint(3, 5)
        dk = rng.choice([5, 12, 20])
        skip, max_num = rng.choice([3, 4, 10]), rng.choice([1, 2, 5])
        v108 = rng.random() < 0.4
        u8 = rng.random() < 0.5
        abandoned = rng.random() < 0.3
        if dk not in models:
            m = helpers.get_model_alpha(cfg, helpers.get_model_trimap(cfg, "Test", dk), "Test", dk)
            m.load_state_dict(sd, strict=True)
            m.precision = args.precision
            models[dk] = m.cuda().eval()
        m = models[dk]
        orc = OtvmOracle(sd, dilate_kernel=dk)
        frames, tri = synthetic_clip(H, W, T, seed=7000 + it)
        bgs, _ = synthetic_clip(H, W, T, seed=8000 + it)
        desc = "%dx%d T%d dk%d skip%d max%d %s %s%s" % (H, W, T, dk, skip, max_num, "v108" if v108 else "demo",
                                                          "u8" if u8 else "f32", " abandoned" if abandoned else "")
        for t in range(T):
            f32 = lambda x: torch.from_numpy(x.astype(np.float32)).permute(2, 0, 1)[None, None].contiguous()
            fg_ref, bg_ref = f32(frames[t]), (f32(bgs[t]) if v108 else f32(frames[t]))
            if u8:
                fg_in, bg_in = torch.from_numpy(frames[t].copy()), torch.from_numpy((bgs[t] if v108 else frames[t]).copy())
            else:
                fg_in, bg_in = fg_ref, bg_ref
            if v108:
                a = torch.from_numpy(soft_alpha(H, W, t))[None, None, None]
                tg = None
            else:
                a = torch.ones(1, 1, 1, H, W)
                tg = torch.from_numpy(np.asarray(tri))[None, None]
            memorize, mx, _ = memory_schedule(t, H, W, skip, max_num)
            # some clips are abandoned without ever passing last_frame=True (the next clip must still start clean)
            kw = dict(first_frame=(t == 0), last_frame=(t == T - 1) and not abandoned, memorize=memorize, max_memory_num=mx)
            out = m(a, fg_in, bg_in, tri_gt=tg, **kw)
            torch.cuda.synchronize()
            pl = m._engine.last_plan
            cls_h = pl.CLS.reshape(pl.Hp, pl.Wp).cpu().long()
            bank_before = list(orc.bank)
            cap = {}
            ref = orc.frame(a, fg_ref, bg_ref, tri_gt=tg, frame_id=t, capture=cap, **kw)
            if not torch.equal(cls_h, cap["cls"]):
                diff = cls_h != cap["cls"]
                top2 = torch.sort(cap["tri_in"][0], dim=0, descending=True)[0]
                gap = float((top2[0] - top2[1])[diff].max())
                if gap >= 2e-3:
                    print("FAIL", desc, "frame", t, "class map differs away from a tie (gap %.3e)" % gap)
                    sys.exit(1)
                ties_total += int(diff.sum())
                orc.bank = bank_before
                ref = orc.frame(a, fg_ref, bg_ref, tri_gt=tg, frame_id=t, capture={}, class_override=cls_h, **kw)
            d = float((out[3].cpu() - ref[3]).abs().max())
            worst = max(worst, d)
            if args.trace:
                print("     %s frame %d alpha max-abs %.3e" % (desc, t, d), flush=True)
            if not (d <= 1e-3 and bool(torch.isfinite(out[3]).all())):
                print("FAIL", desc, "frame", t, "alpha max-abs %.3e" % d)
                if not args.keep_going:
                    sys.exit(1)
                failed += 1
            if m.memories["frames"] != [b[2] for b in orc.bank]:
                print("FAIL", desc, "frame", t, "bank", m.memories["frames"], [b[2] for b in orc.bank])
                sys.exit(1)
        print("ok  ", desc)
    print("frame_fuzz: %d clips, worst alpha max-abs %.3e, tie-breaks %d%s" % (args.n, worst, ties_total, ", %d frames over 1e-3" % failed if failed else ""))
    if failed:
        sys.exit(1)


if __name__ == "__main__":
    main()
