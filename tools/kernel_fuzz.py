"""Randomised checks of the non-convolution kernels against torch / the oracle on the CPU (GPU only).

    python tools/kernel_fuzz.py [--n 60] [--seed 0]

GroupNorm (stats + apply, in place and out of place, residual, odd pixel counts), bilinear upsampling (arbitrary
sizes, fused add), 3x3/2 max pooling, PPM pooling (maps smaller than the 6x6 grid included), the memory read
(T, map sizes that are not multiples of the 64-row tile, both variants) and the trimap distance encoding.
"""
import argparse
import ctypes as C
import os
import random
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as G          # noqa: E402
from otvm_amd import lib as L            # noqa: E402


def check(name, got, ref, tol, desc):
    err = float((got - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    if not (bool(torch.isfinite(got).all()) and err <= tol):
        print("FAIL %s %s err %.3e" % (name, desc, err))
        sys.exit(1)
    return err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    lib = L.load()
    st = G.stream()
    from oracle.otvm_oracle import make_trimap8, memory_read         # test infrastructure (this is a test tool)
    worst = {}
    for it in range(args.n):
        g = torch.Generator().manual_seed(5000 + it)
        # ---- GroupNorm
        Cc = rng.choice([64, 128, 192, 256, 320, 512, 1024, 2048])
        H, W = rng.randint(1, 40), rng.randint(1, 40)
        act, use_res, inplace = rng.choice([0, 1, 2]), rng.random() < 0.4, rng.random() < 0.5
        x = torch.randn(1, Cc, H, W, generator=g) * rng.uniform(0.2, 3) + rng.uniform(-1, 1)
        ga, be = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
        res = torch.randn(1, Cc, H, W, generator=g) if use_res else None
        ref = F.group_norm(x, 32, ga, be, 1e-5)
        if use_res:
            ref = ref + res
        ref = F.relu(ref) if act == 1 else (F.leaky_relu(ref, 0.01) if act == 2 else ref)
        xa = G.to_act(x, ld=Cc + rng.choice([0, 4]), off=rng.choice([0, 4]))
        out = xa if inplace else G.empty_act(H, W, Cc)
        ra = G.to_act(res) if use_res else None
        gd, bd = ga.to(G.DEV), be.to(G.DEV)
        stats = torch.zeros(64, dtype=torch.float64, device=G.DEV)
        L.check(lib.otvm_gn_stats(xa.ptr, H * W, Cc, xa.ld, stats.data_ptr(), st))
        L.check(lib.otvm_gn_apply(xa.ptr, H * W, Cc, xa.ld, stats.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                                  0 if ra is None else ra.ptr, 0 if ra is None else ra.ld, 0, 0, 0, act, out.ptr, out.ld, st))
        torch.cuda.synchronize()
        worst["gn"] = max(worst.get("gn", 0), check("groupnorm", G.from_act(out), ref, 3e-5, "C%d %dx%d act%d res%d" % (Cc, H, W, act, use_res)))
        # ---- upsample (+ add)
        Cu = rng.choice([4, 16, 64, 256])
        hi, wi = rng.randint(1, 24), rng.randint(1, 24)
        ho, wo = rng.randint(hi, 4 * hi + 3), rng.randint(wi, 4 * wi + 3)
        xu = torch.randn(1, Cu, hi, wi, generator=g)
        add = torch.randn(1, Cu, ho, wo, generator=g) if rng.random() < 0.4 else None
        refu = F.interpolate(xu, size=(ho, wo), mode="bilinear", align_corners=False)
        if add is not None:
            refu = refu + add
        xua, oua = G.to_act(xu), G.empty_act(ho, wo, Cu)
        aa = G.to_act(add) if add is not None else None
        L.check(lib.otvm_upsample_bilinear(xua.ptr, hi, wi, Cu, xua.ld, 0, 0, 0, 0 if aa is None else aa.ptr, 0 if aa is None else aa.ld,
                                           oua.ptr, ho, wo, oua.ld, st))
        torch.cuda.synchronize()
        worst["up"] = max(worst.get("up", 0), check("upsample", G.from_act(oua), refu, 1e-5, "C%d %dx%d->%dx%d" % (Cu, hi, wi, ho, wo)))
        # ---- maxpool 3x3 / 2, pad 1
        hm, wm = rng.randint(2, 40), rng.randint(2, 40)
        xm = torch.randn(1, 64, hm, wm, generator=g)
        refm = F.max_pool2d(xm, 3, 2, 1)
        xma, oma = G.to_act(xm), G.empty_act(refm.shape[2], refm.shape[3], 64)
        L.check(lib.otvm_maxpool3x3s2(xma.ptr, hm, wm, 64, xma.ld, oma.ptr, oma.ld, st))
        torch.cuda.synchronize()
        check("maxpool", G.from_act(oma), refm, 0.0, "%dx%d" % (hm, wm))
        # ---- PPM pooling
        hp, wp, Cp = rng.randint(1, 40), rng.randint(1, 40), rng.choice([4, 64, 260, 512, 2048])
        xp = torch.randn(1, Cp, hp, wp, generator=g)
        xpa = G.to_act(xp)
        pool = torch.empty(50 * Cp, device=G.DEV)
        pws = torch.empty(int(lib.otvm_ppm_pool_ws_bytes(hp, Cp)), dtype=torch.uint8, device=G.DEV)
        L.check(lib.otvm_ppm_pool(xpa.ptr, hp, wp, Cp, xpa.ld, pool.data_ptr(), pws.data_ptr(), st))
        torch.cuda.synchronize()
        base = 0
        for s in (1, 2, 3, 6):
            refp = F.adaptive_avg_pool2d(xp, s)[0].permute(1, 2, 0).reshape(s * s, Cp)
            gotp = pool[base * Cp:(base + s * s) * Cp].reshape(s * s, Cp).cpu()
            worst["ppm"] = max(worst.get("ppm", 0), check("ppm_pool", gotp, refp, 1e-5, "C%d %dx%d s%d" % (Cp, hp, wp, s)))
            base += s * s
        # ---- memory read (every 3rd iteration: the CPU reference is the slow part)
        if it % 3 == 0:
            T, h, w = rng.randint(1, 11), rng.randint(2, 20), rng.randint(2, 24)
            hw = h * w
            mk, mv = torch.randn(128, T, h, w, generator=g) * 2.5, torch.randn(512, T, h, w, generator=g)
            qk, qv = torch.randn(128, h, w, generator=g) * 2.5, torch.randn(512, h, w, generator=g)
            refr = memory_read(mk, mv, qk, qv)[:512].reshape(512, hw).t()
            keys = [mk[:, t].reshape(128, hw).t().contiguous().to(G.DEV) for t in range(T)]
            vals = [mv[:, t].reshape(512, hw).t().contiguous().to(G.DEV) for t in range(T)]
            q = qk.reshape(128, hw).t().contiguous().to(G.DEV)
            outr = torch.full((hw, 512), float("nan"), device=G.DEV)
            ws = torch.empty(int(lib.otvm_memory_read_ws_bytes(hw, T)), dtype=torch.uint8, device=G.DEV)
            slots = []
            for t in range(T):
                sl = torch.zeros(int(lib.otvm_bank_slot_bytes_f16x3(hw)), dtype=torch.uint8, device=G.DEV)
                L.check(lib.otvm_bank_pack_f16x3(keys[t].data_ptr(), vals[t].data_ptr(), hw, sl.data_ptr(), st))
                slots.append(sl)
            sp = (C.c_void_p * T)(*[s_.data_ptr() for s_ in slots])
            L.check(lib.otvm_memory_read_f16x3(q.data_ptr(), 128, sp, T, hw, outr.data_ptr(), 512, ws.data_ptr(), st))
            torch.cuda.synchronize()
            worst["mem"] = max(worst.get("mem", 0), check("memory_read_f16x3", outr.cpu(), refr, 2e-5, "T%d %dx%d" % (T, h, w)))
        # ---- trimap distance encoding
        if it % 2 == 0:
            Hp, Wp = 16 * rng.randint(1, 6), 16 * rng.randint(1, 8)
            probs = torch.softmax(torch.randn(3, Hp, Wp, generator=g) * rng.choice([0.5, 3.0]), 0)
            kind = rng.choice(["random", "blob", "no_fg", "no_bg"])
            if kind == "blob":
                yy, xx = torch.meshgrid(torch.arange(Hp), torch.arange(Wp), indexing="ij")
                r = ((yy - Hp / 2) ** 2 + (xx - Wp / 2) ** 2).float().sqrt()
                probs = torch.stack([(r > Hp / 3).float(), ((r <= Hp / 3) & (r > Hp / 5)).float(), (r <= Hp / 5).float()])
            elif kind == "no_fg":
                probs[2] = 0
            elif kind == "no_bg":
                probs[0] = 0
            P = Hp * Wp
            pd = probs.contiguous().to(G.DEV)
            x11 = torch.zeros(P * 12, device=G.DEV)
            d80 = torch.zeros(P * 80, device=G.DEV)
            cls = torch.empty(P, dtype=torch.uint8, device=G.DEV)
            ews = torch.empty(int(lib.otvm_trimap_encode_ws_bytes(Hp, Wp)), dtype=torch.uint8, device=G.DEV)
            L.check(lib.otvm_trimap_encode(pd.data_ptr(), Hp, Wp, 0, cls.data_ptr(), x11.data_ptr(), 12, d80.data_ptr(), 80,
                                           ews.data_ptr(), st))
            torch.cuda.synchronize()
            got8 = x11.reshape(Hp, Wp, 12)[..., 3:11].permute(2, 0, 1).cpu()
            ref8 = make_trimap8(probs)
            worst["edt"] = max(worst.get("edt", 0), check("trimap_encode", got8, ref8, 2e-6, "%dx%d %s" % (Hp, Wp, kind)))
        # ---- matting metrics (SAD / MSE / dtSSD partial sums per frame) on random sizes and masks
        if it % 2 == 1:
            from oracle import metrics_oracle as M
            from otvm_amd.video import ClipMetrics
            B, hm2, wm2 = rng.randint(2, 5), rng.randint(1, 70), rng.randint(1, 90)
            pr = torch.randint(0, 256, (B, hm2, wm2), generator=g).float()
            tg2 = torch.randint(0, 256, (B, hm2, wm2), generator=g).float()
            mk = (torch.rand(B, hm2, wm2, generator=g) < rng.uniform(0.05, 0.9)).float()
            cm = ClipMetrics(G.DEV, capacity=2)
            for i in range(B):
                cm.add(pr[i].to(torch.uint8).to(G.DEV), tg2[i].to(torch.uint8).to(G.DEV), mk[i].to(torch.uint8).to(G.DEV))
            torch.cuda.synchronize()
            r = cm.result()
            e, n = M.dtssd(pr, tg2, mk)
            for name, got_v, ref_v in (("sad", r["sad_per_frame"], M.sad(pr, tg2, mk)), ("mse", r["mse_per_frame"], M.mse(pr, tg2, mk)),
                                       ("dtssd", r["dtssd_per_pair"], e), ("dtssd_n", r["dtssd_num_per_pair"], n)):
                worst["met"] = max(worst.get("met", 0), check("metrics." + name, torch.tensor(got_v, dtype=torch.float64),
                                                                ref_v.double(), 1e-5, "%dx%dx%d" % (B, hm2, wm2)))
    print("kernel_fuzz: %d rounds, worst relative errors %s" % (args.n, {k: "%.2e" % v for k, v in worst.items()}))


if __name__ == "__main__":
    main()
