"""Randomised check of otvm_conv2d (both precisions, every dispatch route) against torch on the CPU (GPU only).

    python tools/conv_fuzz.py [--n 200] [--seed 0]

Shapes are drawn so that all tile families, the patch kernels, the split-K route, ragged M / N tiles, channel-slice
views (ld > C, offset) and the epilogue options (bias, residual, activation, fused GroupNorm sums) are hit.
"""
import argparse
import math
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as G          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--patch64", action="store_true", help="only 3x3 stride-1 layers with 33 ... 64 filters: the nine-tap patch tiles "
                                                          "(either matrix-core form, OTVM_PATCH_M16)")
    ap.add_argument("--verbose", action="store_true", help="print every case BEFORE it is launched (to find a case that faults)")
    args = ap.parse_args()
    rng = random.Random(args.seed)
    ws = torch.empty(16 << 20, device=G.DEV)
    worst = 0.0
    for it in range(args.n):
        k = rng.choice([1, 1, 3, 3, 3, 7])
        stride = rng.choice([1, 1, 1, 2])
        dil = rng.choice([1, 1, 2, 4]) if k == 3 else 1
        Cin = rng.choice([3, 11, 16, 24, 32, 64, 72, 96, 128, 256, 512, 1024, 2048])
        if k == 7:
            Cin = rng.choice([3, 11, 22, 24])
        Cout = rng.choice([3, 16, 32, 48, 64, 128, 192, 256, 384, 512, 1024])
        if args.patch64:
            k, stride, dil = 3, 1, rng.choice([1, 1, 1, 2, 4])
            Cin, Cout = rng.choice([16, 32, 48, 64, 80, 128, 320]), rng.choice([40, 48, 64, 64, 64])
        # keep the CPU reference cheap: bound M * K * Cout
        budget = 6e9
        maxM = int(budget / (Cin * k * k * Cout * 2))
        side = max(6, min(rng.choice([6, 9, 17, 33, 64, 100, 180, 300]), int(math.sqrt(max(36, maxM)))))
        H, W = side, max(6, int(side * rng.uniform(0.7, 1.6)))
        pad = dil * (k - 1) // 2 if rng.random() < 0.85 else 0
        if (H + 2 * pad - dil * (k - 1) - 1) // stride + 1 < 1 or (W + 2 * pad - dil * (k - 1) - 1) // stride + 1 < 1:
            continue
        use_bias, act, in_relu = rng.random() < 0.6, rng.choice([0, 0, 1, 2]), int(rng.random() < 0.2)
        gn = rng.random() < 0.25 and Cout % 64 == 0 and (Cout // 32) & (Cout // 32 - 1) == 0
        use_res = rng.random() < 0.3 and not gn
        if gn:
            act = 0
        prec = 1 if args.patch64 else rng.choice([0, 1, 1, 1])
        g = torch.Generator().manual_seed(1000 + it)
        x = torch.randn(1, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
        b = torch.randn(Cout, generator=g) if use_bias else None
        ref = F.conv2d(F.relu(x) if in_relu else x, w, b, stride, pad, dil)
        res = torch.randn(ref.shape, generator=g) if use_res else None
        if use_res:
            ref = ref + res
        ref = F.relu(ref) if act == 1 else (F.leaky_relu(ref, 0.01) if act == 2 else ref)
        cw = G.pack_weight(w)
        off = rng.choice([0, 4, 8])
        xa = G.to_act(x, ld=cw.I_pad + off, off=off)
        out = G.empty_act(ref.shape[2], ref.shape[3], max(4, (Cout + 3) // 4 * 4), ld=(Cout + 3) // 4 * 4 + rng.choice([0, 4]),
                          off=rng.choice([0, 4]))
        ra = G.to_act(res) if use_res else None
        bd = None if b is None else b.to(G.DEV)
        stats = torch.zeros(64, dtype=torch.float64, device=G.DEV) if gn else None
        use_ws = rng.random() < 0.8
        tag = "Cin %4d Cout %4d k%d s%d d%d p%d %3dx%-3d prec %d bias %d act %d relu_in %d res %d gn %d" % (
            Cin, Cout, k, stride, dil, pad, H, W, prec, use_bias, act, in_relu, use_res, gn)
        if args.verbose:
            print("case %d: %s in_ld %d out_ld %d ws %d" % (it, tag, xa.ld, out.ld, use_ws), flush=True)
        G.conv2d(xa, cw, out, bd, stride, pad, dil, act, in_relu, ra, precision=prec, gn_stats=stats,
                 splitk_ws=ws if use_ws else None)
        got = G.from_act(out, Cout)
        err = G.maxdiff(got, ref) / max(1.0, float(ref.abs().max()))
        worst = max(worst, err)
        ok = bool(torch.isfinite(got).all()) and err <= 3e-5
        if gn:
            gg = ref.double().reshape(32, Cout // 32, -1)
            want = torch.stack([gg.sum((1, 2)), (gg * gg).sum((1, 2))], 1).flatten()
            ok = ok and float((stats.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
        if not ok:
            print("FAIL", tag, "err %.3e" % err)
            sys.exit(1)
        if it % 20 == 0:
            print("ok  ", tag, "err %.2e" % err)
    print("conv_fuzz: %d cases, worst relative error %.3e" % (args.n, worst))


if __name__ == "__main__":
    main()
