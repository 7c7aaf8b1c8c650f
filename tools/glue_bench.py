"""Isolated timings of the serial glue of a 1080p frame (nothing else on the device): the trimap-encode chain
(classify + column pass + row pass / encoding), the PPM chain (pooling, heads, Z table, gather), the decoder tail's softmax,
the heads and the preprocess -- HIP events around each library call, median of N.

    python tools/glue_bench.py [--height 1080 --width 1920 --reps 30]

Run from the repository root (or from a copy of an older tree, e.g. `cd _old && python ../tools/glue_bench.py`: the package
of the current directory is the one imported, so two trees can be compared on one box).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from otvm_amd import lib as L  # noqa: E402
from otvm_amd.engine import pad_amounts  # noqa: E402


def timed(fn, reps):
    st = torch.cuda.current_stream()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000.0)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--probe", action="store_true", help="also time strided partial writes of several widths")
    args = ap.parse_args()
    lib = L.load()
    dev = torch.device("cuda", 0)
    H, W = args.height, args.width
    lw, uw, lh, uh = pad_amounts(H, W, 32)
    Hp, Wp = H + lh + uh, W + lw + uw
    P = Hp * Wp
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(1)
    res = {"padded": [Hp, Wp], "abi": L.ABI_VERSION}

    # ---- trimap encode: noise-like classes (what random weights propagate) and a disc (what a real clip looks like)
    x11 = torch.zeros(P * 12, device=dev)
    d80 = torch.zeros(P * 80, device=dev)
    cls = torch.empty(P, dtype=torch.uint8, device=dev)
    ws = torch.empty(int(lib.otvm_trimap_encode_ws_bytes(Hp, Wp)), dtype=torch.uint8, device=dev)
    yy, xx = torch.meshgrid(torch.arange(Hp, device=dev), torch.arange(Wp, device=dev), indexing="ij")
    r = ((yy - Hp / 2) ** 2 + (xx - Wp / 2) ** 2).float().sqrt()
    disc = torch.stack([(r >= Hp / 3).float(), ((r < Hp / 3) & (r >= Hp / 4)).float(), (r < Hp / 4).float()]).contiguous()
    noise = torch.softmax(torch.randn(3, Hp, Wp, generator=g, device=dev) * 2, 0).contiguous()
    smooth = torch.softmax(torch.nn.functional.interpolate(torch.randn(1, 3, Hp // 32, Wp // 32, generator=g, device=dev) * 4,
                                                           size=(Hp, Wp), mode="bilinear")[0], 0).contiguous()
    for name, pr in (("noise", noise), ("blobs", smooth), ("disc", disc)):
        res["trimap_encode_us[%s]" % name] = timed(
            lambda: L.check(lib.otvm_trimap_encode(pr.data_ptr(), Hp, Wp, 0, cls.data_ptr(), x11.data_ptr(), 12, d80.data_ptr(), 80,
                                                   ws.data_ptr(), st)), args.reps)
    # ---- upsample4 + softmax (STM decoder tail)
    l4 = torch.randn((Hp // 4) * (Wp // 4) * 4, generator=g, device=dev)
    probs = torch.empty(3 * P, device=dev)
    res["upsample4_softmax3_us"] = timed(lambda: L.check(lib.otvm_upsample4_softmax3(l4.data_ptr(), Hp // 4, Wp // 4, 4, probs.data_ptr(), st)),
                                         args.reps)
    # ---- PPM chain
    H8, W8 = Hp // 8, Wp // 8
    l4m = torch.randn(H8 * W8 * 2048, generator=g, device=dev)
    pooled = torch.empty(50 * 2048, device=dev)
    pws = torch.empty(int(lib.otvm_ppm_pool_ws_bytes(H8, 2048)), dtype=torch.uint8, device=dev)
    res["ppm_pool_us"] = timed(lambda: L.check(lib.otvm_ppm_pool(l4m.data_ptr(), H8, W8, 2048, 2048, pooled.data_ptr(), pws.data_ptr(), st)),
                               args.reps)
    hp = L.PpmHeadParams()
    keep = []
    for i, s_ in enumerate((1, 2, 3, 6)):
        w = (torch.randn(256 * 2048, generator=g, device=dev) / math.sqrt(2048)).contiguous()
        b, ga, be = torch.randn(256, device=dev) * 0.1, torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
        o = torch.empty(s_ * s_ * 256, device=dev)
        keep += [w, b, ga, be, o]
        hp.w[i], hp.bias[i], hp.gamma[i], hp.beta[i], hp.out[i] = w.data_ptr(), b.data_ptr(), ga.data_ptr(), be.data_ptr(), o.data_ptr()
    hp.pooled, hp.C, hp.K_pad, hp.Cout, hp.out_ld, hp.act = pooled.data_ptr(), 2048, 2048, 256, 256, 2
    res["ppm_head_us"] = timed(lambda: L.check(lib.otvm_ppm_head(C.byref(hp), st)), args.reps)
    wppm = (torch.randn(4 * 9 * 256 * 256, generator=g, device=dev) / 48).contiguous()
    Z = torch.empty(9 * 50 * 256, device=dev)
    yp = (C.c_void_p * 4)(*[keep[5 * i + 4].data_ptr() for i in range(4)])
    res["ppm_conv_z_us"] = timed(lambda: L.check(lib.otvm_ppm_conv_z(yp, 256, wppm.data_ptr(), Z.data_ptr(), st)), args.reps)
    u1 = torch.randn(H8 * W8 * 256, generator=g, device=dev)
    stats = torch.zeros(64, dtype=torch.float64, device=dev)
    res["ppm_conv_add_us"] = timed(lambda: L.check(lib.otvm_ppm_conv_add(Z.data_ptr(), H8, W8, u1.data_ptr(), 256, stats.data_ptr(), st)),
                                   args.reps)
    # ---- heads
    hid = torch.randn(P * 16, generator=g, device=dev)
    w7, b7 = torch.randn(7 * 16, device=dev) * 0.3, torch.randn(7, device=dev) * 0.1
    w10, b10 = torch.randn(10 * 16, device=dev) * 0.3, torch.randn(10, device=dev) * 0.1
    alpha_p, tri_p, sm = torch.empty(P, device=dev), torch.empty(3 * P, device=dev), torch.zeros(P * 24, device=dev)
    res["fba_head7_us"] = timed(lambda: L.check(lib.otvm_fba_head(hid.data_ptr(), 16, w7.data_ptr(), b7.data_ptr(), 7, d80.data_ptr() + 4 * 67, 80, P,
                                                                  d80.data_ptr() + 4 * 72, 80, 0, 0, 0, st)), args.reps)
    res["fba_head10_us"] = timed(lambda: L.check(lib.otvm_fba_head(hid.data_ptr(), 16, w10.data_ptr(), b10.data_ptr(), 10, d80.data_ptr() + 4 * 67, 80,
                                                                   P, alpha_p.data_ptr(), 1, tri_p.data_ptr(), sm.data_ptr() + 64, 24, st)), args.reps)
    # ---- preprocess (uint8 frames, all destinations)
    fg = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev)
    a = torch.ones(H * W, device=dev)
    pp = L.PreprocessParams()
    pp.a, pp.fg_u8, pp.bg_u8, pp.u8_rgb = a.data_ptr(), fg.data_ptr(), fg.data_ptr(), 0
    pp.H, pp.W, pp.Hp, pp.Wp, pp.lh, pp.lw = H, W, Hp, Wp, lh, lw
    for n_ in ("mean", "mean_q", "mean_m"):
        setattr(pp, n_, (C.c_float * 3)(0.485, 0.456, 0.406))
    for n_ in ("std", "std_q", "std_m"):
        setattr(pp, n_, (C.c_float * 3)(0.229, 0.224, 0.225))
    si = torch.empty(3 * H * W, device=dev)
    sq = torch.empty(P * 4, device=dev)
    pp.scaled_imgs, pp.x11, pp.x11_ld, pp.sq, pp.sq_ld = si.data_ptr(), x11.data_ptr(), 12, sq.data_ptr(), 4
    pp.sm, pp.sm_ld, pp.d80, pp.d80_ld = sm.data_ptr() + 64, 24, d80.data_ptr(), 80
    res["preprocess_us"] = timed(lambda: L.check(lib.otvm_preprocess(C.byref(pp), st)), args.reps)
    # ---- what a strided partial write costs (torch copy kernels; [P][80] fp32 rows of 320 bytes, like the D80 buffer): `w`
    # floats written per row at a 32-byte aligned offset -- is a 32-byte sector written without a read-modify-write?
    if args.probe:
        big = torch.zeros(P * 80, device=dev)
        v2 = big.view(P, 80)
        for w_ in (1, 2, 4, 8, 16, 32, 80):
            src = torch.randn(P, w_, device=dev)
            res["probe_write_%dB_per_320B_us" % (4 * w_)] = timed(lambda: v2[:, 64 - (64 if w_ == 80 else 0):64 - (64 if w_ == 80 else 0) + w_].copy_(src)
                                                                if w_ <= 16 or w_ == 80 else v2[:, 32:32 + w_].copy_(src), args.reps)
        for w_ in (1, 2, 4, 8, 12):
            big12 = torch.zeros(P * 12, device=dev).view(P, 12)
            src = torch.randn(P, w_, device=dev)
            res["probe_write_%dB_per_48B_us" % (4 * w_)] = timed(lambda: big12[:, 12 - w_:].copy_(src), args.reps)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
