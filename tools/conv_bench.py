"""Micro-benchmark of otvm_conv2d on single layer shapes (tuning aid; GPU only).

    python tools/conv_bench.py [--prec 1] [--iters 20] [--shape Cin,Cout,k,stride,dil,H,W ...]
"""
import argparse
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otvm_amd import lib as L                      # noqa: E402
from otvm_amd.engine import Act, pack_conv_weight, conv_params   # noqa: E402

DEFAULT = [
    (256, 256, 3, 1, 1, 272, 480),     # STM decoder RF2 / conv_up2-like, OS4
    (3072, 256, 3, 1, 1, 136, 240),    # conv_up1.0
    (512, 512, 3, 1, 4, 136, 240),     # layer4 conv2 (dilated)
    (64, 64, 3, 1, 1, 1088, 1920),     # refine 64->64 full res
    (1024, 256, 1, 1, 1, 136, 240),    # layer3 conv1 1x1
    (256, 1024, 1, 1, 1, 136, 240),    # layer3 conv3 1x1
    (64, 256, 1, 1, 1, 272, 480),      # res2 conv3 1x1
    (32, 16, 3, 1, 1, 1088, 1920),     # head 32->16
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prec", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--shape", action="append")
    ap.add_argument("--relu", type=int, default=0)
    ap.add_argument("--res", type=int, default=0)
    ap.add_argument("--gn", type=int, default=0, help="fused GroupNorm statistics in the epilogue")
    ap.add_argument("--bias", type=int, default=0)
    ap.add_argument("--pad-ld", type=int, default=0, help="extra floats per pixel row of the input / output / residual views (ld = C + pad)")
    ap.add_argument("--zero", type=int, default=0, help="power probe: 1 = all-zero input, 2 = all-zero weights, 3 = both (the matrix cores' "
                                                        "power depends on the operands; times only)")
    ap.add_argument("--tune", default="0", help="otvm_conv_params.tune codes to time, comma separated; 'all' = every candidate")
    args = ap.parse_args()
    shapes = [tuple(int(v) for v in s.split(",")) for s in args.shape] if args.shape else DEFAULT
    lib = L.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for (Cin, Cout, k, stride, dil, H, W) in shapes:
        pad = dil * (k - 1) // 2
        pl = args.pad_ld
        x = Act(torch.randn(H * W * (Cin + pl), device=dev), H, W, Cin, Cin + pl)
        w = torch.randn(Cout, Cin, k, k, device=dev) / math.sqrt(Cin * k * k)
        if args.zero & 1:
            x.t.zero_()
        if args.zero & 2:
            w.zero_()
        cw = pack_conv_weight(lib, dev, w, split=True, stream=st)
        if cw.w_wfrag is None and cw.I_pad % 32 == 0 and k * k <= 32:       # narrow layers: the one-wave tile's copy, for --tune all
            O_pad = cw.w_hi.numel() // cw.K_pad
            cw.w_wfrag = torch.zeros(int(lib.otvm_wave_weight_bytes_f16x3(O_pad, cw.K_pad)), dtype=torch.uint8, device=dev)
            L.check(lib.otvm_pack_wave_weight_f16x3(cw.w_hi.data_ptr(), cw.w_lo.data_ptr(), O_pad, cw.K_pad, cw.w_wfrag.data_ptr(), st))
        Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        out = Act(torch.empty(Ho * Wo * (max(4, Cout) + pl), device=dev), Ho, Wo, Cout, max(4, Cout) + pl)
        res = Act(torch.randn(Ho * Wo * (Cout + pl), device=dev), Ho, Wo, Cout, Cout + pl) if args.res else None
        bias = torch.randn(Cout, device=dev) if args.bias else None
        p = conv_params(x, cw, out, bias, stride, pad, dil, 0, args.relu, res, args.prec)
        stats = torch.zeros(64, dtype=torch.float64, device=dev)
        if args.gn:
            p.gn_stats = stats.data_ptr()
        ws = torch.empty(16 << 20, device=dev)
        p.splitk_ws, p.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
        if args.tune == "all":
            codes = (C.c_int * 128)()
            n = int(lib.otvm_conv2d_candidates(C.byref(p), codes, 128))
            tunes = [0] + [int(codes[i]) for i in range(n)]
        else:
            tunes = [int(v) for v in args.tune.split(",")]
        names = {0: "256x256", 1: "256x128", 2: "128x128", 3: "128x64", 4: "64x64", 5: "256x64", 6: "256x32", 7: "256x128w4", 8: "128x256w4", 9: "wave64", 10: "64x64D", 11: "128x64D", 12: "stem", 13: "256x256w4", 14: "patch"}
        names.update({32 + t: n + "G" for t, n in list(names.items()) if t in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11)})     # LDS-DMA weight stages
        names.update({64 + t: n + "M" for t, n in list(names.items()) if t in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11)})     # ... on 16x16x32 MFMAs
        for tune in tunes:
            p.tune = tune
            for _ in range(3):
                L.check(lib.otvm_conv2d(C.byref(p), st))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                L.check(lib.otvm_conv2d(C.byref(p), st))
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            fl = 2.0 * Ho * Wo * Cout * k * k * Cin
            label = "heuristic" if tune == 0 else "%s/S%d" % (names.get(tune // 16 - 1, "?"), tune & 15)
            print("Cin %4d Cout %4d k%d s%d d%d %4dx%-4d %-14s: %7.3f ms  %7.1f TFLOP/s" %
                  (Cin, Cout, k, stride, dil, H, W, label, ms, fl / ms / 1e9))


if __name__ == "__main__":
    main()
