"""Micro-benchmark of otvm_conv2d_head (the 32 -> 16 conv3x3 + 1x1 head + fba_fusion launch) alone on the device (tuning aid; GPU only).

    python tools/head_bench.py [--iters 30] [--size H,W ...]        (OTVM_HIP_LIB=<variant .so> for tools/build_variant.sh builds)

Two cases per size, as the frame issues them (engine.py: conv_up4.2 + head7 -> alpha only; pred.2 + head10 -> the hidden state into
the 24-channel Encoder_M input, alpha, the three class probabilities, three values of the same 24-channel rows).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--size", action="append")
    ap.add_argument("--wide16", type=int, default=1)
    args = ap.parse_args()
    sizes = [tuple(int(v) for v in s.split(",")) for s in args.size] if args.size else [(1088, 1920), (480, 832)]
    lib = L.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for (H, W) in sizes:
        P = H * W
        x = Act(torch.randn(P * 32, device=dev), H, W, 32, 32)
        w = torch.randn(16, 32, 3, 3, device=dev) / math.sqrt(32 * 9)
        cw = pack_conv_weight(lib, dev, w, split=True, stream=st)
        bias = torch.randn(16, device=dev) * 0.2
        img = Act(torch.rand(P * 12, device=dev), H, W, 4, 12)
        sm = Act(torch.zeros(P * 24, device=dev), H, W, 24, 24)
        alpha = torch.empty(P, device=dev)
        tri = torch.empty(3 * P, device=dev)
        for n_out in (7, 10):
            hw = (torch.randn(n_out, 16, device=dev) * 0.4).contiguous()
            hb = torch.randn(n_out, device=dev) * 0.3
            hid = sm.ch(0, 16)
            p = conv_params(x, cw, hid, bias, 1, 1, 1, 2, 0, None, 1)
            if n_out == 7:
                p.out, p.out_ld = 0, 0
            h = L.HeadParams()
            h.w, h.b, h.n_out, h.img, h.img_ld, h.P = hw.data_ptr(), hb.data_ptr(), n_out, img.ptr, img.ld, P
            h.alpha_out, h.alpha_stride = alpha.data_ptr(), 1
            if n_out == 10:
                h.tri_out, h.sm, h.sm_ld = tri.data_ptr(), sm.ch(16, 8).ptr, 24
            if args.wide16 and cw.w16 is not None:
                h.w16 = cw.w16.data_ptr()
            for _ in range(3):
                L.check(lib.otvm_conv2d_head(C.byref(p), C.byref(h), st), "conv2d_head")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                L.check(lib.otvm_conv2d_head(C.byref(p), C.byref(h), st), "conv2d_head")
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            rd = P * (128 + 12)
            wr = P * (4 if n_out == 7 else 4 + 12 + 12 + 64)
            print("head%-2d %4dx%-4d : %7.3f ms   algorithmic %4d MB read + %4d MB written = %5.2f TB/s" %
                  (n_out, H, W, ms, rd >> 20, wr >> 20, (rd + wr) / ms / 1e9))


if __name__ == "__main__":
    main()
