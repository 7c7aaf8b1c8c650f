"""Stress of the table-by-the-last-workgroup path (otvm_conv_params.gn_scale_out): the same conv launched over and over,
every table compared bit for bit with otvm_gn_table over the finished statistics.  A workgroup whose statistics were
not yet visible when the last one read them would show up as a mismatch.

    python tools/gn_tail_stress.py [--reps 2000]
"""
import argparse
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_util as G  # noqa: E402
from otvm_amd import lib as L  # noqa: E402
from otvm_amd.engine import Act, conv_params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2000)
    args = ap.parse_args()
    lib = L.load()
    bad = 0
    g = torch.Generator().manual_seed(0)
    # (Cin, Cout, k, H, W): many workgroups on all eight XCDs, small and large tiles, the patch kernel
    for Cin, Cout, k, H, W in ((64, 256, 1, 272, 480), (256, 64, 1, 272, 480), (64, 64, 3, 544, 960), (512, 2048, 1, 136, 240), (256, 128, 1, 68, 120)):
        x = Act(torch.randn(H * W * Cin + 16, generator=g).to(G.DEV), H, W, Cin, Cin, 0)
        w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
        cw = G.pack_weight(w)
        out = G.empty_act(H, W, Cout, fill=0.0)
        gamma, beta = torch.rand(Cout, device=G.DEV) + 0.5, torch.randn(Cout, device=G.DEV) * 0.2
        counter = torch.zeros(1, dtype=torch.int32, device=G.DEV)
        stats = torch.zeros(64, dtype=torch.float64, device=G.DEV)
        tab, want = torch.zeros(2 * Cout, device=G.DEV), torch.zeros(2 * Cout, device=G.DEV)
        ws = torch.empty(8 << 20, device=G.DEV)
        p = conv_params(x, cw, out, None, 1, (k - 1) // 2, 1, 0, 0, None, 1, None, ws)
        p.gn_stats = stats.data_ptr()
        p.gn_gamma, p.gn_beta = gamma.data_ptr(), beta.data_ptr()
        p.gn_scale_out, p.gn_shift_out, p.gn_counter = tab.data_ptr(), tab.data_ptr() + 4 * Cout, counter.data_ptr()
        st = G.stream()
        layer_bad = 0
        for r in range(args.reps):
            stats.zero_()
            L.check(lib.otvm_conv2d(C.byref(p), st), "conv")
            L.check(lib.otvm_gn_table(stats.data_ptr(), H * W, Cout, gamma.data_ptr(), beta.data_ptr(), want.data_ptr(),
                                      want.data_ptr() + 4 * Cout, st))
            if not torch.equal(tab, want):
                layer_bad += 1
        torch.cuda.synchronize()
        print("%4d -> %4d k%d at %dx%d: %d launches, %d tables differ from otvm_gn_table" % (Cin, Cout, k, H, W, args.reps, layer_bad), flush=True)
        bad += layer_bad
    print("gn_tail_stress: %d mismatches" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
