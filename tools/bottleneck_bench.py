"""Times one STM 1/4-resolution bottleneck as the fused kernel and as the three (four) convolution launches.

    python tools/bottleneck_bench.py [--height 272 --width 480] [--reps 50]
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


def timeit(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def planes128(lib, H, W, reps, g):
    """res3.1 - res3.3 of the STM encoders: the fused kernel on each pixel tile against the three convolution launches (the
    plan-time tuner's configuration of each, as the frame runs them)."""
    from otvm_amd.engine import FramePlan, _TUNE_CACHE
    x = Act(torch.rand(H * W * 512 + 16, device=G.DEV), H, W, 512, 512, 0)
    w1 = torch.randn(128, 512, 1, 1, generator=g) / math.sqrt(512)
    w2 = torch.randn(128, 128, 3, 3, generator=g) / math.sqrt(128 * 9)
    w3 = torch.randn(512, 128, 1, 1, generator=g) / math.sqrt(128)
    c1, c2, c3 = G.pack_weight(w1), G.pack_weight(w2), G.pack_weight(w3)
    b128, b512 = torch.zeros(128, device=G.DEV), torch.zeros(512, device=G.DEV)
    t1, t2 = G.empty_act(H, W, 128, fill=0.0), G.empty_act(H, W, 128, fill=0.0)
    out, out2 = G.empty_act(H, W, 512, fill=0.0), G.empty_act(H, W, 512, fill=0.0)
    ws = torch.empty(16 << 20, device=G.DEV)
    st = G.stream()
    ps = [conv_params(x, c1, t1, b128, 1, 0, 1, 1, 0, None, 1, None, ws), conv_params(t1, c2, t2, b128, 1, 1, 1, 1, 0, None, 1, None, ws),
          conv_params(t2, c3, out, b512, 1, 0, 1, 1, 0, x, 1, None, ws)]
    # the tuner's choice per launch (what the frame would run)
    codes = (C.c_int * 128)()
    for p_ in ps:
        n = int(lib.otvm_conv2d_candidates(C.byref(p_), codes, 128))
        best, bt = 0, None
        for c in [0] + [int(codes[i]) for i in range(n)]:
            if c // 16 - 1 == 9:
                continue
            p_.tune = c
            t = timeit(lambda: lib.otvm_conv2d(C.byref(p_), st), 10)
            if bt is None or t < bt:
                best, bt = c, t
        p_.tune = best

    def unfused():
        for p_ in ps:
            lib.otvm_conv2d(C.byref(p_), st)
    tu = timeit(unfused, reps)
    each = [timeit(lambda p_=p_: lib.otvm_conv2d(C.byref(p_), st), reps) for p_ in ps]
    fl = 2 * H * W * (512 * 128 + 9 * 128 * 128 + 128 * 512)
    print("planes-128 identity block at %dx%d: three launches %.1f us (%s; tune codes %s) = %.0f TFLOP/s"
          % (H, W, tu, " + ".join("%.1f" % e for e in each), [p_.tune for p_ in ps], fl / tu / 1e6), flush=True)
    unfused()
    torch.cuda.synchronize()
    for tile, name in ((1, "8x16"), (2, "8x8"), (3, "4x8")):
        q = L.StmBottleneckParams(x.ptr, H, W, 512, x.ld, out2.ptr, out2.ld, c1.w_wfrag.data_ptr(), c2.w_wfrag.data_ptr(),
                                  c3.w_wfrag.data_ptr(), c1.w_scale.data_ptr(), c2.w_scale.data_ptr(), c3.w_scale.data_ptr(),
                                  b128.data_ptr(), b128.data_ptr(), b512.data_ptr(), 1, 0, 0, tile)
        tf = timeit(lambda: L.check(lib.otvm_stm_bottleneck_f16x3(C.byref(q), st), "fused"), reps)
        torch.cuda.synchronize()
        d = float((out.torch() - out2.torch()).abs().max())
        th, tw = {1: (8, 16), 2: (8, 8), 3: (4, 8), 4: (8, 16), 5: (8, 8)}[tile]
        nwg = ((H + th - 1) // th) * ((W + tw - 1) // tw)
        print("   fused, tile %-7s: %7.1f us (%.0f TFLOP/s), %5d workgroups | x%.2f | max-abs diff vs the launches %.2e"
              % (name, tf, fl / tf / 1e6, nwg, tu / tf, d), flush=True)
        try:
            dbg = lib.otvm_debug_bnk128_times                    # experiment build (-DOTVM_BNK_TIMING): per-stage time of wave 0
        except AttributeError:
            dbg = None
        if dbg is not None:
            import numpy as np
            t8 = np.zeros(8, dtype=np.uint64)
            dbg(None, 1)
            L.check(lib.otvm_stm_bottleneck_f16x3(C.byref(q), st), "fused")
            torch.cuda.synchronize()
            dbg(t8.ctypes.data_as(C.c_void_p), 0)
            names = ["A loop", "t1->LDS", "B loop", "t2->LDS", "C0 gemm", "C0 epi", "C1 gemm", "C1 epi"]
            print("      per workgroup (wave 0, us): " + ", ".join("%s %.2f" % (n_, t / nwg / 100.0) for n_, t in zip(names, t8)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=272)
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--planes128", action="store_true", help="the 1/8-resolution identity block (512 -> 128 -> 128 -> 512) instead")
    args = ap.parse_args()
    lib = L.load()
    H, W = args.height, args.width
    g = torch.Generator().manual_seed(0)
    if args.planes128:
        return planes128(lib, H, W, args.reps, g)
    for Cin in (256, 64):
        proj = Cin == 64
        x = Act(torch.rand(H * W * Cin + 16, device=G.DEV), H, W, Cin, Cin, 0)
        w1 = torch.randn(64, Cin, 1, 1, generator=g) / math.sqrt(Cin)
        w2 = torch.randn(64, 64, 3, 3, generator=g) / 24
        w3 = torch.randn(256, 64, 1, 1, generator=g) / 8
        wd = torch.randn(256, Cin, 1, 1, generator=g) / math.sqrt(Cin)
        c1, c2, c3, cd = G.pack_weight(w1), G.pack_weight(w2), G.pack_weight(w3), G.pack_weight(wd)
        cc = G.pack_weight(torch.cat([w3, wd], dim=1)) if proj else c3
        b64, b256 = torch.zeros(64, device=G.DEV), torch.zeros(256, device=G.DEV)
        t1, t2 = G.empty_act(H, W, 64, fill=0.0), G.empty_act(H, W, 64, fill=0.0)
        idt, out, out2 = G.empty_act(H, W, 256, fill=0.0), G.empty_act(H, W, 256, fill=0.0), G.empty_act(H, W, 256, fill=0.0)
        ws = torch.empty(8 << 20, device=G.DEV)
        st = G.stream()
        ps = [conv_params(x, c1, t1, b64, 1, 0, 1, 1, 0, None, 1, None, ws), conv_params(t1, c2, t2, b64, 1, 1, 1, 1, 0, None, 1, None, ws)]
        if proj:
            ps.append(conv_params(x, cd, idt, b256, 1, 0, 1, 0, 0, None, 1, None, ws))
        ps.append(conv_params(t2, c3, out, b256, 1, 0, 1, 1, 0, idt if proj else x, 1, None, ws))

        def unfused():
            for p in ps:
                lib.otvm_conv2d(C.byref(p), st)
        q = L.StmBottleneckParams(x.ptr, H, W, Cin, x.ld, out2.ptr, out2.ld, c1.w_wfrag.data_ptr(), c2.w_wfrag.data_ptr(),
                                  cc.w_wfrag.data_ptr(), c1.w_scale.data_ptr(), c2.w_scale.data_ptr(), cc.w_scale.data_ptr(),
                                  b64.data_ptr(), b64.data_ptr(), b256.data_ptr(), 1, 0, 0, 0)

        def fused():
            L.check(lib.otvm_stm_bottleneck_f16x3(C.byref(q), st), "fused")
        tu, tf = timeit(unfused, args.reps), timeit(fused, args.reps)
        if hasattr(lib, "otvm_debug_bnk_times"):                 # experiment build (-DOTVM_BNK_TIMING): per-stage time of wave 0
            import numpy as np
            t8 = np.zeros(8, dtype=np.uint64)
            lib.otvm_debug_bnk_times(None, 1)
            fused()
            torch.cuda.synchronize()
            lib.otvm_debug_bnk_times(t8.ctypes.data_as(C.c_void_p), 0)
            nwg = ((H + 7) // 8) * ((W + 31) // 32)
            names = ["A conv1", "B conv2", "t2->LDS", "C0 gemm", "C0 epilogue", "C1 gemm", "C1 epilogue"]
            print("   per workgroup (wave 0, us): " + ", ".join("%s %.2f" % (n, t / nwg / 100.0) for n, t in zip(names, t8)), flush=True)
        d = float((out.torch() - out2.torch()).abs().max())
        P = H * W
        fl = 2 * P * (Cin * 64 + 9 * 64 * 64 + 64 * 256 + (Cin * 256 if proj else 0))
        by = 4 * P * (Cin + 256)
        print("bottleneck Cin %3d at %dx%d: %d launches %7.1f us | fused %7.1f us (%.0f TFLOP/s, %.2f TB/s algorithmic) | x%.2f | max-abs diff %.2e"
              % (Cin, H, W, len(ps), tu, tf, fl / tf / 1e6, by / tf / 1e6, tu / tf, d), flush=True)


if __name__ == "__main__":
    main()
