"""Digest of the spatially tiled kernels' outputs on ragged maps (GPU only): the tile walk of csrc/common.h (OTVM_TILE_WALK /
OTVM_TILE_BAND, read once per process) changes which workgroup computes which tile, never a tile's arithmetic, so the digests
of two processes with different walks must be equal.  tests/test_gpu_kernels.py::test_tile_walk_is_bit_identical runs it.

    OTVM_TILE_WALK=0 python tools/tile_walk_check.py        # prints "tile_walk_check: <n> outputs <sha256>"
"""
import ctypes as C
import hashlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from otvm_amd import lib as L                      # noqa: E402
from otvm_amd.engine import Act, pack_conv_weight, conv_params   # noqa: E402


def main():
    lib = L.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(11)
    h = hashlib.sha256()
    n = 0
    # (Cin, Cout, k, stride, dil, H, W, tune): patch tiles of every family (32 / 64 / 256 / 512 filters = one and two channel tiles per
    # position, dilated), the stem; heights that leave a short last band (band 8: 5, 13 and 21 tile rows; band 3: any)
    shapes = [(64, 32, 3, 1, 1, 1000, 300, 241), (80, 32, 3, 1, 1, 100, 260, 241), (64, 64, 3, 1, 1, 165, 250, 241),
              (32, 64, 3, 1, 2, 101, 97, 241), (256, 256, 3, 1, 1, 104, 200, 241), (64, 512, 3, 1, 1, 40, 70, 241),
              (4, 64, 7, 2, 1, 330, 410, 209), (24, 64, 7, 2, 1, 203, 131, 209)]
    for (Cin, Cout, k, stride, dil, H, W, tune) in shapes:
        pad = dil * (k - 1) // 2
        x = Act(torch.randn(H * W * Cin, generator=g).to(dev), H, W, Cin, Cin)
        w = torch.randn(Cout, Cin, k, k, generator=g).to(dev) / math.sqrt(Cin * k * k)
        cw = pack_conv_weight(lib, dev, w, split=True, stream=st)
        Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        out = Act(torch.zeros(Ho * Wo * Cout, device=dev), Ho, Wo, Cout, Cout)
        bias = torch.randn(Cout, generator=g).to(dev)
        p = conv_params(x, cw, out, bias, stride, pad, dil, 1, 0, None, 1)
        p.tune = tune
        L.check(lib.otvm_conv2d(C.byref(p), st), "conv2d")
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out.t).all()) and float(out.t.abs().max()) > 0
        h.update(out.t.cpu().numpy().tobytes())
        n += 1
    # the fused STM bottleneck (identity and projection forms) on a ragged map
    import gpu_util as G                                                   # tests/ helper: weight packing for the bottleneck
    for Cin, (H, W) in ((256, (75, 130)), (64, (90, 70))):
        proj = Cin == 64
        x = Act(torch.rand(H * W * Cin + 16, generator=g).to(dev), H, W, Cin, Cin, 0)
        w1 = torch.randn(64, Cin, 1, 1, generator=g) / math.sqrt(Cin)
        w2 = torch.randn(64, 64, 3, 3, generator=g) / 24
        w3 = torch.randn(256, 64, 1, 1, generator=g) / 8
        wd = torch.randn(256, Cin, 1, 1, generator=g) / math.sqrt(Cin)
        c1, c2, c3 = G.pack_weight(w1), G.pack_weight(w2), G.pack_weight(w3)
        cc = G.pack_weight(torch.cat([w3, wd], dim=1)) if proj else c3
        b64, b256 = torch.zeros(64, device=dev), torch.zeros(256, device=dev)
        out = G.empty_act(H, W, 256, fill=0.0)
        q = L.StmBottleneckParams(x.ptr, H, W, Cin, x.ld, out.ptr, out.ld, c1.w_wfrag.data_ptr(), c2.w_wfrag.data_ptr(),
                                  cc.w_wfrag.data_ptr(), c1.w_scale.data_ptr(), c2.w_scale.data_ptr(), cc.w_scale.data_ptr(),
                                  b64.data_ptr(), b64.data_ptr(), b256.data_ptr(), 1, 0, 0)
        L.check(lib.otvm_stm_bottleneck_f16x3(C.byref(q), G.stream()), "fused bottleneck")
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out.t).all()) and float(out.t.abs().max()) > 0
        h.update(out.t.cpu().numpy().tobytes())
        n += 1
    # the 16-wide head conv (32 -> 16 + the 10-output head in its epilogue) on a ragged map
    H, W = 93, 150
    P = H * W
    x = Act(torch.randn(P * 32, generator=g).to(dev), H, W, 32, 32)
    cw = pack_conv_weight(lib, dev, torch.randn(16, 32, 3, 3, generator=g).to(dev) / math.sqrt(32 * 9), split=True, stream=st)
    bias = (torch.randn(16, generator=g) * 0.2).to(dev)
    img = Act(torch.rand(P * 12, generator=g).to(dev), H, W, 4, 12)
    sm = Act(torch.zeros(P * 24, device=dev), H, W, 24, 24)
    alpha, tri = torch.zeros(P, device=dev), torch.zeros(3 * P, device=dev)
    hw, hb = (torch.randn(10, 16, generator=g) * 0.4).to(dev).contiguous(), (torch.randn(10, generator=g) * 0.3).to(dev)
    p = conv_params(x, cw, sm.ch(0, 16), bias, 1, 1, 1, 2, 0, None, 1)
    hd = L.HeadParams()
    hd.w, hd.b, hd.n_out, hd.img, hd.img_ld, hd.P = hw.data_ptr(), hb.data_ptr(), 10, img.ptr, img.ld, P
    hd.alpha_out, hd.alpha_stride = alpha.data_ptr(), 1
    hd.tri_out, hd.sm, hd.sm_ld = tri.data_ptr(), sm.ch(16, 8).ptr, 24
    assert cw.w16 is not None
    hd.w16 = cw.w16.data_ptr()
    L.check(lib.otvm_conv2d_head(C.byref(p), C.byref(hd), st), "conv2d_head")
    torch.cuda.synchronize()
    for t in (sm.t, alpha, tri):
        assert bool(torch.isfinite(t).all())
        h.update(t.cpu().numpy().tobytes())
    n += 1
    print("tile_walk_check: %d outputs %s" % (n, h.hexdigest()))


if __name__ == "__main__":
    main()
