"""Helpers for the -m gpu tests: NCHW(CPU) <-> NHWC(device) and thin wrappers over the C ABI."""
import ctypes as C

import numpy as np
import torch

from otvm_amd import lib as L
from otvm_amd.engine import Act, ConvW, _rup, pack_conv_weight, conv_params

DEV = "cuda:0"


def stream():
    return torch.cuda.current_stream().cuda_stream


def to_act(x_nchw, c_pad=None, ld=None, off=0):
    """[1,C,H,W] CPU tensor -> device NHWC Act (channels zero-padded to c_pad, optionally inside a wider buffer)."""
    _, Cc, H, W = x_nchw.shape
    c_pad = _rup(Cc, 4) if c_pad is None else c_pad
    ld = c_pad if ld is None else ld
    buf = torch.zeros(H * W * ld + off + 16, dtype=torch.float32)
    v = torch.as_strided(buf, (H, W, Cc), (W * ld, ld, 1), off)
    v.copy_(x_nchw[0].permute(1, 2, 0))
    return Act(buf.to(DEV), H, W, c_pad, ld, off)


def empty_act(H, W, Cc, ld=None, off=0, fill=float("nan")):
    ld = Cc if ld is None else ld
    buf = torch.full((H * W * ld + off + 16,), fill, dtype=torch.float32, device=DEV)
    return Act(buf, H, W, Cc, ld, off)


def from_act(a, Cc=None):
    """device Act -> [1,C,H,W] CPU tensor"""
    Cc = a.C if Cc is None else Cc
    v = torch.as_strided(a.t, (a.H, a.W, Cc), (a.W * a.ld, a.ld, 1), a.off)
    return v.permute(2, 0, 1)[None].cpu().contiguous()


def pack_weight(w, ws=False, scale=None, i_pad=None):
    from otvm_amd import engine
    # kernel tests cover the one-wave tile too (off by default in the product): the switch is flipped for THIS weight only and
    # restored, so tests collected later run the shipped configuration whatever the test order
    old = engine.WAVE_TILE
    engine.WAVE_TILE = True
    try:
        cw = pack_conv_weight(L.load(), DEV, w.contiguous().to(DEV), ws, None if scale is None else scale.contiguous().to(DEV),
                              i_pad, split=True, stream=stream())
    finally:
        engine.WAVE_TILE = old
    torch.cuda.synchronize()
    return cw


def conv2d(x, cw, out, bias=None, stride=1, pad=0, dil=1, act=0, in_relu=0, residual=None, precision=0, gn_stats=None,
           in_norm=None, splitk_ws=None):
    p = conv_params(x, cw, out, bias, stride, pad, dil, act, in_relu, residual, precision, in_norm, splitk_ws)
    if gn_stats is not None:
        p.gn_stats = gn_stats.data_ptr()
    L.check(L.load().otvm_conv2d(C.byref(p), stream()), "conv2d")
    torch.cuda.synchronize()


def maxdiff(a, b):
    return float((a - b).abs().max())
