"""Helpers for the -m gpu tests: NCHW(CPU) <-> NHWC(device) and thin wrappers over the C ABI."""
import ctypes as C

import numpy as np
import torch

from otvm_amd import lib as L
from otvm_amd.engine import Act, ConvW, _rup

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
    lib = L.load()
    O, I, kh, kw = w.shape
    cw = ConvW()
    cw.O, cw.kh, cw.kw = O, kh, kw
    cw.I_pad = _rup(I, 4) if i_pad is None else i_pad
    cw.K_pad = _rup(kh * kw * cw.I_pad, 32)
    O_pad = _rup(O, 128)
    cw.w = torch.empty(O_pad * cw.K_pad, dtype=torch.float32, device=DEV)
    cw.bias = None
    wd = w.contiguous().to(DEV)
    sc = None if scale is None else scale.contiguous().to(DEV)
    L.check(lib.otvm_pack_conv_weight(wd.data_ptr(), O, I, kh, kw, 1 if ws else 0, 0 if sc is None else sc.data_ptr(),
                                      cw.w.data_ptr(), O_pad, cw.I_pad, cw.K_pad, stream()), "pack")
    torch.cuda.synchronize()
    return cw


def conv2d(x, cw, out, bias=None, stride=1, pad=0, dil=1, act=0, in_relu=0, residual=None):
    lib = L.load()
    Ho = (x.H + 2 * pad - dil * (cw.kh - 1) - 1) // stride + 1
    Wo = (x.W + 2 * pad - dil * (cw.kw - 1) - 1) // stride + 1
    p = L.ConvParams(x.ptr, x.H, x.W, x.C, x.ld, cw.w.data_ptr(), cw.K_pad, 0 if bias is None else bias.data_ptr(),
                     0 if residual is None else residual.ptr, 0 if residual is None else residual.ld,
                     out.ptr, Ho, Wo, cw.O, out.ld, cw.kh, cw.kw, stride, pad, dil, in_relu, act)
    L.check(lib.otvm_conv2d(C.byref(p), stream()), "conv2d")
    torch.cuda.synchronize()


def maxdiff(a, b):
    return float((a - b).abs().max())
