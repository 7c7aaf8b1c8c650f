"""CPU oracle of the matting metrics the reference defines (orphan module utils/tmp/metric.py).  TEST INFRASTRUCTURE.

SAD  (metric.py:177-182): sum(|pred-target|/255 * mask) / 1000          per frame
MSE  (metric.py:184-189): sum(((pred-target)/255)^2 * mask) / (sum(mask) + 1)
dtSSD(metric.py:252-264): sqrt(sum(((p1-p0)-(t1-t0))^2 * mask0)) with p,t scaled by 1/255; count = sum(mask0) + 1
Inputs are 0..255-scale images [B,H,W] (B = frames of one clip), mask in {0,1}.
Pinned by fixtures produced with the reference's own functions (tests/golden/make_golden.py::metric_fixtures).
"""
import torch


def sad(pred, target, mask):
    err = (pred - target).abs() / 255.0
    return (err * mask).reshape(target.shape[0], -1).sum(-1) / 1000.0


def mse(pred, target, mask):
    err = (pred - target) / 255.0
    B = target.shape[0]
    return (err.pow(2) * mask).reshape(B, -1).sum(-1) / (mask.reshape(B, -1).sum(-1) + 1.0)


def dtssd(pred, target, mask):
    p, t = pred / 255.0, target / 255.0
    e = ((p[1:] - p[:-1]) - (t[1:] - t[:-1])).pow(2)
    m0 = mask[:-1]
    n = m0.shape[0]
    return (e * m0).reshape(n, -1).sum(1).sqrt(), m0.reshape(n, -1).sum(1) + 1.0
