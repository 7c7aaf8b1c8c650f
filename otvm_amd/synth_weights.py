"""Deterministic synthetic checkpoint for the 785-key OTVM stage-4 ``state_dict``.

The trained ``weights/s4_OTVM.pth`` (reference README.md:57-67) is a Google-Drive download and is not
available offline, so every parity/bench run uses this generator.  It is pure numpy (PCG64 streams
seeded from crc32(key)), so this container and the GPU box produce bit-identical tensors.

The distributions are chosen to resemble a *trained* network's conditioning:
  * He-style fan-in scaling for plain / BN convs, so activations stay O(1);
  * the last norm of every residual branch gets a small gain (gamma ~ 0.2) so the identity path
    dominates, as in a converged ResNet;
  * prediction heads get gains/biases that give spread-out alpha in [0,1] and confident trimap
    logits, so that argmax near-ties (which the distance-transform encoding amplifies) are rare.
"""
import zlib

import numpy as np
import torch

from .state_spec import state_dict_spec

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _rng(key, seed):
    return np.random.Generator(np.random.PCG64([zlib.crc32(key.encode()), seed]))


def _is_residual_last_norm(key):
    # Bottleneck.bn3 / BasicBlock.bn2 (resnet_GN_WS.py:45,83; torchvision Bottleneck.bn3)
    parts = key.split(".")
    if parts[-2] == "bn3":
        return True
    if parts[-2] == "bn2" and "refine" in key:
        return True
    return False


def synthetic_state_dict(seed=0, dtype=torch.float32):
    spec = state_dict_spec()
    out = {}
    for key, (shape, dt) in spec.items():
        r = _rng(key, seed)
        leaf = key.split(".")[-1]
        if dt == "int64":
            out[key] = torch.zeros(shape, dtype=torch.int64)
            continue
        if key.endswith("IMG_MEAN") or key.endswith(".mean"):
            a = np.asarray(IMAGENET_MEAN, np.float32).reshape(shape)
        elif key.endswith("IMG_STD") or key.endswith(".std"):
            a = np.asarray(IMAGENET_STD, np.float32).reshape(shape)
        elif key == "LAPLOSS.KERNEL":
            g = np.array([1, 4, 6, 4, 1], np.float32)
            a = np.outer(g, g) / 256.0
        elif key == "trimap.LOSS.weight":
            a = np.ones(shape, np.float32)
        elif leaf == "running_mean":
            a = r.normal(0.0, 0.1, shape)
        elif leaf == "running_var":
            a = r.uniform(0.5, 1.5, shape)
        elif len(shape) == 1 and leaf == "weight":      # norm gamma
            if _is_residual_last_norm(key):
                a = r.uniform(0.1, 0.3, shape)
            else:
                a = r.uniform(0.5, 1.5, shape)
        elif len(shape) == 1 and leaf == "bias":
            a = _bias(key, shape, r)
        elif len(shape) == 4:
            a = _conv_weight(key, shape, r)
        else:
            raise KeyError(key)
        out[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dtype)
    return out


def _conv_weight(key, shape, r):
    cout, cin, kh, kw = shape
    fan_in = cin * kh * kw
    gain = np.sqrt(2.0 / fan_in)
    if key.startswith("NET.encoder") or ".ppm." in key or "conv_up1" in key or "conv_up2" in key \
            or "conv_up3" in key or key.startswith("NET.refine.conv1") or key.startswith("NET.refine.layer"):
        # weight-standardised: scale is removed by layers_WS.py:15-21; keep a non-zero filter mean
        return r.normal(0.02, 0.05, shape)
    if key.endswith("conv_up4.4.weight") or key.endswith("pred.4.weight"):
        w = r.normal(0.0, np.sqrt(1.0 / fan_in), shape)
        w[0] *= 0.6                      # alpha logit spread
        if cout == 10:
            w[7:] *= 6.0                 # trimap-refine logits: confident
            w[7:] -= w[7:].mean(axis=(1, 2, 3), keepdims=True)   # no common-mode class offset
        return w
    if "Encoder_M.conv1_" in key:
        return r.normal(0.0, gain * 0.5, shape)
    if ".Key." in key:
        return r.normal(0.0, 1.2 * np.sqrt(1.0 / fan_in), shape)   # attention logits std ~6 (peaky, not one-hot)
    if "Decoder.pred" in key:
        w = r.normal(0.0, 4.0 * np.sqrt(1.0 / fan_in), shape)
        return w - w.mean(axis=(1, 2, 3), keepdims=True)         # balanced classes on relu(m2) >= 0
    if "Decoder" in key and ("ResMM" in key or "ResFS" in key):
        return r.normal(0.0, 0.6 * gain, shape)
    return r.normal(0.0, gain, shape)


def _bias(key, shape, r):
    if key.endswith("conv_up4.4.bias") or key.endswith("pred.4.bias"):
        b = r.normal(0.0, 0.1, shape)
        b[0] = 0.5
        return b
    if key.startswith("NET") and (".bn" in key or ".1.bias" in key or ".2.bias" in key
                                   or ".4.bias" in key) and not key.endswith(("conv_up4.2.bias", "pred.2.bias")):
        return r.normal(0.0, 0.1, shape)
    return r.normal(0.0, 0.05, shape)
