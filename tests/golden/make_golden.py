"""Generate the golden fixtures by running the REFERENCE itself (development container only).

    python -m tests.golden.make_golden

Imports /root/reference through tools/ref_import.py (four shims), loads the
deterministic synthetic checkpoint (otvm_amd.synth_weights), drives ``EvalModel.forward`` over
seeded synthetic clips exactly as eval.py:157-228 does, and stores inputs-as-seeds + outputs.
The fixtures are data only; nothing of the reference's source travels with them.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tools.ref_import import build_reference_model, load_reference  # noqa: E402
from tools.ref_run import frame_inputs  # noqa: E402
from otvm_amd.synth_weights import synthetic_state_dict  # noqa: E402
from otvm_amd.synth_data import synthetic_clip, soft_alpha  # noqa: E402

SEQUENCES = [
    # name, H, W, T, style, skip, max_num, dilate_kernel, clip_seed
    ("demo_100x150_s5m5", 100, 150, 8, "demo", 5, 5, 12, 1),
    ("demo_64x96_s3m3", 64, 96, 12, "demo", 3, 3, 12, 2),
    ("v108_64x96_s3m3", 64, 96, 7, "v108", 3, 3, 5, 3),
    ("demo_64x64_m0", 64, 64, 4, "demo", 3, 0, 12, 4),
    ("demo_64x64_m1", 64, 64, 4, "demo", 3, 1, 12, 4),
    ("demo_64x64_m2_noskip", 64, 64, 5, "demo", 2, 2, 12, 4),
    ("demo_70x90_single", 70, 90, 1, "demo", 5, 5, 12, 5),
]


def run_sequence(H, W, T, style, skip, max_num, dk, clip_seed, wseed=0):
    m = build_reference_model(dk)
    m.load_state_dict(synthetic_state_dict(wseed), strict=True)
    frames, tri = synthetic_clip(H, W, T, clip_seed)
    alphas, tris, banks, keys0 = [], [], [], []
    for t in range(T):
        if style == "demo":
            a, fg, bg, tri_gt = frame_inputs(frames, t, trimap=tri)
        else:
            a, fg, bg, tri_gt = frame_inputs(frames, t, trimap=None, alpha=soft_alpha(H, W, t))
        memorize = (t % skip == 0) if skip > 2 else False          # eval.py:188-189
        out = m(a, fg, bg, tri=None, tri_gt=tri_gt, first_frame=(t == 0), last_frame=(t == T - 1),
                memorize=memorize, max_memory_num=max_num, large_input=False)
        alphas.append(out[3][0, 0, 0].numpy().copy())
        tris.append(out[1][0, 0].numpy().copy())
        k = m.memories["key"]
        banks.append(0 if k is None else int(k.shape[3]))
        # checksum of the bank keys: pins the slot policy (which frames are resident)
        keys0.append(np.zeros(0, np.float32) if k is None else k[0, 0, :8, :, 0, 0].numpy().copy().ravel())
    return dict(alpha=np.stack(alphas), trimap=np.stack(tris), bank=np.asarray(banks),
                tri_gt=out[2][0, 0].numpy().copy(), key_probe=np.concatenate(keys0))


def stage_fixture(H=64, W=64, clip_seed=6, dk=12):
    """Per-stage tensors of the REFERENCE (forward hooks) for two consecutive frames."""
    m = build_reference_model(dk)
    m.load_state_dict(synthetic_state_dict(0), strict=True)
    cap = {}
    stm = m.trimap.model

    def keep(name):
        def hook(mod, inp, out):
            cap.setdefault(name, []).append(out)
        return hook
    stm.Encoder_Q.register_forward_hook(keep("enc_q"))
    stm.Encoder_M.register_forward_hook(keep("enc_m"))
    stm.KV_Q_r4.register_forward_hook(keep("kv_q"))
    stm.KV_M_r4.register_forward_hook(keep("kv_m"))
    stm.Memory.register_forward_hook(keep("mem"))
    stm.Decoder.register_forward_hook(keep("dec"))
    def enc_hook(mod, i, o):                       # (a hook must return None, or it replaces the output)
        cap.setdefault("x11", []).append(i[0])
        cap.setdefault("feats", []).append(o[0])
    m.NET.encoder.register_forward_hook(enc_hook)
    m.NET.decoder.register_forward_hook(keep("fba_dec"))
    m.NET.refine.register_forward_hook(keep("fba_ref"))
    frames, tri = synthetic_clip(H, W, 2, clip_seed)
    for t in range(2):
        a, fg, bg, tri_gt = frame_inputs(frames, t, trimap=tri)
        m(a, fg, bg, tri=None, tri_gt=tri_gt, first_frame=(t == 0), last_frame=False, memorize=(t == 0), max_memory_num=5)
    f = lambda x: x.detach().numpy().copy()
    out = dict(H=H, W=W, clip_seed=clip_seed, dk=dk)
    out["r4_q"], out["r3_q"], out["r2_q"] = (f(v) for v in cap["enc_q"][0][:3])
    out["k4"], out["v4"] = (f(v) for v in cap["kv_q"][0])
    out["m4"] = f(cap["mem"][0])
    out["seg_logits"] = f(cap["dec"][0])
    for t in range(2):
        out["x11_%d" % t] = f(cap["x11"][t])
        out["l1_%d" % t] = f(cap["feats"][t][2])
        out["l4_%d" % t] = f(cap["feats"][t][5])
        out["dec_hid_%d" % t], out["dec_out_%d" % t] = f(cap["fba_dec"][t][0]), f(cap["fba_dec"][t][1])
        out["hid_%d" % t], out["ref7_%d" % t], out["tri_logits_%d" % t] = (f(v) for v in cap["fba_ref"][t])
        out["r4_m_%d" % t] = f(cap["enc_m"][t][0])
        out["key_m_%d" % t], out["val_m_%d" % t] = (f(v) for v in cap["kv_m"][t])
    return out


def self_noise(H, W, T, style, skip, max_num, dk, clip_seed):
    """Reference vs reference with a different fp32 summation order (oneDNN off): the floor any fp32
    re-implementation is compared against.  Per-frame alpha max-abs and class flips."""
    torch.backends.mkldnn.enabled = False
    try:
        res = run_sequence(H, W, T, style, skip, max_num, dk, clip_seed)
    finally:
        torch.backends.mkldnn.enabled = True
    return res


def metric_fixtures():
    """SAD / MSE / dtSSD from the reference's own BatchMetric methods (utils/tmp/metric.py:177-189,252-264).
    The module imports skimage (absent here) at the top; only for this import a stub module is registered."""
    import types
    load_reference()
    sk = types.ModuleType("skimage"); skm = types.ModuleType("skimage.measure"); sk.measure = skm
    sys.modules.setdefault("skimage", sk); sys.modules.setdefault("skimage.measure", skm)
    from utils.tmp.metric import BatchMetric                    # reference utils/tmp/metric.py:89
    rng = np.random.Generator(np.random.PCG64(11))
    B, H, W = 5, 23, 31
    pred = torch.from_numpy(rng.integers(0, 256, (B, H, W)).astype(np.float32))
    target = torch.from_numpy(rng.integers(0, 256, (B, H, W)).astype(np.float32))
    mask = torch.from_numpy((rng.uniform(0, 1, (B, H, W)) < 0.4).astype(np.float32))
    bm = BatchMetric.__new__(BatchMetric)
    e, n = bm.dtSSD(pred, target, mask)
    return dict(met_pred=pred.numpy(), met_target=target.numpy(), met_mask=mask.numpy(),
                met_sad=bm.BatchSAD(pred, target, mask), met_mse=bm.BatchMSE(pred, target, mask), met_dt_err=e, met_dt_num=n)


def op_fixtures():
    """Per-function vectors from the reference's own functions."""
    load_reference()
    import math
    from utils.utils import trimap_transform                    # reference utils/utils.py:25-39
    from models.alpha.FBA.models import fba_fusion              # reference FBA/models.py:279-288
    from models.alpha.common import pad_divide_by               # reference alpha/common.py:6-27
    from models.trimap.STM import Memory                        # reference STM.py:140-163
    from models.alpha.FBA import layers_WS as L                 # reference layers_WS.py
    out = {}
    rng = np.random.Generator(np.random.PCG64(7))
    # trimap_transform on crafted masks: empty fg class, single pixel, full, random blobs
    H, W = 37, 53
    masks = np.zeros((4, 2, H, W), np.float32)
    masks[0, 0] = 1.0                                            # all bg, fg empty
    masks[1, 0] = 1.0; masks[1, 0, 20, 30] = 0.0; masks[1, 1, 20, 30] = 1.0   # single fg pixel
    masks[2, 1] = 1.0                                            # all fg, bg empty
    blob = rng.uniform(0, 1, (H, W))
    masks[3, 0] = blob < 0.2
    masks[3, 1] = blob > 0.85
    out["tt_masks"] = masks
    out["tt_out"] = np.stack([trimap_transform(torch.from_numpy(m)[None, None])[0, 0].numpy() for m in masks])
    # fba_fusion
    a = torch.from_numpy(rng.uniform(-0.2, 1.2, (1, 1, 9, 11)).astype(np.float32)).clamp(0, 1)
    img = torch.from_numpy(rng.uniform(0, 1, (1, 3, 9, 11)).astype(np.float32))
    Fg = torch.from_numpy(rng.uniform(0, 1, (1, 3, 9, 11)).astype(np.float32))
    Bg = torch.from_numpy(rng.uniform(0, 1, (1, 3, 9, 11)).astype(np.float32))
    fa, fF, fB = fba_fusion(a, img, Fg, Bg)
    out.update(ff_a=a.numpy(), ff_img=img.numpy(), ff_F=Fg.numpy(), ff_B=Bg.numpy(),
               ff_out=torch.cat([fa, fF, fB], 1).numpy())
    # pad_divide_by
    pads = []
    for (h, w, d) in [(100, 150, 32), (1080, 1920, 32), (33, 65, 16), (64, 64, 32), (70, 90, 32), (1, 31, 32)]:
        _, pad = pad_divide_by([torch.zeros(1, 1, h, w)], d, (h, w))
        pads.append([h, w, d] + list(pad))
    out["pads"] = np.asarray(pads)
    # Memory.forward
    mem = Memory()
    for T in (1, 2, 5):
        h, w = 5, 7
        mk = torch.from_numpy(rng.normal(0, 3, (1, 128, T, h, w)).astype(np.float32))
        mv = torch.from_numpy(rng.normal(0, 1, (1, 512, T, h, w)).astype(np.float32))
        qk = torch.from_numpy(rng.normal(0, 3, (1, 128, h, w)).astype(np.float32))
        qv = torch.from_numpy(rng.normal(0, 1, (1, 512, h, w)).astype(np.float32))
        out["mem%d_mk" % T], out["mem%d_mv" % T] = mk.numpy(), mv.numpy()
        out["mem%d_qk" % T], out["mem%d_qv" % T] = qk.numpy(), qv.numpy()
        out["mem%d_out" % T] = mem(mk, mv, qk, qv).numpy()
    # weight-standardised conv + GroupNorm(32) block
    conv = L.Conv2d(24, 64, 3, padding=2, dilation=2, bias=True)
    gn = L.BatchNorm2d(64)
    w = rng.normal(0.02, 0.05, (64, 24, 3, 3)).astype(np.float32)
    b = rng.normal(0, 0.1, 64).astype(np.float32)
    g = rng.uniform(0.5, 1.5, 64).astype(np.float32)
    be = rng.normal(0, 0.1, 64).astype(np.float32)
    conv.weight.data = torch.from_numpy(w); conv.bias.data = torch.from_numpy(b)
    gn.weight.data = torch.from_numpy(g); gn.bias.data = torch.from_numpy(be)
    x = torch.from_numpy(rng.normal(0, 1, (1, 24, 13, 17)).astype(np.float32))
    out.update(ws_w=w, ws_b=b, ws_g=g, ws_be=be, ws_x=x.numpy(), ws_out=gn(conv(x)).detach().numpy())
    return out


# BASELINE configs[1] geometry (832x480, no padding), produced by the reference itself: pins the oracle -- and through it the
# HIP path -- at a size where the split-K route and the small-map tiles carry most layers.  skip=3 / max=3 so that frames 2
# and 3 read two memory slots (frame 3 after a replace-last).  ~4 s per frame on 8 threads.
FULLSIZE = ("c480_832x480_s3m3", 480, 832, 4, "demo", 3, 3, 12, 7)


def fullsize_fixture():
    name, H, W, T, style, skip, max_num, dk, cs = FULLSIZE
    res = run_sequence(H, W, T, style, skip, max_num, dk, cs)
    alt = self_noise(H, W, T, style, skip, max_num, dk, cs)
    noise = [float(np.abs(alt["alpha"][t] - res["alpha"][t]).max()) for t in range(T)]
    flips = [int((alt["trimap"][t].argmax(0) != res["trimap"][t].argmax(0)).sum()) for t in range(T)]
    # alpha as float32 (the compared quantity); the trimap as its class map + the winning probability in fp16 (the 3-channel
    # fp32 probabilities would be 19 MB)
    tri = res["trimap"]
    np.savez_compressed(os.path.join(HERE, "seq_%s.npz" % name), alpha=res["alpha"], bank=res["bank"],
                        trimap_cls=tri.argmax(1).astype(np.uint8), trimap_top=tri.max(1).astype(np.float16),
                        key_probe=res["key_probe"])
    meta = dict(H=H, W=W, T=T, style=style, skip=skip, max_num=max_num, dilate_kernel=dk, clip_seed=cs, weight_seed=0,
                bank=res["bank"].tolist(), reference_self_noise_alpha_maxabs=noise, reference_self_noise_trimap_flips=flips)
    json.dump({name: meta}, open(os.path.join(HERE, "fullsize.json"), "w"), indent=1)
    print(name, "bank", res["bank"].tolist(), "alpha mean %.4f" % res["alpha"].mean(), "self-noise", noise, flips)


# BASELINE configs[2] geometry (1920x1080 -> padded 1088x1920, the only padded BASELINE size), produced by the reference itself:
# the first frame and one propagated frame (memory read over one slot).  Round 5 (at this size the HIP path was
# compared with the oracle only).  Kept small (6.2 MB): the alpha of frame 1 as float32 (the compared quantity), frame 0's alpha
# as its per-row sums (an anchor for the first frame), the class maps as uint8.  ~1 min per frame on 8 threads.
C1080 = ("c1080_1920x1080_s5m5", 1080, 1920, 2, "demo", 5, 5, 12, 9)


def c1080_fixture():
    name, H, W, T, style, skip, max_num, dk, cs = C1080
    res = run_sequence(H, W, T, style, skip, max_num, dk, cs)
    alt = self_noise(H, W, T, style, skip, max_num, dk, cs)
    noise = [float(np.abs(alt["alpha"][t] - res["alpha"][t]).max()) for t in range(T)]
    flips = [int((alt["trimap"][t].argmax(0) != res["trimap"][t].argmax(0)).sum()) for t in range(T)]
    tri = res["trimap"]
    np.savez_compressed(os.path.join(HERE, "seq_%s.npz" % name), alpha1=res["alpha"][1].astype(np.float32),
                        alpha0_rowsum=res["alpha"][0].astype(np.float64).sum(1).astype(np.float32), bank=res["bank"],
                        trimap_cls=tri.argmax(1).astype(np.uint8), key_probe=res["key_probe"])
    meta = dict(H=H, W=W, T=T, style=style, skip=skip, max_num=max_num, dilate_kernel=dk, clip_seed=cs, weight_seed=0,
                bank=res["bank"].tolist(), reference_self_noise_alpha_maxabs=noise, reference_self_noise_trimap_flips=flips)
    path = os.path.join(HERE, "fullsize.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[name] = meta
    json.dump(doc, open(path, "w"), indent=1)
    print(name, "bank", res["bank"].tolist(), "alpha mean %.4f" % res["alpha"].mean(), "self-noise", noise, flips)


# Round 6: the same geometry through the STEADY read -- five frames, memory every 3, at most 3 slots, so that frames
# 2, 3 and 4 read 2, 2 and 3 slots (alpha/model.py:472-493 over 8160 positions per slot, STM.py:148-159) in the REFERENCE itself.
# Alpha as float32 for the last two frames (the compared quantity), per-row sums for the others, class maps as uint8.
C1080B = ("c1080_1920x1080_s3m3", 1080, 1920, 5, "demo", 3, 3, 12, 10)


def c1080_steady_fixture():
    name, H, W, T, style, skip, max_num, dk, cs = C1080B
    res = run_sequence(H, W, T, style, skip, max_num, dk, cs)
    alt = self_noise(H, W, T, style, skip, max_num, dk, cs)
    noise = [float(np.abs(alt["alpha"][t] - res["alpha"][t]).max()) for t in range(T)]
    flips = [int((alt["trimap"][t].argmax(0) != res["trimap"][t].argmax(0)).sum()) for t in range(T)]
    tri = res["trimap"]
    keep = [T - 2, T - 1]
    np.savez_compressed(os.path.join(HERE, "seq_%s.npz" % name), alpha_frames=np.asarray(keep),
                        alpha=np.stack([res["alpha"][t] for t in keep]).astype(np.float32),
                        alpha_rowsum=res["alpha"].astype(np.float64).sum(2).astype(np.float32), bank=res["bank"],
                        trimap_cls=tri.argmax(1).astype(np.uint8), key_probe=res["key_probe"])
    meta = dict(H=H, W=W, T=T, style=style, skip=skip, max_num=max_num, dilate_kernel=dk, clip_seed=cs, weight_seed=0,
                bank=res["bank"].tolist(), alpha_frames=keep, reference_self_noise_alpha_maxabs=noise,
                reference_self_noise_trimap_flips=flips)
    path = os.path.join(HERE, "fullsize.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[name] = meta
    json.dump(doc, open(path, "w"), indent=1)
    print(name, "bank", res["bank"].tolist(), "alpha mean %.4f" % res["alpha"].mean(), "self-noise", noise, flips)


def main():
    torch.set_num_threads(8)
    if "--c480" in sys.argv:
        return fullsize_fixture()
    if "--c1080" in sys.argv:
        return c1080_fixture()
    if "--c1080-steady" in sys.argv:
        return c1080_steady_fixture()
    meta = {}
    for (name, H, W, T, style, skip, max_num, dk, cs) in SEQUENCES:
        res = run_sequence(H, W, T, style, skip, max_num, dk, cs)
        np.savez_compressed(os.path.join(HERE, "seq_%s.npz" % name), **res)
        alt = self_noise(H, W, T, style, skip, max_num, dk, cs)
        noise = [float(np.abs(alt["alpha"][t] - res["alpha"][t]).max()) for t in range(T)]
        flips = [int((alt["trimap"][t].argmax(0) != res["trimap"][t].argmax(0)).sum()) for t in range(T)]
        meta[name] = dict(H=H, W=W, T=T, style=style, skip=skip, max_num=max_num, dilate_kernel=dk,
                          clip_seed=cs, weight_seed=0, bank=res["bank"].tolist(),
                          reference_self_noise_alpha_maxabs=noise, reference_self_noise_trimap_flips=flips)
        print(name, "bank", res["bank"].tolist(), "alpha mean %.4f" % res["alpha"].mean())
    ops = op_fixtures()
    ops.update(metric_fixtures())
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops)
    np.savez_compressed(os.path.join(HERE, "stages_64x64.npz"), **stage_fixture())
    json.dump(meta, open(os.path.join(HERE, "sequences.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
