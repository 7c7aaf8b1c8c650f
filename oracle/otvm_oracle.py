"""CPU oracle for the OTVM per-frame inference path.  TEST INFRASTRUCTURE -- never on the product path.

A restatement, in plain functional PyTorch-CPU fp32, of what the reference executes per frame in
``EvalModel.forward`` (reference models/alpha/model.py:391-512) and everything it calls.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file;
``otvm_amd`` (the product) must not and fails loudly without its HIP library.

Pinning: the reference ships no tests / golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, imported in the development container
(``tools/ref_import.py`` + ``tests/golden/make_golden.py``) and committed as fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks them.  Known deviations of that pinning:
OpenCV's ``distanceTransform(DIST_L2, DIST_MASK_PRECISE)`` is replaced by an exact Euclidean
transform (scipy) on both sides, and trained weights are unavailable (synthetic checkpoint).

Differences from the reference that do NOT change results (documented in DESIGN.md):
  * weight standardisation (layers_WS.py:15-21) is evaluated once at construction, with the
    same operations, instead of on every forward;
  * the dead distance transforms of ``make_trimap_gt`` / the first ``make_trimap`` call for t>0
    (alpha/model.py:387,399,416) are not evaluated -- their results are discarded by the reference;
  * the bank is a python list of per-slot tensors instead of ``torch.cat`` along dim 3.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SIGMAS_L = (0.02 * 320, 0.08 * 320, 0.16 * 320)       # utils/utils.py:34-37 (L = 320)


# --------------------------------------------------------------------------- small helpers
def pad_amounts(h, w, d):
    """(lw, uw, lh, uh) of models/alpha/common.py:6-27 -- symmetric, extra pixel bottom/right."""
    nh = h + (d - h % d) % d
    nw = w + (d - w % d) % d
    lh = int((nh - h) / 2)
    lw = int((nw - w) / 2)
    return lw, nw - w - lw, lh, nh - h - lh


def exact_edt(mask_nonzero):
    """Exact Euclidean distance of every non-zero pixel to the nearest zero pixel, float32.

    Stand-in for cv2.distanceTransform(src, DIST_L2, DIST_MASK_PRECISE) (utils/utils.py:21)."""
    from scipy import ndimage
    return ndimage.distance_transform_edt(mask_nonzero).astype(np.float32)


def exact_edt_sq_bruteforce(mask_nonzero):
    """O(N^2) integer squared distance; only for tiny test images (cross-check of exact_edt)."""
    H, W = mask_nonzero.shape
    zy, zx = np.nonzero(~mask_nonzero.astype(bool))
    yy, xx = np.mgrid[0:H, 0:W]
    d2 = (yy[..., None] - zy) ** 2 + (xx[..., None] - zx) ** 2
    return d2.min(-1)


def trimap_transform(trimap2):
    """utils/utils.py:25-39 on a [2, H, W] hard {bg, fg} mask pair -> [6, H, W]."""
    out = torch.zeros((6,) + tuple(trimap2.shape[1:]), dtype=torch.float32)
    for k in range(2):
        tk = trimap2[k]
        if torch.sum(tk != 0) > 0:
            src = ((1.0 - tk).numpy() * 255).astype(np.uint8)       # utils/utils.py:21
            d = torch.from_numpy(exact_edt(src != 0))
            dt_mask = -d ** 2
            for j, s in enumerate(SIGMAS_L):
                out[3 * k + j] = torch.exp(dt_mask / (2 * (s ** 2)))
    return out


def class_map(tri3):
    """argmax over [bg, un, fg] (alpha/model.py:42), ties -> lowest index."""
    return tri3.max(dim=0)[1]


def make_trimap8(tri3, cls=None):
    """alpha/model.py:40-53 for one [3, H, W] soft/one-hot trimap -> [8, H, W]."""
    if cls is None:
        cls = class_map(tri3)
    scaled = cls.float() * 0.5
    t2 = torch.stack([(scaled == 0).float(), (scaled == 1).float()])
    enc6 = trimap_transform(t2)
    return torch.cat([enc6.to(tri3.dtype), tri3[0:1], tri3[2:3]], dim=0)


def fba_fusion(alpha, img, Fg, Bg):
    """FBA/models.py:279-288.  NB: the B update at :281 reads the F already overwritten at :280."""
    Fn = alpha * img + (1 - alpha ** 2) * Fg - alpha * (1 - alpha) * Bg
    Bn = (1 - alpha) * img + (2 * alpha - alpha ** 2) * Bg - alpha * (1 - alpha) * Fn
    Fn = torch.clamp(Fn, 0, 1)
    Bn = torch.clamp(Bn, 0, 1)
    la = 0.1
    alpha = (alpha * la + torch.sum((img - Bn) * (Fn - Bn), 1, keepdim=True)) / (
        torch.sum((Fn - Bn) * (Fn - Bn), 1, keepdim=True) + la)
    return torch.clamp(alpha, 0, 1), Fn, Bn


def standardise_weight(w):
    """layers_WS.py:15-21 (unbiased variance, +1e-12 inside sqrt, +1e-5 outside)."""
    wm = w.mean(dim=1, keepdim=True).mean(dim=2, keepdim=True).mean(dim=3, keepdim=True)
    w = w - wm
    std = torch.sqrt(torch.var(w.view(w.size(0), -1), dim=1) + 1e-12).view(-1, 1, 1, 1) + 1e-5
    return w / std.expand_as(w)


def up(x, scale=None, size=None):
    return F.interpolate(x, scale_factor=scale, size=size, mode="bilinear", align_corners=False)


def memory_read(keys, vals, q_key, q_val, dtype=None, max_bytes=1 << 30):
    """STM.py:144-163.  keys [128, T, h, w], vals [512, T, h, w], q_key [128, h, w], q_val [512, h, w].

    The reference materialises p = softmax(K^T Q / sqrt(128)) over the memory axis as one [T*h*w, h*w] matrix.  Every
    query column of that matrix is independent (the softmax runs over dim 1, the memory axis), so when the matrix would
    exceed ``max_bytes`` the same expression is evaluated for blocks of query columns -- identical arithmetic per column,
    bounded memory (a 4K frame with three slots is 12.8 GB in fp32, several live copies; 200 slots do not fit anywhere).
    dtype (e.g. torch.float64): evaluate this stage in that precision and return the inputs' dtype -- the tests use it at
    sizes where the fp32 bmm + softmax over ~1e5 memory positions is itself ~1e-3 away from the exact value."""
    De, Do = keys.shape[0], vals.shape[0]
    out_dtype = vals.dtype
    cd = out_dtype if dtype is None else dtype
    mi = torch.transpose(keys.reshape(1, De, -1), 1, 2).to(cd)   # [1, THW, De]
    qi = q_key.reshape(1, De, -1).to(cd)                          # [1, De, HW]
    mo = vals.reshape(1, Do, -1).to(cd)                           # [1, Do, THW]
    thw, hw = mi.shape[1], qi.shape[2]
    esz = torch.empty((), dtype=cd).element_size()
    if thw * hw * esz <= max_bytes:
        p = torch.bmm(mi, qi) / math.sqrt(De)
        p = F.softmax(p, dim=1)                                   # over the memory axis
        mem = torch.bmm(mo, p)
    else:
        step = max(64, int(max_bytes // (thw * esz)) // 64 * 64)
        mem = torch.empty((1, Do, hw), dtype=cd)
        for c0 in range(0, hw, step):
            p = torch.bmm(mi, qi[:, :, c0:c0 + step]) / math.sqrt(De)
            p = F.softmax(p, dim=1)
            mem[:, :, c0:c0 + step] = torch.bmm(mo, p)
    mem = mem.reshape(Do, *q_val.shape[1:]).to(out_dtype)
    return torch.cat([mem, q_val], dim=0)


def bank_update(bank, new, first_frame, memorize, max_memory_num):
    """alpha/model.py:472-493 on a python list of slots (each slot = (key, val, frame_id))."""
    if max_memory_num == 0:
        return [new] if first_frame else bank
    if max_memory_num == 1:
        return [new]
    if first_frame:
        bank = [new]
    elif memorize:
        bank = bank + [new]
    elif len(bank) == 1:
        bank = bank + [new]
    else:
        bank = bank[:-1] + [new]
    if len(bank) > max_memory_num:
        bank = bank[:1] + bank[2:]
    return bank


# --------------------------------------------------------------------------- the model
class OtvmOracle:
    def __init__(self, state_dict, dilate_kernel=None, threads=None, dtype=torch.float32, read_dtype=None):
        """dtype=torch.float64 evaluates the SAME operations in double precision (the distance encoding keeps the
        reference's float32 arithmetic, it is part of the definition): the tests use it to measure how far the
        reference's own fp32 forward is from the exact value of its algorithm on a given frame -- the summation-order
        noise any two fp32 implementations differ by."""
        # Host threads.  PyTorch's default (one per physical core) is the slowest setting on a many-core GPU host: measured on the
        # MI355X box (256 logical CPUs, default 128 threads; profiles/r05_oracle_threads.txt) a 1080p first frame takes 19.2 s on
        # 128 threads, 11.2 s on 64, 8.3 s on 32, 7.5 s on 16 -- oneDNN's convolutions of this size do not scale past a few dozen
        # cores and pay for the synchronisation.  Default: at most 32.
        if threads is None and torch.get_num_threads() > 32:
            threads = 32
        if threads:
            torch.set_num_threads(threads)
        self.dtype = dtype
        # read_dtype=torch.float64: only Memory.forward (the softmax over T*h*w memory positions, the one stage whose fp32
        # CPU evaluation drifts ~1e-3 from the exact value at 1080p / 4K bank sizes) is evaluated in double precision; every
        # other stage stays in ``dtype``.  A full float64 frame costs 4x the CPU time, this costs a few seconds.
        self.read_dtype = read_dtype
        self.p = {k: v.detach().to(dtype) if v.is_floating_point() else v.detach() for k, v in state_dict.items()}
        self.dilate_kernel = dilate_kernel
        self.ws = {}
        for k, v in self.p.items():
            if k.endswith(".weight") and v.dim() == 4 and self._is_ws(k):
                self.ws[k] = standardise_weight(v)
        self.bank = []
        self.mean = self.p["IMG_MEAN"].reshape(1, 3, 1, 1)
        self.std = self.p["IMG_STD"].reshape(1, 3, 1, 1)

    @staticmethod
    def _is_ws(k):
        if not k.startswith("NET."):
            return False
        if "conv_up4" in k or ".pred." in k:
            return False
        return True

    # ---- primitive layers
    def conv(self, x, name, stride=1, padding=0, dilation=1):
        w = self.ws.get(name + ".weight", self.p[name + ".weight"])
        return F.conv2d(x, w, self.p.get(name + ".bias"), stride, padding, dilation)

    def gn(self, x, name):
        return F.group_norm(x, 32, self.p[name + ".weight"], self.p[name + ".bias"], 1e-5)

    def bn(self, x, name):
        p = self.p
        return F.batch_norm(x, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"],
                            p[name + ".bias"], False, 0.0, 1e-5)

    # ---- FBA encoder (FBA/models.py:208-269, resnet_GN_WS.py:51-86)
    def _gn_bottleneck(self, x, p, stride, dil, has_ds, ds_stride):
        o = F.relu(self.gn(self.conv(x, p + ".conv1"), p + ".bn1"))
        o = F.relu(self.gn(self.conv(o, p + ".conv2", stride, dil, dil), p + ".bn2"))
        o = self.gn(self.conv(o, p + ".conv3"), p + ".bn3")
        idt = x
        if has_ds:
            idt = self.gn(self.conv(x, p + ".downsample.0", ds_stride), p + ".downsample.1")
        return F.relu(o + idt)

    def fba_encoder(self, x):
        e = "NET.encoder."
        c1 = F.relu(self.gn(self.conv(x, e + "conv1", 2, 3), e + "bn1"))
        y = F.max_pool2d(c1, 3, 2, 1)
        # (stride of first block's 3x3, dilation first block, dilation other blocks)
        cfg = {"layer1": (1, 1, 1), "layer2": (2, 1, 1), "layer3": (1, 1, 2), "layer4": (1, 2, 4)}
        feats = [x, c1]
        for lname, n in zip(("layer1", "layer2", "layer3", "layer4"), (3, 4, 6, 3)):
            s0, d0, dn = cfg[lname]
            for b in range(n):
                y = self._gn_bottleneck(y, e + "%s.%d" % (lname, b), s0 if b == 0 else 1,
                                        d0 if b == 0 else dn, b == 0, s0)
            feats.append(y)
        return feats                                   # [x, c1, l1, l2, l3, l4]

    # ---- FBA decoder (FBA/models.py:351-392)
    def fba_decoder(self, feats, img, tri2):
        d = "NET.decoder."
        conv5 = feats[-1]
        parts = [conv5]
        for i, s in enumerate((1, 2, 3, 6)):
            y = F.adaptive_avg_pool2d(conv5, s)
            y = F.leaky_relu(self.gn(self.conv(y, d + "ppm.%d.1" % i), d + "ppm.%d.2" % i), 0.01)
            parts.append(up(y, size=conv5.shape[2:]))
        x = torch.cat(parts, 1)
        x = F.leaky_relu(self.gn(self.conv(x, d + "conv_up1.0", 1, 1), d + "conv_up1.1"), 0.01)
        x = F.leaky_relu(self.gn(self.conv(x, d + "conv_up1.3", 1, 1), d + "conv_up1.4"), 0.01)
        x = torch.cat((up(x, 2), feats[-4]), 1)
        x = F.leaky_relu(self.gn(self.conv(x, d + "conv_up2.0", 1, 1), d + "conv_up2.1"), 0.01)
        x = torch.cat((up(x, 2), feats[-5]), 1)
        x = F.leaky_relu(self.gn(self.conv(x, d + "conv_up3.0", 1, 1), d + "conv_up3.1"), 0.01)
        x = torch.cat((up(x, 2), feats[-6][:, :3], img), 1)            # 70 channels
        x2 = torch.cat((x, tri2), 1)
        h = F.leaky_relu(self.conv(x2, d + "conv_up4.0", 1, 1), 0.01)
        hid = F.leaky_relu(self.conv(h, d + "conv_up4.2", 1, 1), 0.01)
        out = self.conv(hid, d + "conv_up4.4")
        alpha = torch.clamp(out[:, 0:1], 0, 1)
        alpha, Fg, Bg = fba_fusion(alpha, img, torch.sigmoid(out[:, 1:4]), torch.sigmoid(out[:, 4:7]))
        return hid, torch.cat((alpha, Fg, Bg), 1), x

    # ---- refinement (FBA/models.py:417-435, resnet_GN_WS.py:19-48)
    def fba_refine(self, x_dec, img, tri2, pred_alpha):
        r = "NET.refine."
        x = torch.cat((x_dec, tri2, pred_alpha), 1)
        x = F.leaky_relu(self.gn(self.conv(x, r + "conv1.0", 1, 1), r + "conv1.1"), 0.01)
        for l in ("layer1", "layer2"):
            o = F.relu(self.gn(self.conv(x, r + l + ".conv1", 1, 1), r + l + ".bn1"))
            o = self.gn(self.conv(o, r + l + ".conv2", 1, 1), r + l + ".bn2")
            x = F.relu(o + x)
        h = F.leaky_relu(self.conv(x, r + "pred.0", 1, 1), 0.01)
        hid = F.leaky_relu(self.conv(h, r + "pred.2", 1, 1), 0.01)
        out = self.conv(hid, r + "pred.4")
        alpha = torch.clamp(out[:, 0:1], 0, 1)
        alpha, Fg, Bg = fba_fusion(alpha, img, torch.sigmoid(out[:, 1:4]), torch.sigmoid(out[:, 4:7]))
        return hid, torch.cat((alpha, Fg, Bg), 1), out[:, 7:10]

    def fba(self, inputs11, img, tri2, capture=None):
        """MattingModule.forward (FBA/models.py:32-45)."""
        feats = self.fba_encoder(inputs11)
        hid_d, out7, x_dec = self.fba_decoder(feats, img, tri2)
        hid, ref7, tri_logits = self.fba_refine(x_dec, img, tri2, out7[:, :1])
        if capture is not None:
            capture.update(feats=feats, dec_hid=hid_d, dec_out=out7, x_dec=x_dec)
        return out7, hid, ref7, tri_logits

    # ---- STM (STM.py)
    def _bn_bottleneck(self, x, p, stride, has_ds):
        o = F.relu(self.bn(self.conv(x, p + ".conv1"), p + ".bn1"))
        o = F.relu(self.bn(self.conv(o, p + ".conv2", stride, 1), p + ".bn2"))
        o = self.bn(self.conv(o, p + ".conv3"), p + ".bn3")
        idt = x
        if has_ds:
            idt = self.bn(self.conv(x, p + ".downsample.0", stride), p + ".downsample.1")
        return F.relu(o + idt)

    def _stm_trunk(self, x, e):
        c1 = F.relu(self.bn(x, e + "bn1"))
        y = F.max_pool2d(c1, 3, 2, 1)
        outs = []
        for lname, n, s0 in (("res2", 3, 1), ("res3", 4, 2), ("res4", 6, 2)):
            for b in range(n):
                y = self._bn_bottleneck(y, e + "%s.%d" % (lname, b), s0 if b == 0 else 1, b == 0)
            outs.append(y)
        return outs[2], outs[1], outs[0]               # r4, r3, r2

    def encoder_q(self, img01):
        e = "trimap.model.Encoder_Q."
        f = (img01 - self.p[e + "mean"]) / self.p[e + "std"]
        return self._stm_trunk(self.conv(f, e + "conv1", 2, 3), e)

    def encoder_m(self, img01, p_un, p_fg, alpha, hid):
        e = "trimap.model.Encoder_M."
        f = (img01 - self.p[e + "mean"]) / self.p[e + "std"]
        x = self.conv(p_un, e + "conv1_m", 2, 3) + self.conv(p_fg, e + "conv1_o", 2, 3) \
            + self.conv(alpha, e + "conv1_a", 2, 3) + self.conv(hid, e + "conv1_h", 2, 3)
        x = self.conv(f, e + "conv1", 2, 3) + x
        return self._stm_trunk(x, e)

    def _resblock(self, x, p):
        r = self.conv(F.relu(x), p + ".conv1", 1, 1)
        r = self.conv(F.relu(r), p + ".conv2", 1, 1)
        return x + r

    def stm_decoder(self, m4in, r3, r2):
        d = "trimap.model.Decoder."
        m4 = self._resblock(self.conv(m4in, d + "convFM", 1, 1), d + "ResMM")
        s = self._resblock(self.conv(r3, d + "RF3.convFS", 1, 1), d + "RF3.ResFS")
        m3 = self._resblock(s + up(m4, 2), d + "RF3.ResMM")
        s = self._resblock(self.conv(r2, d + "RF2.convFS", 1, 1), d + "RF2.ResFS")
        m2 = self._resblock(s + up(m3, 2), d + "RF2.ResMM")
        p2 = self.conv(F.relu(m2), d + "pred", 1, 1)
        return up(p2, 4)

    def stm_segment(self, img01, bank, capture=None):
        """STM.segment (STM.py:239-257) with the bank as a list of (key[128,h,w], val[512,h,w])."""
        r4, r3, r2 = self.encoder_q(img01)
        k4 = self.conv(r4, "trimap.model.KV_Q_r4.Key", 1, 1)
        v4 = self.conv(r4, "trimap.model.KV_Q_r4.Value", 1, 1)
        keys = torch.stack([b[0] for b in bank], dim=1)
        vals = torch.stack([b[1] for b in bank], dim=1)
        m4 = memory_read(keys, vals, k4[0], v4[0], dtype=self.read_dtype)[None]
        logits = self.stm_decoder(m4, r3, r2)
        if capture is not None:
            capture.update(r4=r4, r3=r3, r2=r2, k4=k4, v4=v4, m4=m4, seg_logits=logits)
        return logits

    def stm_memorize(self, img01, tri3, alpha, hid):
        """STM.memorize (STM.py:201-228) through trimap/model.py:227-239."""
        es = torch.cat([tri3, alpha, hid], dim=1)                       # trimap/model.py:231
        r4, _, _ = self.encoder_m(img01, es[:, 1].unsqueeze(1), es[:, 2].unsqueeze(1),
                                  es[:, 3].unsqueeze(1), es[:, 4:])
        k = self.conv(r4, "trimap.model.KV_M_r4.Key", 1, 1)
        v = self.conv(r4, "trimap.model.KV_M_r4.Value", 1, 1)
        return k[0], v[0]

    # ---- first-frame trimap from GT alpha (alpha/model.py:342-362), V108-style flow
    def trimap_from_alpha(self, a):
        trimask = ((a > 0) & (a < 1.0)).to(a.dtype)
        r = self.dilate_kernel
        tm = F.max_pool2d(trimask, kernel_size=r * 2 + 1, stride=1, padding=r)
        t1 = torch.where(tm > 0.5, torch.ones_like(a), 2 * a).long()
        return F.one_hot(t1[:, 0], num_classes=3).permute(0, 3, 1, 2).to(a.dtype)

    # ---- one frame (alpha/model.py:391-512)
    def reset(self):
        self.bank = []

    def frame(self, a, fg, bg, tri_gt=None, first_frame=False, last_frame=False, memorize=False,
              max_memory_num=2, frame_id=0, class_override=None, capture=None):
        """a [1,1,1,H,W] in [0,1]; fg,bg [1,1,3,H,W] BGR 0..255; tri_gt [1,1,3,H,W] one-hot or None."""
        dt = self.dtype
        a4, fg4, bg4 = a[0].to(dt).contiguous(), fg[0].to(dt).contiguous(), bg[0].to(dt).contiguous()
        s = 1.0 / 255
        img = (fg4.flip([1]) * s) * a4 + (bg4.flip([1]) * s) * (1.0 - a4)        # :384-386
        if tri_gt is not None:
            tri = tri_gt[0].to(dt)
            tri_gt_out = F.one_hot(tri.max(dim=1)[1], 3).permute(0, 3, 1, 2).to(dt)   # :356-362
        else:
            tri = self.trimap_from_alpha(a4)
            tri_gt_out = tri
        H, W = img.shape[-2:]
        pad = pad_amounts(H, W, 32)
        imgp = F.pad(img, pad) if sum(pad) else img
        if sum(pad):
            tri = torch.cat((F.pad(tri[:, :1], pad, value=1.0), F.pad(tri[:, 1:], pad, value=0.0)), 1)
        imgn = (imgp - self.mean) / self.std

        if first_frame:
            self.bank = []
            tri_in = tri
        else:
            logits = self.stm_segment(imgp, self.bank, capture)
            tri_in = F.softmax(logits, dim=1)
        cls = class_map(tri_in[0])
        if class_override is not None:
            cls = class_override
        tri8 = make_trimap8(tri_in[0], cls)[None]
        x11 = torch.cat([imgn, tri8], dim=1)
        out7, hid, ref7, tri_logits = self.fba(x11, imgp, tri8[:, -2:], capture)
        alpha = ref7[:, :1]
        tri_out = F.softmax(tri_logits, dim=1)
        new_kv = None
        if not last_frame:
            k, v = self.stm_memorize(imgp, tri_out, alpha, hid)
            new_kv = (k, v)
            self.bank = bank_update(self.bank, (k, v, frame_id), first_frame, memorize, max_memory_num)
        lw, uw, lh, uh = pad
        Hp, Wp = imgp.shape[-2:]
        crop = (slice(None), slice(None), slice(lh, Hp - uh), slice(lw, Wp - uw))
        if capture is not None:
            capture.update(cls=cls, tri8=tri8, x11=x11, imgp=imgp, hid=hid, ref7=ref7, tri_logits=tri_logits,
                           tri_in=tri_in, alpha_p=alpha, tri_out_p=tri_out, new_kv=new_kv, pad=pad)
        return (img[None], tri_out[crop][None], tri_gt_out[None], alpha[crop][None], a)
