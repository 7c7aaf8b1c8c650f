"""Training-mode forward of the reference's joint model, FORWARD ONLY.

Mirror of ``FullModel.forward`` (reference models/alpha/model.py:189-312) for stage 4: a batch of B clips of S frames goes
through the network frame by frame -- the B clips in lock-step, one launch per layer over the B images
(``HipEngine.frame_batch``) -- with the trimap propagated by the STM (``single_step``: every frame is memorised, nothing
is evicted, models/trimap/model.py:138-158), then the FBA losses (model.py:100-187, utils/loss_func.py) of the decoder
and refinement heads and the trimap cross-entropies are evaluated by the kernels of ``csrc/losses.hip``.  BatchNorm is in
eval mode as train.py:311-319 arranges.  No backward: there are no gradient kernels in this library.

The frame step is the inference step: per frame t the reference runs  alpha network(t) -> memorize(t) -> segment(t+1),
which is the order the engine executes (the memorize of frame t opens the call of frame t+1).  Differences from inference:
frame 0 memorises the GROUND-TRUTH trimap (model.py:212), the heads' F / B outputs and the raw logits are kept, sizes must be
multiples of 32 (the reference does not pad here).
"""
import ctypes as C

import torch

from . import lib as L

EPS = 1.001e-5            # utils/loss_func.py: epsilon of L1_grad / exclusion_loss


def _fba_loss(lib, st, dev, pred7, gts, trimask, fgs, bgs, imgs, B, S, H, W):
    """fba_single_image_loss (model.py:100-187) with start = 0, end = S, normalize = True, on the device.
    Returns (L_alpha_comp, L_lap, L_grad, alphas, comps, Fs, Bs) -- the three losses as Python floats."""
    f32, P, N = torch.float32, H * W, B * S
    cF = torch.empty((B, S, 3, H, W), dtype=f32, device=dev)
    cB, comp = torch.empty_like(cF), torch.empty_like(cF)
    alphas = torch.empty((B, S, 1, H, W), dtype=f32, device=dev)
    n_lev = 3
    acc = torch.zeros(16 + n_lev * (S * 4 + N * 2), dtype=torch.float64, device=dev)
    ap = acc.data_ptr()
    L.check(lib.otvm_loss_fba_comp(pred7.data_ptr(), gts.data_ptr(), trimask.data_ptr(), fgs.data_ptr(), bgs.data_ptr(), imgs.data_ptr(),
                                   N, P, cF.data_ptr(), cB.data_ptr(), comp.data_ptr(), alphas.data_ptr(), ap, st), "loss_fba_comp")
    L.check(lib.otvm_loss_grad_l1(alphas.data_ptr(), gts.data_ptr(), N, H, W, EPS, ap + 8 * 5, st), "loss_grad_l1")
    # exclusion loss, three levels (loss_func.py:56-82)
    i1, i2, h, w = cF, cB, H, W
    keep = []
    for lv in range(n_lev):
        a1 = ap + 8 * (16 + lv * (S * 4 + N * 2))
        L.check(lib.otvm_loss_exclusion_level(i1.data_ptr(), i2.data_ptr(), B, S, h, w, EPS, a1, a1 + 8 * S * 4, st), "loss_exclusion")
        if lv + 1 < n_lev:
            n1 = torch.empty((N * 3, h // 2, w // 2), dtype=f32, device=dev)
            n2 = torch.empty_like(n1)
            L.check(lib.otvm_avgpool2(i1.data_ptr(), N * 3, h, w, n1.data_ptr(), st), "avgpool2")
            L.check(lib.otvm_avgpool2(i2.data_ptr(), N * 3, h, w, n2.data_ptr(), st), "avgpool2")
            keep += [i1, i2]
            i1, i2, h, w = n1, n2, h // 2, w // 2
    # Laplacian pyramid loss, five levels (loss_func.py:95-155): (alpha, gt), (F, fgs), (B, bgs)
    for k, (x, y, n) in enumerate(((alphas, gts, N), (cF, fgs, N * 3), (cB, bgs, N * 3))):
        ci, ct, h, w = x, y, H, W
        for lv in range(5):
            di = torch.empty((n, h // 2, w // 2), dtype=f32, device=dev)
            dt = torch.empty_like(di)
            L.check(lib.otvm_loss_lap_level(ci.data_ptr(), ct.data_ptr(), n, h, w, float(2 ** lv), di.data_ptr(), dt.data_ptr(),
                                            ap + 8 * (6 + k), st), "loss_lap_level")
            keep += [ci, ct]
            ci, ct, h, w = di, dt, h // 2, w // 2
    if S > 1:
        for k, (x, y, cp) in enumerate(((alphas, gts, P), (cF, fgs, 3 * P), (cB, bgs, 3 * P))):
            L.check(lib.otvm_loss_temporal(x.data_ptr(), y.data_ptr(), B, S, cp, ap + 8 * (9 + k), st), "loss_temporal")
    v = acc.cpu().tolist()                                     # one read-back of the partial sums (the call returns scalars anyway)
    c1, c3 = float(N * P), float(N * 3 * P)
    L_a1, L_ac, L_FBc, L_FB1 = v[0] / c1, v[1] / c3, v[2] / c3, v[3] / c3 + v[4] / c3
    L_alpha_comp = L_a1 + L_ac + 0.25 * (L_FBc + L_FB1)
    excl, h, w = 0.0, H, W
    for lv in range(n_lev):
        o = 16 + lv * (S * 4 + N * 2) + S * 4
        cnt = 3.0 * h * w
        excl += sum((v[o + 2 * i] / cnt + EPS) ** 0.25 + (v[o + 2 * i + 1] / cnt + EPS) ** 0.25 for i in range(N))
        h, w = h // 2, w // 2
    excl /= float(B * n_lev * S)                               # mean over the batch, / level, mean over the frames
    L_grad = v[5] / c1 + 0.25 * excl
    L_lap = v[6] / c1 + 0.25 * (v[7] / c3 + v[8] / c3)
    if S > 1:
        t1, t3 = float(B * (S - 1) * P), float(B * (S - 1) * 3 * P)
        L_grad += v[9] / t1 + 0.25 * (v[10] / t3 + v[11] / t3)
    return L_alpha_comp, L_lap, L_grad, alphas, comp, cF, cB


@torch.no_grad()
def train_forward(model, a, fg, bg, tri):
    """a [B,S,1,H,W] in [0,1]; fg, bg [B,S,3,H,W] BGR 0..255; tri [B,S,3,H,W] one-hot GT trimaps ([bg, unknown, fg]).
    Returns the reference's list: [loss1, loss2, loss3, loss_trimap, scaled_imgs, tris_vis, alphas, comps, scaled_gts, Fs, Bs,
    preds_trimap] (losses as 0-dim device tensors)."""
    eng = model._get_engine()
    eng.keep_hid_d = True            # (the fused decoder head otherwise never writes the hidden state the training head reads)
    lib, dev, f32 = eng.lib, eng.dev, torch.float32
    if tri is None:
        raise NotImplementedError("otvm_amd training forward: per-frame ground-truth trimaps (tri) are required, as train.py passes them")
    a, fg, bg, tri = (x.to(dev, f32).contiguous() for x in (a, fg, bg, tri))
    B, S, _, H, W = a.shape
    if H % 32 or W % 32 or H < 64 or W < 64:
        raise ValueError("otvm_amd training forward: H and W must be multiples of 32 and >= 64 (the reference pads nothing here; "
                         "the Laplacian loss takes five pyramid levels), got %dx%d" % (W, H))
    P = H * W
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        dec7 = torch.empty((B, S, 7, H, W), dtype=f32, device=dev)
        ref7 = torch.empty_like(dec7)
        ref_logits = torch.empty((B, S, 3, H, W), dtype=f32, device=dev)
        seg_logits = torch.empty((B, max(S - 1, 1), 3, H, W), dtype=f32, device=dev)
        preds_trimap = torch.empty((B, S, 3, H, W), dtype=f32, device=dev)
        scaled_imgs = torch.empty((B, S, 3, H, W), dtype=f32, device=dev)
        for t in range(S):
            cap = dict(dec7=[dec7[b, t] for b in range(B)], ref7=[ref7[b, t] for b in range(B)],
                       ref_logits=[ref_logits[b, t] for b in range(B)],
                       seg_logits=[seg_logits[b, t - 1] for b in range(B)] if t > 0 else None,
                       tri0=[tri[b, 0] for b in range(B)] if t == 0 else None)
            outs = eng._frame_batch([a[b:b + 1, t:t + 1] for b in range(B)], [fg[b:b + 1, t:t + 1] for b in range(B)],
                                    [bg[b:b + 1, t:t + 1] for b in range(B)], [tri[b:b + 1, 0:1] for b in range(B)],
                                    first_frame=(t == 0), last_frame=(t == S - 1), memorize=True, max_memory_num=S + 1,
                                    inputs_ready=None, train=cap)
            for b in range(B):
                scaled_imgs[b, t].copy_(outs[b][0][0, 0])
                preds_trimap[b, t].copy_(tri[b, 0] if t == 0 else outs[b][1][0, 0])      # model.py:212,231
        # ---- losses (model.py:255-290)
        fgs, bgs = torch.empty_like(fg), torch.empty_like(bg)
        L.check(lib.otvm_scale_flip3(fg.data_ptr(), B * S, P, 1.0 / 255, fgs.data_ptr(), st), "scale_flip3")
        L.check(lib.otvm_scale_flip3(bg.data_ptr(), B * S, P, 1.0 / 255, bgs.data_ptr(), st), "scale_flip3")
        trimask = torch.empty((B, S, 1, H, W), dtype=f32, device=dev)
        tris_vis = torch.empty_like(trimask)
        cls = torch.empty((B, S, H, W), dtype=torch.uint8, device=dev)
        L.check(lib.otvm_trimask(tri.data_ptr(), B * S, P, trimask.data_ptr(), cls.data_ptr(), a.data_ptr(), tris_vis.data_ptr(), st),
                "trimask")
        L1 = _fba_loss(lib, st, dev, dec7, a, trimask, fgs, bgs, scaled_imgs, B, S, H, W)
        L2 = _fba_loss(lib, st, dev, ref7, a, trimask, fgs, bgs, scaled_imgs, B, S, H, W)
        ce = torch.zeros(2, dtype=torch.float64, device=dev)
        L.check(lib.otvm_loss_ce3(ref_logits.data_ptr(), cls.data_ptr(), B * S, P, ce.data_ptr() + 8, st), "loss_ce3")
        if S > 1:
            for b in range(B):
                L.check(lib.otvm_loss_ce3(seg_logits[b].data_ptr(), cls[b, 1:].data_ptr(), S - 1, P, ce.data_ptr(), st), "loss_ce3")
        cev = ce.cpu().tolist()
        loss_trimap = cev[1] / float(B * S * P) + (cev[0] / float(B * (S - 1) * P) if S > 1 else 0.0)
        mk = lambda v: torch.tensor(v, dtype=f32, device=dev)
        return [mk(L1[0] + L2[0]), mk(L1[1] + L2[1]), mk(L1[2] + L2[2]), mk(loss_trimap), scaled_imgs, tris_vis, L2[3], L2[4], a, L2[5],
                L2[6], preds_trimap]
