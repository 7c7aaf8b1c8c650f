"""CPU oracle of the TRAINING-mode forward.  TEST INFRASTRUCTURE -- never on the product path.

Restates ``FullModel.forward`` of the reference's training class (models/alpha/model.py:189-312, stage 4: refinement on,
trimap network attached) and the loss functions it calls (utils/loss_func.py) on top of ``OtvmOracle``'s network pieces.
BatchNorm runs in eval mode as train.py:311-319 arranges; GroupNorm is per sample, so the network is evaluated sample by
sample and the losses over the stacked batch, as the reference's batched tensors do.  Pinned by tests/golden/train_*.npz
(outputs of the reference itself, tests/golden/make_train_golden.py).
"""
import torch
import torch.nn.functional as F

from .otvm_oracle import OtvmOracle, make_trimap8


# ---------------------------------------------------------------- utils/loss_func.py
def l1_mean(x, y):
    """L1_mask(x, y, mask=None, normalize=True): loss_func.py:4-17."""
    return torch.mean(torch.abs(x - y))


def get_gradient(img):
    """loss_func.py:35-42: forward differences, zero in the last row / column."""
    dy = img[:, :, 1:, :] - img[:, :, :-1, :]
    dx = img[:, :, :, 1:] - img[:, :, :, :-1]
    return F.pad(dx, (0, 1, 0, 0)), F.pad(dy, (0, 0, 0, 1))


def l1_grad(pred, gt, eps=1.001e-5):
    """loss_func.py:44-51."""
    fx, fy = get_gradient(pred)
    tx, ty = get_gradient(gt)
    return l1_mean(torch.sqrt(fx ** 2 + fy ** 2 + eps), torch.sqrt(tx ** 2 + ty ** 2 + eps))


def exclusion_loss(img1, img2, level=3, eps=1.001e-5):
    """loss_func.py:56-82 (normalize=True)."""
    gx_l, gy_l = [], []
    for _ in range(level):
        gx1, gy1 = get_gradient(img1)
        gx2, gy2 = get_gradient(img2)
        ax = 2.0 * torch.mean(torch.abs(gx1)) / (torch.mean(torch.abs(gx2)) + eps)
        ay = 2.0 * torch.mean(torch.abs(gy1)) / (torch.mean(torch.abs(gy2)) + eps)
        gx1s, gy1s = torch.sigmoid(gx1) * 2 - 1, torch.sigmoid(gy1) * 2 - 1
        gx2s, gy2s = torch.sigmoid(gx2 * ax) * 2 - 1, torch.sigmoid(gy2 * ay) * 2 - 1
        gx_l.append((torch.mean((gx1s ** 2) * (gx2s ** 2), dim=(1, 2, 3)) + eps) ** 0.25)
        gy_l.append((torch.mean((gy1s ** 2) * (gy2s ** 2), dim=(1, 2, 3)) + eps) ** 0.25)
        img1, img2 = F.avg_pool2d(img1, 2, 2), F.avg_pool2d(img2, 2, 2)
    return torch.mean(sum(gx_l) / float(level)) + torch.mean(sum(gy_l) / float(level))


GAUSS = torch.tensor([[1., 4., 6., 4., 1.], [4., 16., 24., 16., 4.], [6., 24., 36., 24., 6.], [4., 16., 24., 16., 4.],
                      [1., 4., 6., 4., 1.]]) / 256.


def _conv_gauss(img, kernel):
    img = F.pad(img, (2, 2, 2, 2), mode="reflect")
    return F.conv2d(img, kernel, groups=img.shape[1])


def _lap_up(x, k):
    """LapLoss.upsample (loss_func.py:111-121): zeros interleaved (values at even rows / columns), then 4 x Gaussian."""
    up = torch.zeros(x.shape[0], x.shape[1], x.shape[2] * 2, x.shape[3] * 2, dtype=x.dtype)
    up[:, :, ::2, ::2] = x
    return _conv_gauss(up, 4 * k)


def laplacian_pyramid(img, levels=5):
    k = GAUSS.to(img.dtype).repeat(img.shape[1], 1, 1, 1)
    cur, pyr = img, []
    for _ in range(levels):
        down = _conv_gauss(cur, k)[:, :, ::2, ::2]
        pyr.append(cur - _lap_up(down, k))
        cur = down
    return pyr


def lap_loss(img, tgt):
    """LapLoss.forward (loss_func.py:141-155, normalize=True, mask=None); inputs are padded to a multiple of 32 with zeros."""
    h, w = img.shape[2:]
    nh, nw = (h + 31) // 32 * 32, (w + 31) // 32 * 32
    lh, lw = (nh - h) // 2, (nw - w) // 2
    pad = (lw, nw - w - lw, lh, nh - h - lh)
    img, tgt = F.pad(img, pad), F.pad(tgt, pad)
    loss = sum((2 ** lv) * torch.sum(torch.abs(a - b)) for lv, (a, b) in enumerate(zip(laplacian_pyramid(img), laplacian_pyramid(tgt))))
    return loss / float(tgt.numel())


# ---------------------------------------------------------------- models/alpha/model.py:100-187
def fba_loss(preds, trimasks, gts, fgs, bgs, imgs):
    """fba_single_image_loss with start = 0, end = S, normalize = True.  preds [B,S,7,H,W]."""
    S = preds.shape[1]
    la, ll, lg, alphas, comps, Fs, Bs = [], [], [], [], [], [], []
    for c in range(S):
        gt, tm, img = gts[:, c], trimasks[:, c], imgs[:, c]
        ref = preds[:, c, :1]
        cF = torch.where((tm.bool() & (gt > 0)).repeat(1, 3, 1, 1), preds[:, c, 1:4], fgs[:, c])
        cB = torch.where(tm.bool().repeat(1, 3, 1, 1), preds[:, c, 4:], bgs[:, c])
        alphas.append(ref), comps.append(cF * ref + cB * (1. - ref)), Fs.append(cF), Bs.append(cB)
        L_a1 = l1_mean(ref, gt)
        L_ac = l1_mean(cF * gt + cB * (1. - gt), img)
        L_FBc = l1_mean(fgs[:, c] * ref + bgs[:, c] * (1. - ref), img)
        L_FB1 = l1_mean(cF, fgs[:, c]) + l1_mean(cB, bgs[:, c])
        la.append(L_a1 + L_ac + 0.25 * (L_FBc + L_FB1))
        lg.append(l1_grad(ref, gt) + 0.25 * exclusion_loss(cF, cB, 3))
        ll.append(lap_loss(ref, gt) + 0.25 * (lap_loss(cF, fgs[:, c]) + lap_loss(cB, bgs[:, c])))
    la, lg, ll = sum(la) / float(S), sum(lg) / float(S), sum(ll) / float(S)
    alphas, comps, Fs, Bs = (torch.stack(v, dim=1) for v in (alphas, comps, Fs, Bs))
    if S > 1:
        tc = F.mse_loss(alphas[:, 1:] - alphas[:, :-1], gts[:, 1:] - gts[:, :-1]) + 0.25 * (
            F.mse_loss(Fs[:, 1:] - Fs[:, :-1], fgs[:, 1:] - fgs[:, :-1]) + F.mse_loss(Bs[:, 1:] - Bs[:, :-1], bgs[:, 1:] - bgs[:, :-1]))
        lg = lg + tc
    return la, ll, lg, alphas, comps, Fs, Bs


def train_forward(orc, a, fg, bg, tri):
    """models/alpha/model.py:189-312.  a [B,S,1,H,W], fg / bg [B,S,3,H,W] BGR 0..255, tri [B,S,3,H,W] one-hot.
    Returns dict(loss1, loss2, loss3, loss_trimap, alphas, comps, Fs, Bs, preds_trimap, preds_alpha, preds_alpha_refine,
    logit_trimap, logit_trimap_refine)."""
    dt = orc.dtype
    a, fg, bg, tri = a.to(dt), fg.to(dt), bg.to(dt), tri.to(dt)
    B, S, _, H, W = a.shape
    if H % 32 or W % 32:
        raise ValueError("the training forward takes sizes that are multiples of 32 (the reference does not pad here)")
    s = 1.0 / 255
    fgs, bgs = fg.flip([2]) * s, bg.flip([2]) * s                        # :59-60
    imgs = fgs * a + bgs * (1. - a)                                       # :61
    cls = tri.max(dim=2)[1]                                               # make_trimap :42-43
    trimasks = (cls == 1).unsqueeze(2).to(dt)
    pa, par, ptr, lt, ltr = [], [], [], [], []
    for b in range(B):
        tri_t = tri[b, 0][None]                                           # preds_trimap[0]
        tri_ref = tri[b, 0][None]                                         # preds_trimap_refine[0]
        bank = []
        pa_b, par_b, ptr_b, lt_b, ltr_b = [], [], [tri_ref], [], []
        for t in range(S):
            img = imgs[b, t][None]
            imgn = (img - orc.mean) / orc.std
            tri8 = make_trimap8(tri_t[0])[None]
            out7, hid, ref7, tri_logits = orc.fba(torch.cat([imgn, tri8], 1), img, tri8[:, -2:])
            pa_b.append(out7), par_b.append(ref7), ltr_b.append(tri_logits)
            if t > 0:
                tri_ref = F.softmax(tri_logits, dim=1)
                ptr_b.append(tri_ref)
            if t < S - 1:
                k, v = orc.stm_memorize(img, tri_ref, ref7[:, :1], hid)   # trimap/model.py:138-158 (single_step)
                bank.append((k, v))
                logits = orc.stm_segment(imgs[b, t + 1][None], bank)
                lt_b.append(logits)
                tri_t = F.softmax(logits, dim=1)
        pa.append(torch.cat(pa_b)), par.append(torch.cat(par_b)), ptr.append(torch.cat(ptr_b))
        lt.append(torch.cat(lt_b) if lt_b else torch.zeros(0, 3, H, W, dtype=dt)), ltr.append(torch.cat(ltr_b))
    pa, par, ptr, lt, ltr = (torch.stack(v) for v in (pa, par, ptr, lt, ltr))
    L1 = fba_loss(pa, trimasks, a, fgs, bgs, imgs)
    L2 = fba_loss(par, trimasks, a, fgs, bgs, imgs)
    gt_cls = cls
    loss_tri = F.cross_entropy(ltr.reshape(-1, 3, H, W), gt_cls.reshape(-1, H, W))
    if S > 1:
        loss_tri = loss_tri + F.cross_entropy(lt.reshape(-1, 3, H, W), gt_cls[:, 1:].reshape(-1, H, W))
    return dict(loss1=L1[0] + L2[0], loss2=L1[1] + L2[1], loss3=L1[2] + L2[2], loss_trimap=loss_tri, alphas=L2[3], comps=L2[4],
                Fs=L2[5], Bs=L2[6], preds_trimap=ptr, preds_alpha=pa, preds_alpha_refine=par, logit_trimap=lt,
                logit_trimap_refine=ltr, scaled_imgs=imgs, scaled_fgs=fgs, scaled_bgs=bgs, trimasks=trimasks)
