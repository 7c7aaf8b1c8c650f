"""``run_video_matte``: the build-owned counterpart of the reference's per-sequence loop
(reference eval.py:157-228) -- frame flags, memory schedule, large-input rule, u8 quantisation.

BASELINE.json names a ``run_video_matte()`` call surface; the reference has no such function (its loop
is inlined in eval.py), so this reproduces that loop around the reference-shaped ``EvalModel``.
"""
import numpy as np
import torch


def memory_schedule(i_seq, height, width, skip=10, max_num=5):
    """(memorize, max_memory_num, large_input) for frame ``i_seq`` -- eval.py:180-190, config.py:22-23."""
    large = min(height, width) > 1100
    if large:
        skip, max_num = int(skip * 2), int(max_num / 2)
    memorize = (i_seq % skip) == 0 if skip > 2 else False
    return memorize, max_num, large


def trimap_file_to_onehot(tri):
    """Grayscale / 3-channel trimap image -> one-hot float32 [3,H,W] (bg, unknown, fg): dataset.py:880-893."""
    tri = np.asarray(tri)
    if tri.ndim == 3:
        t = tri[..., :3] > 1                           # BGR order as read by cv2 (an alpha plane carries no class)
        out = np.zeros(t.shape, np.float32)
        out[..., 0][np.logical_not(t[..., 1] + t[..., 2])] = 1
        out[..., 1][t[..., 2]] = 1
        out[..., 2][t[..., 1]] = 1
    else:
        t = tri.copy()
        out = np.zeros(t.shape + (3,), np.float32)
        out[..., 0][t == 0] = 1
        out[..., 2][t == t.max()] = 1
        t[t == t.max()] = 0
        out[..., 1][t == t.max()] = 1                  # second-largest level = unknown (incl. the two-level quirk)
    return np.ascontiguousarray(out.transpose(2, 0, 1))


@torch.no_grad()
def _as_tensor(x):
    """numpy / tensor -> tensor without the read-only-array warning (PIL hands out non-writable buffers)."""
    if torch.is_tensor(x):
        return x
    a = np.ascontiguousarray(x)
    return torch.from_numpy(a if a.flags.writeable else a.copy())


def run_video_matte(model, frames, trimap=None, alphas=None, backgrounds=None, skip=10, max_num=5,
                    frames_are_rgb=False, on_frame=None, device=None, keep_on_device=False, gt_alpha_u8=None,
                    gt_mask_u8=None, gt_mask=None):
    """Matte one sequence.

    model       : EvalModel (optionally wrapped in nn.DataParallel), on the GPU
    frames      : iterable/array of uint8 or float [H,W,3] images, BGR (cv2 order) unless frames_are_rgb
    trimap      : first-frame trimap, one-hot float [3,H,W] (demo flow, dataset.py:866-893) or None
    alphas      : per-frame GT alpha [H,W] in [0,1] (VideoMatting108 flow: first-frame trimap derived
                  from alpha with the model's dilate kernel) -- required when trimap is None
    gt_alpha_u8 : optional per-frame ground-truth alpha, uint8 [H,W]; with it SAD/MSE/dtSSD are accumulated on
                  the device (ClipMetrics) and returned under "metrics"; gt_mask_u8 = optional {0,1} evaluation masks,
                  gt_mask="unknown" = the reference metric's default mask (0 < gt < 255) instead
    backgrounds : optional per-frame BG images (V108 composites fg*a + bg*(1-a)), same dtype / channel order as the
                  frames; default bg = fg
    Returns dict(alpha=[T,H,W] float32, alpha_u8=[T,H,W] uint8 (truncated, eval.py:209), trimap=[T,3,H,W],
    bank_frames=[per frame: ids of the frames resident in the memory bank after that frame's update]).
    """
    frames = list(frames) if not (hasattr(frames, "shape") or hasattr(frames, "__getitem__")) else frames
    T = len(frames)
    dev = device or next(model.parameters()).device
    out_a, out_u8, out_t, bank_log = [], [], [], []
    core = model.module if hasattr(model, "module") else model
    metrics = ClipMetrics(dev) if gt_alpha_u8 is not None else None
    # loop invariants: the user trimap (25 MB as fp32 at 1080p) is uploaded once, not once per frame; the dummy alpha
    # of the trimap flow is one tensor for the whole clip
    tri_dev = None
    if trimap is not None:
        t = _as_tensor(trimap)
        tri_dev = t.to(dev).float()[None, None]
    ones = None
    for i in range(T):
        f = _as_tensor(frames[i])
        b = _as_tensor(backgrounds[i]) if backgrounds is not None else None
        extra = {}
        if f.dtype == torch.uint8 and (b is None or b.dtype == torch.uint8):
            # decoded images go to the device as they are ([H,W,3] uint8): the preprocess kernel converts, flips the
            # channel order and composites -- no per-frame torch conversion / transposition kernels
            ready = getattr(f, "_otvm_ready", None)           # upload event of the IO pipeline's prefetcher
            fg = f.to(dev, non_blocking=True)
            bg = fg if b is None else b.to(dev, non_blocking=True)
            if b is not None and getattr(b, "_otvm_ready", None) is not None:
                torch.cuda.current_stream(dev).wait_event(b._otvm_ready)
            H, W = fg.shape[:2]
            extra["_frames_rgb"] = bool(frames_are_rgb)
            if ready is not None:
                # the upload event may only travel to the model (whose query-encoder stream then starts on it, ahead of the
                # launch stream) when EVERYTHING the preprocess reads is covered by it or loop-invariant: the frame itself
                # (no copy made here), no separately uploaded background, and the trimap flow's constant alpha -- a per-frame
                # alpha is written by the launch stream just below.  Otherwise the launch stream waits for the upload
                # and the model orders its side streams behind the launch stream.
                if b is None and fg is f and trimap is not None:
                    extra["_inputs_ready"] = ready
                else:
                    torch.cuda.current_stream(dev).wait_event(ready)
        else:
            f = f.to(dev).float()
            if frames_are_rgb:
                f = f.flip(-1)
            fg = f.permute(2, 0, 1)[None, None].contiguous()
            H, W = fg.shape[-2:]
            if b is not None:                                 # backgrounds come in the same channel order as the frames
                b = b.to(dev).float()
                if frames_are_rgb:
                    b = b.flip(-1)
                bg = b.permute(2, 0, 1)[None, None].contiguous()
            else:
                bg = fg
        if trimap is not None:
            if ones is None or ones.shape[-2:] != (H, W):
                ones = torch.ones(1, 1, 1, H, W, device=dev)
            a, tri_gt = ones, tri_dev
        else:
            al = alphas[i]
            al = _as_tensor(al)
            a = al.to(dev).float()[None, None, None]
            tri_gt = None
        memorize, max_memory_num, large = memory_schedule(i, H, W, skip, max_num)
        out = model(a, fg, bg, tri=None, tri_gt=tri_gt, first_frame=(i == 0), last_frame=(i == T - 1),
                    memorize=memorize, max_memory_num=max_memory_num, large_input=large, **extra)
        alpha = out[3][0, 0, 0]
        u8 = core._engine.last_alpha_u8
        bank_log.append(list(core.memories["frames"]))          # frame ids resident after this frame's update (host list)
        if metrics is not None:
            g = gt_alpha_u8[i]
            g = _as_tensor(g)
            mk = gt_mask
            if gt_mask_u8 is not None:
                mk = gt_mask_u8[i]
                mk = (_as_tensor(mk)).to(dev)
            metrics.add(u8, g.to(dev), mk)
        if on_frame is not None:
            on_frame(i, alpha, u8, out)
        if keep_on_device:
            out_a.append(alpha), out_u8.append(u8), out_t.append(out[1][0, 0])
        else:
            out_a.append(alpha.cpu()), out_u8.append(u8.cpu()), out_t.append(out[1][0, 0].cpu())
    res = dict(alpha=torch.stack(out_a), alpha_u8=torch.stack(out_u8), trimap=torch.stack(out_t), bank_frames=bank_log)
    if metrics is not None:
        res["metrics"] = metrics.result()
    return res


def run_video_matte_batch(model, clips, trimaps=None, alphas=None, backgrounds=None, skip=10, max_num=5, frames_are_rgb=False,
                          device=None, keep_on_device=False, on_frame=None, gt_alpha_u8=None, gt_mask=None):
    """Matte B sequences of one resolution in LOCK-STEP (round 3): frame i of every clip goes through the network in one
    batched step (EvalModel.forward_batch: one launch per layer over the B images, per-sequence memory banks).
    clips: list of B frame arrays ([T_b,H,W,3] uint8 / float, BGR unless frames_are_rgb); trimaps: list of B first-frame
    one-hot trimaps [3,H,W] (demo flow) or None with alphas = list of B per-frame GT alpha lists (V108 flow: the first-frame
    trimap is derived from the alpha); backgrounds: optional list of B per-frame background lists; gt_alpha_u8: optional
    list of B per-frame uint8 ground truths (SAD / MSE / dtSSD per clip, as run_video_matte); on_frame(b, i, alpha, u8, out).
    Clips may differ in LENGTH: the batch runs max(T_b) steps, a clip that has ended keeps feeding its last frame (its
    outputs from then on are discarded) -- sequences are independent (all recurrent state is per sequence), so
    this changes no result of the others; the frame flags follow the frame index, which the clips share.
    Each returned dict's alpha / alpha_u8 / trimap / metrics equal run_video_matte of that clip alone (bit for bit under the
    same kernel configurations).  bank_frames follows the SHARED schedule: ``last_frame`` is keyed to the longest clip, so a
    shorter clip's final entry still lists the frame whose memorize is pending -- the stand-alone run, whose last frame
    memorises nothing, does not (the one documented difference).
    Returns a list of B dicts (alpha, alpha_u8, trimap, bank_frames[, metrics])."""
    B = len(clips)
    lens = [len(c) for c in clips]
    T = max(lens)
    if (trimaps is None) == (alphas is None):
        raise ValueError("run_video_matte_batch: give either trimaps (demo flow) or alphas (VideoMatting108 flow)")
    core = model.module if hasattr(model, "module") else model
    dev = device or next(core.parameters()).device
    res = [dict(alpha=[], alpha_u8=[], trimap=[], bank_frames=[]) for _ in range(B)]
    metrics = [ClipMetrics(dev) for _ in range(B)] if gt_alpha_u8 is not None else None
    tri_dev = None if trimaps is None else [_as_tensor(t).to(dev).float()[None, None] for t in trimaps]
    ones = None
    for i in range(T):
        A, FG, BG = [], [], []
        for b in range(B):
            j = min(i, lens[b] - 1)                               # a clip that has ended repeats its last frame (discarded)
            f = _as_tensor(clips[b][j])
            bk = _as_tensor(backgrounds[b][j]) if backgrounds is not None else None
            if f.dtype == torch.uint8 and (bk is None or bk.dtype == torch.uint8):
                fg = f.to(dev, non_blocking=True)
                bg = fg if bk is None else bk.to(dev, non_blocking=True)
                # frames handed over by the IO pipeline's prefetcher carry their upload event: the launch stream waits for
                # it before anything reads them (as run_video_matte does; the batched step makes no promise to the engine
                # about its inputs, so its side streams order themselves behind the launch stream)
                for src in (f, bk):
                    ev = getattr(src, "_otvm_ready", None) if src is not None else None
                    if ev is not None:
                        torch.cuda.current_stream(dev).wait_event(ev)
                H, W = fg.shape[:2]
                rgb = bool(frames_are_rgb)
            else:
                f = f.to(dev).float()
                if frames_are_rgb:
                    f = f.flip(-1)
                fg = f.permute(2, 0, 1)[None, None].contiguous()
                H, W = fg.shape[-2:]
                if bk is not None:
                    bk = bk.to(dev).float()
                    if frames_are_rgb:
                        bk = bk.flip(-1)
                    bg = bk.permute(2, 0, 1)[None, None].contiguous()
                else:
                    bg = fg
                rgb = False
            if trimaps is not None:
                if ones is None or ones.shape[-2:] != (H, W):
                    ones = torch.ones(1, 1, 1, H, W, device=dev)
                a = ones
            else:
                a = _as_tensor(alphas[b][j]).to(dev).float()[None, None, None]
            A.append(a), FG.append(fg), BG.append(bg)
        memorize, max_memory_num, large = memory_schedule(i, H, W, skip, max_num)
        outs = core.forward_batch(A, FG, BG, tri_dev if tri_dev is not None else [None] * B, first_frame=(i == 0),
                                  last_frame=(i == T - 1), memorize=memorize, max_memory_num=max_memory_num, large_input=large,
                                  _frames_rgb=rgb)
        u8s = core._engine.last_alpha_u8_b
        bank = list(core.memories["frames"])
        for b in range(B):
            if i >= lens[b]:
                continue
            al, u8, tr = outs[b][3][0, 0, 0], u8s[b], outs[b][1][0, 0]
            if metrics is not None:
                metrics[b].add(u8, _as_tensor(gt_alpha_u8[b][i]).to(dev), gt_mask)
            if on_frame is not None:
                on_frame(b, i, al, u8, outs[b])
            if not keep_on_device:
                al, u8, tr = al.cpu(), u8.cpu(), tr.cpu()
            res[b]["alpha"].append(al), res[b]["alpha_u8"].append(u8), res[b]["trimap"].append(tr)
            res[b]["bank_frames"].append(bank)
    out = []
    for b, r in enumerate(res):
        d = dict(alpha=torch.stack(r["alpha"]), alpha_u8=torch.stack(r["alpha_u8"]), trimap=torch.stack(r["trimap"]),
                 bank_frames=r["bank_frames"])
        if metrics is not None:
            d["metrics"] = metrics[b].result()
        out.append(d)
    return out


class ClipMetrics:
    """SAD / MSE / dtSSD of a clip accumulated on the device (otvm_matting_metrics), reference definitions
    utils/tmp/metric.py:177-189,252-264 on the 8-bit alphas the path writes (eval.py:209).  One row of partial sums
    per frame stays on the device (no per-frame synchronisation); result() turns them into the reference's per-frame
    values.  mask: None = all pixels, "unknown" = the reference's default (0 < target < 255, metric.py:113-115), or
    explicit uint8 {0,1} masks."""

    def __init__(self, device, capacity=256):
        from . import lib as L
        self.L, self.lib = L, L.load()
        self.device = device
        self.acc = torch.zeros(capacity, 5, dtype=torch.float64, device=device)
        self.prev = None
        self.frames = 0

    def add(self, pred_u8, target_u8, mask_u8=None):
        """pred/target: uint8 [H,W] device tensors (0..255); mask: uint8 {0,1}, "unknown" or None (all pixels)."""
        st = torch.cuda.current_stream(pred_u8.device).cuda_stream
        pred_u8, target_u8 = pred_u8.contiguous(), target_u8.contiguous()
        if isinstance(mask_u8, str):
            if mask_u8 != "unknown":
                raise ValueError("ClipMetrics: mask must be None, 'unknown' or a uint8 tensor")
            mask_u8 = ((target_u8 > 0) & (target_u8 < 255)).to(torch.uint8)
        mask_u8 = None if mask_u8 is None else mask_u8.contiguous()
        if self.frames == self.acc.shape[0]:
            self.acc = torch.cat([self.acc, torch.zeros_like(self.acc)])
        pp, tp, mp = self.prev if self.prev is not None else (None, None, None)
        ptr = lambda x: 0 if x is None else x.data_ptr()
        self.L.check(self.lib.otvm_matting_metrics(ptr(pred_u8), ptr(target_u8), ptr(mask_u8), ptr(pp), ptr(tp), ptr(mp),
                                                   pred_u8.numel(), self.acc[self.frames].data_ptr(), st), "matting_metrics")
        self.prev = (pred_u8, target_u8, mask_u8)
        self.frames += 1

    def result(self):
        rows = self.acc[:self.frames].cpu()
        tot = rows.sum(0).tolist()
        sad_f = (rows[:, 0] / 255.0 / 1000.0).tolist()
        mse_f = (rows[:, 1] / 255.0 ** 2 / (rows[:, 2] + 1.0)).tolist()
        # row i > 0 holds the temporal term of the pair (i-1, i), masked by frame i-1's mask (metric.py:252-264)
        dt_f = (rows[1:, 3] / 255.0 ** 2).sqrt().tolist()
        dt_n = (rows[1:, 4] + 1.0).tolist()
        return dict(frames=self.frames, sad_sum=tot[0] / 255.0 / 1000.0, mse_num=tot[1] / 255.0 ** 2, mask_sum=tot[2],
                    dt_err2_sum=tot[3] / 255.0 ** 2, dt_mask_sum=tot[4],
                    sad_mean=tot[0] / 255.0 / 1000.0 / max(1, self.frames),
                    sad_per_frame=sad_f, mse_per_frame=mse_f, dtssd_per_pair=dt_f, dtssd_num_per_pair=dt_n,
                    dtssd_sum=float(sum(dt_f)))


def sad(pred, ref, mask=None):
    """Sum of absolute differences / 1000 (reference utils/tmp/metric.py:177-182)."""
    d = (pred.float() - ref.float()).abs()
    if mask is not None:
        d = d * mask
    return float(d.sum()) / 1000.0
