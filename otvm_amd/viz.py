"""Visualisation composite of the reference's `--viz` mode (eval.py:96-115, 199-203, 229-242).

Per frame the reference stacks six half-resolution panels, two per row:

    frame (RGB 0..1)            | frame over a green background with the predicted alpha
    first-frame trimap (GT)     | ground-truth alpha (3 channels)
    predicted trimap (softmax)  | predicted alpha (3 channels)

through `F.interpolate(size=(h//2, w//2), bilinear, align_corners=False)` and `torchvision.utils.save_image(..., nrow=2)`
(make_grid defaults: padding 2, pad value 0; `x*255 + 0.5`, clamp, truncate to uint8).  torchvision is not a dependency
here: the grid layout is restated below (tests/test_host_logic.py checks it against the documented layout).
"""
import os
import shutil
import subprocess

import torch
import torch.nn.functional as F

PAD = 2


def viz_panels(out5):
    """out5 = the 5-tuple of EvalModel.forward for one frame -> [6,3,h//2,w//2] float tensor (device of the inputs)."""
    scaled_imgs, tri_pred, tri_gt, alphas, scaled_gts = out5
    b, s, _, h, w = scaled_imgs.shape
    green = torch.zeros_like(scaled_imgs)
    green[:, :, 1] = 1.0
    comps = scaled_imgs * alphas + green * (1.0 - alphas)                       # eval.py:199-201
    a3 = alphas.expand(-1, -1, 3, -1, -1)
    g3 = scaled_gts.expand(-1, -1, 3, -1, -1)
    panels = [scaled_imgs, comps, tri_gt, g3, tri_pred, a3]                      # eval.py:104-110
    imgs = torch.cat([p.reshape(b * s, 3, h, w) for p in panels], dim=0).float()
    return F.interpolate(imgs, size=(h // 2, w // 2), mode="bilinear", align_corners=False)


def make_grid_u8(imgs, nrow=2):
    """[N,3,h,w] float -> uint8 [H,W,3] laid out like torchvision.utils.make_grid(padding=2) + save_image's rounding."""
    n, c, h, w = imgs.shape
    xmaps = min(nrow, n)
    ymaps = (n + xmaps - 1) // xmaps
    grid = imgs.new_zeros((c, ymaps * (h + PAD) + PAD, xmaps * (w + PAD) + PAD))
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * (h + PAD) + PAD:y * (h + PAD) + PAD + h, x * (w + PAD) + PAD:x * (w + PAD) + PAD + w] = imgs[k]
            k += 1
    return grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()


def write_viz_frame(path, out5):
    """One 'f%d.jpg' of the reference's viz folder."""
    from PIL import Image
    Image.fromarray(make_grid_u8(viz_panels(out5), nrow=2)).save(path)


def make_viz_video(frame_pattern, vid_path, framerate=10):
    """eval.py:238-242: ffmpeg over the written frames; skipped (returns False) when ffmpeg is not installed."""
    exe = shutil.which("ffmpeg")
    if exe is None:
        return False
    os.makedirs(os.path.dirname(vid_path) or ".", exist_ok=True)
    subprocess.run([exe, "-framerate", str(framerate), "-i", frame_pattern, vid_path, "-nostats", "-loglevel", "0", "-y"],
                   check=False)
    return os.path.exists(vid_path)
