"""Sequence enumeration and frame decoding of the reference's two evaluation datasets (host side, PIL + numpy).

Mirrors the iteration protocol of the reference so a loop written against it keeps working:

    for data_name, data_root, FG, BG, a, tri, seq_name in Demo_Test(root):            # dataset.py:1019-1070
    for data_name, data_root, FG, BG, a, tri, seq_name in VideoMatting108_Test(root): # dataset.py:959-1017

    demo  : <root>/<seq>/frames/*  and  <root>/<seq>/trimap/<frame stem>.png  (the most recent existing trimap
            path is repeated for later frames; '' before the first one)
    V108  : <root>/VideoMatting108/{frame_corr.json, val_videos.txt, FG_done/<video>/<clip>/*.png (RGBA),
            BG_done2/...}; frame_corr maps an FG frame to its BG frame; a video's frames are the sorted keys of
            frame_corr whose dirname is the line of the set file

Decoding follows EvalDataset.get_data (dataset.py:857-920): images in cv2 channel order (BGR), the V108 foreground
comes with its alpha in the 4th channel (alpha / 255, eps = [0, 1] clamps are no-ops), a missing BG '.jpg' falls
back to '.png'.
"""
import json
import os

import numpy as np


class Demo_Test:
    def __init__(self, data_root):
        self.idx = 0
        self.data_root = data_root
        self.FG, self.TRI, self.seq_name = self.parse_DemoVideo(data_root)
        self.FG_len, self.TRI_len = len(self.FG), len(self.TRI)

    def __len__(self):
        return self.FG_len

    @staticmethod
    def parse_DemoVideo(data_root):
        FG, TRI, seq_name = [], [], []
        for v in sorted(os.listdir(data_root)):
            fdir = os.path.join(data_root, v, "frames")
            if not os.path.isdir(fdir):
                continue
            fg_cur, tri_cur, tri_exist = [], [], ""
            for img_name in sorted(os.listdir(fdir)):
                fg_cur.append(os.path.join(v, "frames", img_name))
                tri_path = os.path.join(v, "trimap", os.path.splitext(img_name)[0] + ".png")
                if os.path.isfile(os.path.join(data_root, tri_path)):
                    tri_exist = tri_path
                tri_cur.append(tri_exist)
            FG.append(fg_cur), TRI.append(tri_cur), seq_name.append(v)
        return FG, TRI, seq_name

    def __iter__(self):
        return self

    def __next__(self):
        if self.idx >= len(self):
            raise StopIteration
        i = self.idx
        self.idx += 1
        return "demo", self.data_root, self.FG[i], None, None, self.TRI[i], self.seq_name[i]


class VideoMatting108_Test:
    FG_FOLDER = "FG_done"
    BG_FOLDER = "BG_done2"

    def __init__(self, data_root, mode="val", use_subset=False):
        assert mode in ("train", "val")
        self.idx = 0
        self.mode = mode
        self.data_root_V108 = os.path.join(data_root, "VideoMatting108")
        setname = ("{}_videos_subset.txt" if use_subset else "{}_videos.txt").format(mode)
        with open(os.path.join(self.data_root_V108, "frame_corr.json")) as f:
            self.frame_corr = json.load(f)
        with open(os.path.join(self.data_root_V108, setname)) as f:
            self.FG, self.BG, self.seq_name = self.parse_VideoMatting108(f, self.frame_corr)
        self.FG_len, self.BG_len = len(self.FG), len(self.BG)

    def __len__(self):
        return self.FG_len

    def parse_VideoMatting108(self, lines, frame_corr):
        FG, BG, seq_name = [], [], []
        keys = sorted(frame_corr.keys())
        for v in lines:
            v = v.strip()
            fns = [k for k in keys if os.path.dirname(k) == v]
            FG.append([os.path.join(self.FG_FOLDER, k) for k in fns])
            BG.append([os.path.join(self.BG_FOLDER, frame_corr[k]) for k in fns])
            seq_name.append(v)
        return FG, BG, seq_name

    def __iter__(self):
        return self

    def __next__(self):
        if self.idx >= len(self):
            raise StopIteration
        i = self.idx
        self.idx += 1
        return "V108", self.data_root_V108, self.FG[i], self.BG[i], None, None, self.seq_name[i]


def _imread(path, mode=None):
    from PIL import Image
    im = Image.open(path)
    if mode is not None:
        im = im.convert(mode)
    return np.asarray(im)


def read_bgr(path):
    """cv2.imread(path, IMREAD_COLOR) equivalent: uint8 [H,W,3] in BGR order."""
    return np.ascontiguousarray(_imread(path, "RGB")[..., ::-1])


def read_fg_with_alpha(path):
    """V108 foreground frame (dataset.py:862-866): returns (fg uint8 [H,W,3] BGR, alpha uint8 [H,W]).  The reference
    divides the alpha by 255 and clamps with eps = [0, 1] (no-ops); callers that feed run_video_matte do the same."""
    im = _imread(path)
    if im.ndim != 3 or im.shape[-1] != 4:
        raise ValueError("VideoMatting108 foreground %s has no alpha channel (shape %s)" % (path, im.shape))
    return np.ascontiguousarray(im[..., 2::-1]), np.ascontiguousarray(im[..., 3])


def read_trimap_unchanged(path):
    """cv2.imread(path, IMREAD_UNCHANGED) for a trimap file (dataset.py:879): grayscale stays 2-D (8 or 16 bit),
    colour comes back in cv2's BGR(A) order (trimap_file_to_onehot reads unknown from channel 2 = R and foreground from
    channel 1 = G), palette images are expanded to colour as OpenCV does."""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("L", "I;16", "I;16B", "I"):
        return np.asarray(im)
    if im.mode == "1":
        return np.asarray(im.convert("L"))
    if im.mode in ("RGBA", "LA", "PA") or (im.mode == "P" and "transparency" in im.info):
        a = np.asarray(im.convert("RGBA"))
        return np.ascontiguousarray(np.concatenate([a[..., 2::-1], a[..., 3:]], -1))
    return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])


def resolve_bg(path):
    """dataset.py:896-899: a listed background that does not exist is looked up with a .png extension."""
    return path if os.path.exists(path) else os.path.splitext(path)[0] + ".png"


def load_sequence(item, max_frames=None, decode_frames=True):
    """Decode one item of either iterator into what run_video_matte takes.

    Returns dict(name, names=[file stems], frames=[uint8 BGR], and either trimap=one-hot [3,H,W] (demo) or
    alphas=[float32 [H,W] in 0..1], backgrounds=[uint8 BGR], gt_alpha_u8=[uint8 [H,W]] (V108)).
    """
    from .video import trimap_file_to_onehot
    data_name, root, FG, BG, _a, TRI, seq_name = item
    n = len(FG) if max_frames is None else min(len(FG), max_frames)
    names = [os.path.splitext(os.path.basename(p))[0] for p in FG[:n]]
    out = dict(name=seq_name, names=names, data_name=data_name)
    if data_name == "demo":
        if decode_frames:                          # False: the caller decodes ahead itself from "frame_paths"
            out["frames"] = [read_bgr(os.path.join(root, p)) for p in FG[:n]]
        # the reference reads the trimap listed for each frame (dataset.py:879) and the model consumes the first
        # frame's: a clip whose FIRST frame has no trimap file cannot be evaluated (cv2.imread('') -> None there)
        tri = TRI[0] if TRI else ""
        if not tri:
            raise FileNotFoundError("sequence %s: no trimap for its first frame (%s/%s/trimap/%s.png)"
                                    % (seq_name, root, seq_name, names[0] if names else "?"))
        out["trimap"] = trimap_file_to_onehot(read_trimap_unchanged(os.path.join(root, tri)))
        out["frame_paths"] = [os.path.join(root, p) for p in FG[:n]]
        return out
    frames, gts = [], []
    for p in FG[:n]:
        f, a = read_fg_with_alpha(os.path.join(root, p))
        frames.append(f), gts.append(a)
    out["frames"] = frames
    out["gt_alpha_u8"] = gts
    out["alphas"] = [g.astype(np.float32) / 255.0 for g in gts]
    out["backgrounds"] = [read_bgr(resolve_bg(os.path.join(root, p))) for p in BG[:n]]
    return out
