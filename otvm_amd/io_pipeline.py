"""Frame IO around ``run_video_matte``: decode, upload, download and encode overlap the GPU.

The reference decodes each frame with cv2.imread inside the frame loop, uploads it synchronously through
DataParallel's scatter, and blocks on ``.cpu()`` + ``cv2.imwrite`` for every alpha (dataset.py:857-920,
eval.py:209-217).  At >30 frames/s that host work is the bottleneck, so here
  * a thread pool decodes ahead (PIL releases the GIL while decoding) into pinned host buffers,
  * uploads run on a copy stream and are handed to the compute stream through events,
  * the 8-bit alpha is copied back asynchronously into pinned memory and PNG-encoded by another pool.
Image codecs: PIL (cv2 is not in this image).  Frames are handed to the model as RGB (``frames_are_rgb``).
"""
import io
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def _decode(src):
    from PIL import Image
    if isinstance(src, (bytes, bytearray)):
        im = Image.open(io.BytesIO(src))
    else:
        im = Image.open(src)
    return np.asarray(im.convert("RGB"))


class FramePrefetcher:
    """Iterates device tensors [H,W,3] uint8 (RGB) of ``sources`` (paths or encoded bytes), decoded ``depth`` ahead."""

    def __init__(self, sources, device, workers=4, depth=6):
        self.sources = list(sources)
        self.dev = torch.device(device)
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.depth = depth
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self._futs = {}
        self._next = 0

    def __len__(self):
        return len(self.sources)

    def _submit(self):
        while self._next < len(self.sources) and len(self._futs) < self.depth:
            self._futs[self._next] = self.pool.submit(self._load, self.sources[self._next])
            self._next += 1

    def _load(self, src):
        arr = np.ascontiguousarray(_decode(src))
        host = torch.empty(arr.shape, dtype=torch.from_numpy(np.empty(0, arr.dtype)).dtype, pin_memory=True)
        host.numpy()[...] = arr                       # one copy, straight into the pinned staging buffer
        return host

    def __iter__(self):
        self._submit()
        for i in range(len(self.sources)):
            host = self._futs.pop(i).result()
            self._submit()
            with torch.cuda.stream(self.copy_stream):
                d = host.to(self.dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            # the consumer waits for the upload itself (run_video_matte hands the event to the model as _inputs_ready:
            # launch stream and query-encoder stream both wait on it, and the query encoder of this frame may start under
            # the previous frame's alpha network instead of behind it)
            d.record_stream(torch.cuda.current_stream(self.dev))
            d._otvm_ready = ev
            yield d

    def close(self):
        self.pool.shutdown(wait=False)


class AlphaWriter:
    """Asynchronous sink for 8-bit alphas: D2H on a copy stream into pinned memory, PNG encoding in a thread pool."""

    def __init__(self, device, outdir=None, names=None, workers=4, keep=False):
        self.dev = torch.device(device)
        self.outdir, self.names, self.keep = outdir, names, keep
        if outdir:
            os.makedirs(outdir, exist_ok=True)
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.futs = []
        self.encoded = {}

    def put(self, i, u8):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ev)
            host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
            host.copy_(u8, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        u8.record_stream(self.copy_stream)
        self.futs.append(self.pool.submit(self._encode, i, host, done))

    def _encode(self, i, host, done):
        from PIL import Image
        done.synchronize()
        im = Image.fromarray(host.numpy())
        if self.outdir:
            name = self.names[i] if self.names else "%05d.png" % i
            im.save(os.path.join(self.outdir, name), compress_level=1)
        if self.keep:
            buf = io.BytesIO()
            im.save(buf, format="PNG", compress_level=1)
            self.encoded[i] = buf.getvalue()
        return i

    def close(self):
        for f in self.futs:
            f.result()
        self.pool.shutdown(wait=True)


def run_video_matte_io(model, sources, trimap, skip=10, max_num=5, outdir=None, names=None, device=None,
                       decode_workers=4, encode_workers=4, keep_encoded=False):
    """``run_video_matte`` with overlapped IO.  sources: image paths or encoded bytes (JPEG/PNG), RGB order."""
    from .video import run_video_matte
    dev = device or next(model.parameters()).device
    frames = FramePrefetcher(sources, dev, workers=decode_workers)
    writer = AlphaWriter(dev, outdir, names, workers=encode_workers, keep=keep_encoded)
    n = {"frames": 0}

    def sink(i, alpha, u8, out):
        writer.put(i, u8)
        n["frames"] += 1
    res = run_video_matte(model, _Listish(frames), trimap=trimap, skip=skip, max_num=max_num, frames_are_rgb=True,
                          on_frame=sink, device=dev, keep_on_device=True)
    writer.close()
    frames.close()
    res["encoded"] = writer.encoded
    return res


class _Listish:
    """Adapter: run_video_matte indexes ``frames[i]`` sequentially and asks for len()."""

    def __init__(self, prefetcher):
        self.p = prefetcher
        self.it = iter(prefetcher)
        self.i = 0
        self.shape = None

    def __len__(self):
        return len(self.p)

    def __getitem__(self, i):
        assert i == self.i, "frames are consumed in order"
        self.i += 1
        return next(self.it)
