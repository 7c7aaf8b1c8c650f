"""Mirror of reference ``models/alpha/model.py::EvalModel`` (lines 314-512): THE drop-in boundary.

Same constructor (``dilate_kernel, trimap, stage``), same 785-key ``state_dict``, same stateful
``forward(a, fg, bg, tri=None, tri_gt=None, first_frame=False, last_frame=False, memorize=False,
max_memory_num=2, large_input=False)`` returning the same 5-tuple.  All device work goes through
``libotvm_hip.so``; there is no PyTorch/CPU fallback -- on a CPU device ``forward`` raises.
"""
import torch
from torch import nn

from .engine import HipEngine
from .modules import attach_from_spec


class EvalModel(nn.Module):
    def __init__(self, dilate_kernel=None, eps=0, trimap=None, stage=1):
        super().__init__()
        if stage != 4 or trimap is None:
            raise NotImplementedError("otvm_amd implements the stage-4 (joint trimap+alpha) inference path only")
        self.stage = stage
        self.refinement = True
        self.DILATION_KERNEL = dilate_kernel
        self.EPS = eps
        self.IMG_SCALE = 1.0 / 255
        self.TRIMAP_CHANNEL = 8
        self.memory_update = False
        attach_from_spec(self, "", "")           # everything except the trimap.* keys ...
        for k in [k for k in self._modules if k == "trimap"]:
            del self._modules[k]
        self.trimap = trimap                      # ... which live in the FullModel_eval passed in (alpha/model.py:37)
        self._engine = None
        self._engine_key = None
        self.precision = None                     # None -> OTVM_PRECISION env or "f16x3"; "f32" = exact-fp32 MFMA

    # -- engine lifetime: rebuilt when the weights change or the module moves
    def _get_engine(self):
        dev = self.IMG_MEAN.device
        if dev.type != "cuda":
            raise RuntimeError("otvm_amd.EvalModel: parameters are on %s; move the model to the GPU (.cuda()) -- "
                               "the HIP path has no CPU fallback" % dev)
        key = (str(dev), self.precision, tuple(p._version for p in self.parameters()),
               tuple(b._version for b in self.buffers()))
        if self._engine is None or key != self._engine_key:
            self._engine = HipEngine(self.state_dict(), dev, precision=self.precision)
            self._engine_key = key
        return self._engine

    @property
    def memories(self):
        """Bank introspection (reference: self.memories['key'].shape[3] slots)."""
        eng = self._engine
        if eng is None:
            return {"frames": []}
        return {"frames": eng.bank_frames()}

    @torch.no_grad()
    def forward(self, a, fg, bg, tri=None, tri_gt=None, first_frame=False, last_frame=False, memorize=False,
                max_memory_num=2, large_input=False, _frame_id=None, _cls_override=None, _frames_rgb=False,
                _inputs_ready=None):
        if tri is not None:
            # alpha/model.py:395-396: unreachable from eval.py (EvalDataset is built with trimap=None, eval.py:133)
            raise NotImplementedError("per-frame `tri` input is not part of the reference eval path")
        eng = self._get_engine()
        out = eng.frame(a, fg, bg, tri_gt=tri_gt, first_frame=bool(first_frame), last_frame=bool(last_frame),
                        memorize=bool(memorize), max_memory_num=int(max_memory_num),
                        dilate_kernel=self.DILATION_KERNEL, frame_id=_frame_id, cls_override=_cls_override,
                        frames_rgb=bool(_frames_rgb), inputs_ready=_inputs_ready)
        self.memory_update = memorize
        return out

    @torch.no_grad()
    def forward_batch(self, a, fg, bg, tri_gt, first_frame=False, last_frame=False, memorize=False, max_memory_num=2,
                      large_input=False, _frames_rgb=False, _inputs_ready=None, _cls_override=None):
        """Round 3 extension (not part of the reference surface): the same frame step for B independent sequences stepped in
        LOCK-STEP -- ``a``, ``fg``, ``bg``, ``tri_gt`` are lists of B per-sequence inputs shaped as ``forward`` takes them
        (one resolution, one memory schedule; every sequence keeps its own memory bank).  Every layer runs as one launch
        over the B images, which fills the chip on the small maps a single sequence leaves mostly idle.  Returns a list of
        B 5-tuples; each equals what ``forward`` returns for that sequence run alone with the same kernel configurations."""
        eng = self._get_engine()
        out = eng.frame_batch(list(a), list(fg), list(bg), list(tri_gt), first_frame=bool(first_frame), last_frame=bool(last_frame),
                              memorize=bool(memorize), max_memory_num=int(max_memory_num), dilate_kernel=self.DILATION_KERNEL,
                              cls_override=_cls_override, frames_rgb=bool(_frames_rgb), inputs_ready=_inputs_ready)
        self.memory_update = memorize
        return out


class FullModel(EvalModel):
    """Mirror of the reference's TRAINING class ``models/alpha/model.py::FullModel`` (lines 9-312), forward only: same
    constructor and state_dict as ``EvalModel``, ``forward(a, fg, bg, ignore_region=None, tri=None)`` over a batch of clips
    ``[B, S, C, H, W]`` returning the reference's list ``[loss1, loss2, loss3, loss_trimap, scaled_imgs, tris_vis, alphas, comps,
    scaled_gts, Fs, Bs, preds_trimap]`` (otvm_amd/train.py).  There is no backward: the library holds no gradient kernels."""
    FBA_LOSS_NORMALIZE = True

    @torch.no_grad()
    def forward(self, a, fg, bg, ignore_region=None, tri=None):
        if ignore_region is not None:
            raise NotImplementedError("ignore_region is not used by the stage-4 training loop of the reference")
        from .train import train_forward
        return train_forward(self, a, fg, bg, tri)
