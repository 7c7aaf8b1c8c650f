"""Mirror of reference ``models/trimap/model.py`` (``FullModel_eval``, lines 173-281) for stage 4.

Owns the STM weights under ``model.*`` exactly like the reference (Encoder_M / Encoder_Q / KV heads /
Decoder, reference models/trimap/STM.py:179-191).  The memorize / segment dispatch of the reference
(``forward(..., memorize=True | segment=True)``, trimap/model.py:247-264) is executed by the owning
``EvalModel`` on the HIP engine, which needs both networks' buffers in one launch plan.
"""
import torch
from torch import nn

from .modules import attach_from_spec


class FullModel_eval(nn.Module):
    def __init__(self, dilate_kernel=None, eps=0, ignore_label=255, stage=1, hdim=-1):
        super().__init__()
        if stage != 4:
            raise NotImplementedError("otvm_amd implements the stage-4 inference path only (got stage=%r)" % stage)
        self.DILATION_KERNEL = dilate_kernel
        self.EPS = eps
        self.stage = stage
        self.hdim = hdim if stage > 2 else -1
        self.num_object = 1
        self.ignore_label = ignore_label
        attach_from_spec(self, "trimap.", "trimap.")

    def forward(self, *args, **kwargs):
        raise RuntimeError("otvm_amd.FullModel_eval is driven by EvalModel.forward (HIP engine); "
                           "call the alpha model as the reference's eval.py does")
