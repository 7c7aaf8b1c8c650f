"""Factories with the reference's names/signatures (reference helpers.py:323-362)."""
from types import SimpleNamespace


def default_cfg():
    """The three knobs eval.py reads (reference config.py:22-23, eval.py:32)."""
    return SimpleNamespace(TRAIN=SimpleNamespace(STAGE=4), TEST=SimpleNamespace(MEMORY_MAX_NUM=5, MEMORY_SKIP_FRAME=10))


def get_model_name(cfg):
    return {1: "s1_OTVM_alpha", 2: "s2_OTVM_alpha", 3: "s3_OTVM", 4: "s4_OTVM"}[cfg.TRAIN.STAGE]


def get_model_trimap(cfg, mode="Test", dilate_kernel=None):
    # mode='Train' (helpers.py:332-346 returns the training class there): the same parameter container -- the training
    # forward (otvm_amd/train.py, forward only) is driven by the alpha model as well
    if mode not in ("Test", "Train"):
        raise ValueError("mode must be 'Test' or 'Train'")
    from .trimap_model import FullModel_eval
    return FullModel_eval(eps=0, stage=cfg.TRAIN.STAGE, dilate_kernel=dilate_kernel, hdim=16)


def get_model_alpha(cfg, model_trimap, mode="Test", dilate_kernel=None):
    if mode not in ("Test", "Train"):
        raise ValueError("mode must be 'Test' or 'Train'")
    from .alpha_model import EvalModel, FullModel
    cls = EvalModel if mode == "Test" else FullModel          # 'Train': forward-only mirror of models/alpha/model.py::FullModel
    return cls(dilate_kernel=dilate_kernel, trimap=model_trimap, stage=cfg.TRAIN.STAGE)
