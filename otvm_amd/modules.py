"""Parameter containers that reproduce the reference modules' ``state_dict`` layout.

The product does no compute in these modules: they only own the tensors under the reference's key
names (so ``load_state_dict(torch.load('weights/s4_OTVM.pth'))`` works, strict), and
``EvalModel.forward`` hands them to the HIP engine.
"""
import torch
from torch import nn

from .state_spec import state_dict_spec

_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked", "IMG_MEAN", "IMG_STD", "mean", "std",
                  "KERNEL")


class _Node(nn.Module):
    """A bare container; children/parameters are attached by dotted path."""


def _is_buffer(key):
    leaf = key.split(".")[-1]
    return leaf in _BUFFER_LEAVES or key.endswith("LOSS.weight")


def attach_from_spec(root, prefix, strip):
    """Create parameters/buffers for every spec key starting with ``prefix`` under ``root``;
    ``strip`` is removed from the front of the key to get the path relative to ``root``."""
    dtypes = {"float32": torch.float32, "int64": torch.int64}
    for key, (shape, dt) in state_dict_spec().items():
        if not key.startswith(prefix):
            continue
        rel = key[len(strip):]
        parts = rel.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Node())
            mod = mod._modules[p]
        t = torch.zeros(shape, dtype=dtypes[dt])
        if _is_buffer(key):
            mod.register_buffer(parts[-1], t)
        else:
            mod.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))
