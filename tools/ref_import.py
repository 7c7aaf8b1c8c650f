"""Import the read-only reference (/root/reference) on CPU for fixture generation ONLY.

Test infrastructure.  Never imported by the product (otvm_amd/), by bench.py's GPU leg,
or on the GPU box (the reference does not exist there).  Recipe = SURVEY.md A.5:
  * `cv2` stand-in: DIST_L2 + distanceTransform(src, DIST_L2, 0) via scipy exact EDT
  * `torchvision.models.resnet50` stand-in with the v1.5 Bottleneck layout
  * `os.popen('stty size')` patched (helpers.py:211 needs a TTY)
  * torch.cuda.current_device -> 'cpu' (models/trimap/model.py:228,242)
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"


def _make_cv2():
    from scipy import ndimage
    m = types.ModuleType("cv2")
    m.DIST_L2 = 2

    def distanceTransform(src, distanceType, maskSize):
        assert distanceType == 2 and maskSize == 0
        return ndimage.distance_transform_edt(src != 0).astype(np.float32)
    m.distanceTransform = distanceTransform
    return m


def _make_torchvision():
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")

    class Bottleneck(nn.Module):
        def __init__(self, inp, planes, stride=1, downsample=None):
            super().__init__()
            self.conv1 = nn.Conv2d(inp, planes, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
            self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
            self.bn3 = nn.BatchNorm2d(planes * 4)
            self.relu = nn.ReLU(inplace=True)
            self.downsample = downsample

        def forward(self, x):
            idt = x
            o = self.relu(self.bn1(self.conv1(x)))
            o = self.relu(self.bn2(self.conv2(o)))
            o = self.bn3(self.conv3(o))
            if self.downsample is not None:
                idt = self.downsample(x)
            return self.relu(o + idt)

    class ResNet50(nn.Module):
        def __init__(self):
            super().__init__()
            self.inplanes = 64
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            self.layer1 = self._make(64, 3, 1)
            self.layer2 = self._make(128, 4, 2)
            self.layer3 = self._make(256, 6, 2)
            self.layer4 = self._make(512, 3, 2)

        def _make(self, planes, n, stride):
            ds = None
            if stride != 1 or self.inplanes != planes * 4:
                ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False),
                                   nn.BatchNorm2d(planes * 4))
            layers = [Bottleneck(self.inplanes, planes, stride, ds)]
            self.inplanes = planes * 4
            for _ in range(1, n):
                layers.append(Bottleneck(self.inplanes, planes))
            return nn.Sequential(*layers)

    models.resnet50 = lambda pretrained=False, **kw: ResNet50()
    tv.models = models
    utils = types.ModuleType("torchvision.utils")
    utils.save_image = lambda *a, **k: None
    tv.utils = utils
    return tv, models, utils


_loaded = {}


def load_reference():
    """Returns the reference `helpers` module (factories) with shims installed."""
    if "helpers" in _loaded:
        return _loaded["helpers"]
    sys.modules["cv2"] = _make_cv2()
    tv, models, utils = _make_torchvision()
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models
    sys.modules["torchvision.utils"] = utils
    _popen = os.popen

    class _Fake:
        def read(self):
            return "24 80"
    os.popen = lambda cmd, *a, **k: _Fake() if "stty" in cmd else _popen(cmd, *a, **k)
    torch.cuda.current_device = lambda: "cpu"
    torch.set_grad_enabled(False)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import helpers  # noqa: E402  (reference helpers.py)
    os.popen = _popen
    _loaded["helpers"] = helpers
    return helpers


def build_reference_model(dilate_kernel=12):
    helpers = load_reference()
    cfg = types.SimpleNamespace(TRAIN=types.SimpleNamespace(STAGE=4))
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        mt = helpers.get_model_trimap(cfg, "Test", dilate_kernel)
        m = helpers.get_model_alpha(cfg, mt, "Test", dilate_kernel)
    return m.eval()


if __name__ == "__main__":
    m = build_reference_model()
    sd = m.state_dict()
    print(len(sd), sum(v.numel() for v in sd.values()))
    import json
    spec = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
    json.dump(spec, open("/root/repo/tests/golden/state_dict_spec.json", "w"))
