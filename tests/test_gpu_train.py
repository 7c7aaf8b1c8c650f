"""GPU (-m gpu): the training-mode forward (forward only) -- loss kernels against the oracle's restatement
of utils/loss_func.py, and FullModel.forward against the CPU oracle and the reference-generated fixtures."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_loss_kernels_vs_oracle_functions():
    """csrc/losses.hip through the C ABI against oracle/train_oracle.py's functions (= utils/loss_func.py) on random data."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import train_oracle as T
    from otvm_amd import lib as L
    from otvm_amd.train import _fba_loss
    lib = L.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    B, S, H, W = 2, 3, 64, 96
    pred = torch.rand(B, S, 7, H, W, generator=g)
    gts = torch.rand(B, S, 1, H, W, generator=g)
    gts[gts < 0.3] = 0.0
    gts[gts > 0.8] = 1.0
    tm = (torch.rand(B, S, 1, H, W, generator=g) > 0.5).float()
    fgs, bgs, imgs = (torch.rand(B, S, 3, H, W, generator=g) for _ in range(3))
    want = T.fba_loss(pred, tm, gts, fgs, bgs, imgs)
    got = _fba_loss(lib, st, dev, *(x.to(dev).contiguous() for x in (pred, gts, tm, fgs, bgs, imgs)), B, S, H, W)
    for i, name in enumerate(("L_alpha_comp", "L_lap", "L_grad")):
        assert abs(got[i] - float(want[i])) <= 2e-5 * max(1.0, abs(float(want[i]))), (name, got[i], float(want[i]))
    for i, name in ((3, "alphas"), (4, "comps"), (5, "Fs"), (6, "Bs")):
        assert float((got[i].cpu() - want[i]).abs().max()) <= 1e-6, name
    # cross-entropy, trimask, scale / flip
    lg = torch.randn(5, 3, H, W, generator=g) * 8
    tri = torch.rand(5, 3, H, W, generator=g)
    mask, vis = torch.empty(5, H, W, device=dev), torch.empty(5, H, W, device=dev)
    cls = torch.empty(5, H, W, dtype=torch.uint8, device=dev)
    gt5 = torch.rand(5, H, W, generator=g)
    tri_d, gt5_d, lg_d = tri.to(dev), gt5.to(dev), lg.to(dev)        # (kept alive: the raw pointers go to the C ABI)
    L.check(lib.otvm_trimask(tri_d.data_ptr(), 5, H * W, mask.data_ptr(), cls.data_ptr(), gt5_d.data_ptr(), vis.data_ptr(), st))
    assert torch.equal(cls.cpu().long(), tri.max(dim=1)[1]) and torch.equal(mask.cpu(), (tri.max(dim=1)[1] == 1).float())
    assert torch.equal(vis.cpu(), torch.where(mask.cpu().bool(), torch.ones_like(gt5) * 128 * (1. / 255), gt5))
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    L.check(lib.otvm_loss_ce3(lg_d.data_ptr(), cls.data_ptr(), 5, H * W, acc.data_ptr(), st))
    want_ce = float(F.cross_entropy(lg, tri.max(dim=1)[1]))
    assert abs(float(acc[0]) / (5 * H * W) - want_ce) <= 2e-5 * want_ce
    x = torch.rand(4, 3, H, W, generator=g) * 255
    y = torch.empty(4, 3, H, W, device=dev)
    x_d = x.to(dev)
    L.check(lib.otvm_scale_flip3(x_d.data_ptr(), 4, H * W, 1.0 / 255, y.data_ptr(), st))
    assert torch.equal(y.cpu(), x.flip([1]) * (1.0 / 255))


@pytest.mark.parametrize("name", ["b2_s3_64x64", "b1_s4_64x96"])
def test_training_forward_vs_oracle_and_reference_fixture(name, synth_sd):
    """FullModel.forward (models/alpha/model.py:189-312) on the HIP path: B clips in lock-step through the batched kernels, every
    frame memorised, frame 0 with the ground-truth trimap; the four losses within 1e-3 (relative) of the CPU oracle and of the
    reference's own outputs, the returned tensors within 1e-3."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle.otvm_oracle import OtvmOracle
    from oracle.train_oracle import train_forward as oracle_forward
    from otvm_amd import helpers
    from otvm_amd.synth_data import train_batch
    g = np.load(os.path.join(GOLDEN, "train_%s.npz" % name))
    B, S, H, W, seed = (int(g[k]) for k in ("B", "S", "H", "W", "seed"))
    a, fg, bg, tri = (torch.from_numpy(x) for x in train_batch(B, S, H, W, seed))
    cfg = helpers.default_cfg()
    m = helpers.get_model_alpha(cfg, helpers.get_model_trimap(cfg, "Train", None), "Train", None)
    m.load_state_dict(synth_sd, strict=True)
    m = m.cuda().eval()
    out = m(a.cuda(), fg.cuda(), bg.cuda(), tri=tri.cuda())
    torch.cuda.synchronize()
    ref = oracle_forward(OtvmOracle(synth_sd), a, fg, bg, tri)
    names = ("loss1", "loss2", "loss3", "loss_trimap", "scaled_imgs", "tris_vis", "alphas", "comps", "scaled_gts", "Fs", "Bs", "preds_trimap")
    got = dict(zip(names, out))
    for k in ("loss1", "loss2", "loss3", "loss_trimap"):
        v, wo, wg = float(got[k]), float(ref[k]), float(g[k])
        print("%s %s: hip %.6f oracle %.6f reference %.6f" % (name, k, v, wo, wg))
        assert abs(v - wo) <= 1e-3 * max(1.0, abs(wo)) and abs(v - wg) <= 1e-3 * max(1.0, abs(wg)), (k, v, wo, wg)
    for k in ("alphas", "comps", "Fs", "Bs", "preds_trimap", "scaled_imgs", "tris_vis", "scaled_gts"):
        d = float((got[k].cpu() - torch.from_numpy(g[k])).abs().max())
        print("%s %s: max-abs vs the reference %.3e" % (name, k, d))
        assert d <= 1e-3, (k, d)
    assert m.memories["frames"] == list(range(S - 1))               # every frame but the last was memorised, none evicted
