"""CPU: the oracle (oracle/otvm_oracle.py) against fixtures produced by the reference itself."""
import os

import numpy as np
import pytest
import torch

from oracle import otvm_oracle as O
from tests.common import GOLDEN, clip_inputs, frame_flags, load_golden, load_sequences_meta

META = load_sequences_meta()


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(GOLDEN, "ops.npz"))


def test_state_dict_spec_matches_reference():
    import json
    from otvm_amd.state_spec import state_dict_spec
    gold = json.load(open(os.path.join(GOLDEN, "state_dict_spec.json")))
    spec = state_dict_spec()
    assert [k for k, _, _ in gold] == list(spec.keys())
    for k, shape, dt in gold:
        assert tuple(shape) == tuple(spec[k][0]) and dt == spec[k][1], k
    assert len(spec) == 785


def test_trimap_transform_vs_reference(ops):
    for m, ref in zip(ops["tt_masks"], ops["tt_out"]):
        got = O.trimap_transform(torch.from_numpy(m)).numpy()
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
    # empty class -> all-zero triple (utils/utils.py:32)
    assert ops["tt_out"][0][3:].max() == 0 and ops["tt_out"][2][:3].max() == 0


def test_exact_edt_is_exact():
    rng = np.random.Generator(np.random.PCG64(3))
    for _ in range(5):
        m = rng.uniform(0, 1, (19, 23)) < 0.9
        m[3, 4] = False
        d = O.exact_edt(m)
        d2 = O.exact_edt_sq_bruteforce(m)
        np.testing.assert_array_equal(np.round(d.astype(np.float64) ** 2).astype(np.int64), d2)
        np.testing.assert_array_equal(d, np.sqrt(d2.astype(np.float64)).astype(np.float32))


def test_fba_fusion_vs_reference(ops):
    a, F_, B_ = O.fba_fusion(torch.from_numpy(ops["ff_a"]), torch.from_numpy(ops["ff_img"]),
                             torch.from_numpy(ops["ff_F"]), torch.from_numpy(ops["ff_B"]))
    np.testing.assert_allclose(torch.cat([a, F_, B_], 1).numpy(), ops["ff_out"], rtol=0, atol=1e-6)


def test_pad_amounts_vs_reference(ops):
    for h, w, d, lw, uw, lh, uh in ops["pads"]:
        assert O.pad_amounts(int(h), int(w), int(d)) == (lw, uw, lh, uh)


@pytest.mark.parametrize("T", [1, 2, 5])
def test_memory_read_vs_reference(ops, T):
    got = O.memory_read(torch.from_numpy(ops["mem%d_mk" % T][0]), torch.from_numpy(ops["mem%d_mv" % T][0]),
                        torch.from_numpy(ops["mem%d_qk" % T][0]), torch.from_numpy(ops["mem%d_qv" % T][0]))
    np.testing.assert_allclose(got.numpy(), ops["mem%d_out" % T][0], rtol=1e-5, atol=1e-5)


def test_ws_conv_gn_vs_reference(ops):
    import torch.nn.functional as F
    w = O.standardise_weight(torch.from_numpy(ops["ws_w"]))
    y = F.conv2d(torch.from_numpy(ops["ws_x"]), w, torch.from_numpy(ops["ws_b"]), 1, 2, 2)
    y = F.group_norm(y, 32, torch.from_numpy(ops["ws_g"]), torch.from_numpy(ops["ws_be"]), 1e-5)
    np.testing.assert_allclose(y.numpy(), ops["ws_out"], rtol=1e-5, atol=1e-5)


def test_bank_policy_trajectory():
    """simulated frame ids resident in the bank, skip=5, max=5."""
    bank = []
    seen = {}
    for t in range(30):
        seen[t] = [b[2] for b in bank]
        bank = O.bank_update(bank, (None, None, t), t == 0, t % 5 == 0, 5)
    assert seen[1] == [0] and seen[2] == [0, 1] and seen[3] == [0, 2] and seen[6] == [0, 4, 5]
    assert seen[11] == [0, 4, 9, 10] and seen[16] == [0, 4, 9, 14, 15] and seen[21] == [0, 9, 14, 19, 20]
    assert seen[26] == [0, 14, 19, 24, 25]


@pytest.mark.parametrize("name", sorted(META.keys()))
def test_oracle_sequence_vs_reference(name, synth_sd):
    meta = META[name]
    gold = load_golden(name)
    orc = O.OtvmOracle(synth_sd, dilate_kernel=meta["dilate_kernel"])
    worst = 0.0
    for t, (a, fg, bg, tri_gt) in enumerate(clip_inputs(meta)):
        out = orc.frame(a, fg, bg, tri_gt=tri_gt, frame_id=t, **frame_flags(meta, t))
        assert len(orc.bank) == gold["bank"][t]
        da = np.abs(out[3][0, 0, 0].numpy() - gold["alpha"][t]).max()
        dt = np.abs(out[1][0, 0].numpy() - gold["trimap"][t]).max()
        worst = max(worst, da, dt)
        # bitwise on the machine that generated the fixtures; 1e-3 is the fp32 contract elsewhere
        assert da <= 1e-3 and dt <= 5e-3, (name, t, da, dt)
    np.testing.assert_array_equal(out[2][0, 0].numpy(), gold["tri_gt"])


def test_oracle_fullsize_sequence_vs_reference(synth_sd):
    """BASELINE configs[1] geometry: a 4-frame 832x480 clip matted by the REFERENCE itself (tests/golden/make_golden.py
    --c480; skip 3 / max 3, frames 2 and 3 read two memory slots) against the oracle.  Pins the oracle at a size where the
    memory read runs over 2 x 1560 positions and every pyramid level has its real aspect ratio -- the golden sequences above
    are <= 100 x 150."""
    import json
    meta = json.load(open(os.path.join(GOLDEN, "fullsize.json")))["c480_832x480_s3m3"]
    gold = load_golden("c480_832x480_s3m3")
    orc = O.OtvmOracle(synth_sd, dilate_kernel=meta["dilate_kernel"])
    for t, (a, fg, bg, tri_gt) in enumerate(clip_inputs(meta)):
        out = orc.frame(a, fg, bg, tri_gt=tri_gt, frame_id=t, **frame_flags(meta, t))
        assert len(orc.bank) == gold["bank"][t]
        da = float(np.abs(out[3][0, 0, 0].numpy() - gold["alpha"][t]).max())
        tri = out[1][0, 0].numpy()
        flips = int((tri.argmax(0) != gold["trimap_cls"][t]).sum())
        dtop = float(np.abs(tri.max(0) - gold["trimap_top"][t].astype(np.float32)).max())
        print("c480 t=%d alpha %.2e class flips %d top-prob %.2e (reference's own reorder noise: %.1e, %d flips)"
              % (t, da, flips, dtop, meta["reference_self_noise_alpha_maxabs"][t], meta["reference_self_noise_trimap_flips"][t]))
        # bitwise on the generating machine; elsewhere the fp32 contract (and the reference's own reorder noise flips 0-2 of
        # 399 360 output classes at this size)
        assert da <= 1e-3 and flips <= 8 and dtop <= 5e-3, (t, da, flips, dtop)


def test_oracle_1080p_frame_pair_vs_reference(synth_sd):
    """BASELINE configs[2] geometry (1920x1080, padded to 1088x1920 -- the only padded BASELINE size): the first frame and one
    propagated frame matted by the REFERENCE itself (tests/golden/make_golden.py --c1080, round 5) against the oracle.  Pins the
    oracle where the big tiles, the padding crop and the 8160-position memory read are real.  ~1 min of CPU."""
    import json
    meta = json.load(open(os.path.join(GOLDEN, "fullsize.json")))["c1080_1920x1080_s5m5"]
    gold = load_golden("c1080_1920x1080_s5m5")
    orc = O.OtvmOracle(synth_sd, dilate_kernel=meta["dilate_kernel"])
    for t, (a, fg, bg, tri_gt) in enumerate(clip_inputs(meta)):
        out = orc.frame(a, fg, bg, tri_gt=tri_gt, frame_id=t, **frame_flags(meta, t))
        assert len(orc.bank) == gold["bank"][t]
        alpha = out[3][0, 0, 0].numpy()
        flips = int((out[1][0, 0].numpy().argmax(0) != gold["trimap_cls"][t]).sum())
        if t == 0:
            d = float(np.abs(alpha.astype(np.float64).sum(1) - gold["alpha0_rowsum"]).max())
            print("c1080 t=0 row sums max-abs %.2e, class flips %d" % (d, flips))
            assert d <= 0.05 and flips <= 16, (d, flips)
        else:
            d = float(np.abs(alpha - gold["alpha1"]).max())
            print("c1080 t=1 alpha %.2e class flips %d (reference's own reorder noise: %.1e, %d flips)"
                  % (d, flips, meta["reference_self_noise_alpha_maxabs"][1], meta["reference_self_noise_trimap_flips"][1]))
            assert d <= 1e-3 and flips <= 16, (d, flips)


def test_oracle_1080p_steady_read_vs_reference(synth_sd):
    """Round 6: the same geometry through the STEADY memory read -- five 1920x1080 frames matted by the REFERENCE
    itself (tests/golden/make_golden.py --c1080-steady: memory every 3, at most 3 slots, so frames 2 / 3 / 4 read 2 / 2 / 3 slots
    of 8160 positions each: alpha/model.py:472-493, STM.py:148-159) against the oracle: per-row alpha sums on every frame, the
    full alpha of the last two.  ~5 min of CPU."""
    import json
    meta = json.load(open(os.path.join(GOLDEN, "fullsize.json")))["c1080_1920x1080_s3m3"]
    gold = load_golden("c1080_1920x1080_s3m3")
    keep = [int(t) for t in gold["alpha_frames"]]
    assert [int(b) for b in gold["bank"]] == [1, 2, 2, 3, 3]                # slots resident AFTER each frame: frame 4 reads 3
    orc = O.OtvmOracle(synth_sd, dilate_kernel=meta["dilate_kernel"])
    for t, (a, fg, bg, tri_gt) in enumerate(clip_inputs(meta)):
        out = orc.frame(a, fg, bg, tri_gt=tri_gt, frame_id=t, **frame_flags(meta, t))
        assert len(orc.bank) == gold["bank"][t]
        alpha = out[3][0, 0, 0].numpy()
        flips = int((out[1][0, 0].numpy().argmax(0) != gold["trimap_cls"][t]).sum())
        ds = float(np.abs(alpha.astype(np.float64).sum(1) - gold["alpha_rowsum"][t]).max())
        assert ds <= 0.05 and flips <= 16, (t, ds, flips)
        if t in keep:
            d = float(np.abs(alpha - gold["alpha"][keep.index(t)]).max())
            print("c1080 steady t=%d alpha %.2e row sums %.2e class flips %d (reference's own reorder noise: %.1e, %d flips)"
                  % (t, d, ds, flips, meta["reference_self_noise_alpha_maxabs"][t], meta["reference_self_noise_trimap_flips"][t]))
            assert d <= 1e-3, (t, d)


def test_oracle_stages_vs_reference(synth_sd):
    """Per-stage tensors of two consecutive frames (reference forward hooks) against the oracle's captures."""
    from otvm_amd.synth_data import synthetic_clip
    g = np.load(os.path.join(GOLDEN, "stages_64x64.npz"))
    H, W = int(g["H"]), int(g["W"])
    frames, tri = synthetic_clip(H, W, 2, int(g["clip_seed"]))
    orc = O.OtvmOracle(synth_sd, dilate_kernel=int(g["dk"]))
    for t in range(2):
        fg = torch.from_numpy(frames[t].astype(np.float32)).permute(2, 0, 1)[None, None].contiguous()
        a = torch.ones(1, 1, 1, H, W)
        cap = {}
        orc.frame(a, fg, fg.clone(), tri_gt=torch.from_numpy(tri)[None, None], first_frame=(t == 0), last_frame=False,
                  memorize=(t == 0), max_memory_num=5, frame_id=t, capture=cap)
        pairs = [("x11_%d" % t, cap["x11"]), ("l1_%d" % t, cap["feats"][2]), ("l4_%d" % t, cap["feats"][5]),
                 ("dec_hid_%d" % t, cap["dec_hid"]), ("dec_out_%d" % t, cap["dec_out"]), ("hid_%d" % t, cap["hid"]),
                 ("ref7_%d" % t, cap["ref7"]), ("tri_logits_%d" % t, cap["tri_logits"]),
                 ("key_m_%d" % t, cap["new_kv"][0][None]), ("val_m_%d" % t, cap["new_kv"][1][None])]
        if t == 1:
            pairs += [("r4_q", cap["r4"]), ("r3_q", cap["r3"]), ("r2_q", cap["r2"]), ("k4", cap["k4"]), ("v4", cap["v4"]),
                      ("m4", cap["m4"]), ("seg_logits", cap["seg_logits"])]
        for name, got in pairs:
            ref = g[name]
            d = float(np.abs(got.numpy() - ref).max())
            assert d <= 2e-4 * max(1.0, float(np.abs(ref).max())), (name, d)


def test_reference_self_noise_is_recorded():
    """The fixtures carry the reference's own fp32 reorder noise (oneDNN on/off) as the tolerance floor."""
    for name, meta in META.items():
        assert len(meta["reference_self_noise_alpha_maxabs"]) == meta["T"]


def test_metrics_oracle_vs_reference(ops):
    from oracle import metrics_oracle as M
    p, t, m = (torch.from_numpy(ops[k]) for k in ("met_pred", "met_target", "met_mask"))
    np.testing.assert_allclose(M.sad(p, t, m).numpy(), ops["met_sad"], rtol=1e-6)
    np.testing.assert_allclose(M.mse(p, t, m).numpy(), ops["met_mse"], rtol=1e-6)
    e, n = M.dtssd(p, t, m)
    np.testing.assert_allclose(e.numpy(), ops["met_dt_err"], rtol=1e-6)
    np.testing.assert_allclose(n.numpy(), ops["met_dt_num"], rtol=0)


def test_memory_read_query_blocks_equal_one_matrix():
    """oracle.memory_read in blocks of query columns (used when the [T*hw, hw] matrix would not fit: 4K banks) is the same
    function as the one-matrix evaluation of STM.py:144-163, and its float64 variant agrees with fp32 to fp32 rounding."""
    import torch
    from oracle.otvm_oracle import memory_read
    g = torch.Generator().manual_seed(5)
    T, h, w = 3, 9, 23
    keys, vals = torch.randn(128, T, h, w, generator=g), torch.randn(512, T, h, w, generator=g)
    qk, qv = torch.randn(128, h, w, generator=g), torch.randn(512, h, w, generator=g)
    one = memory_read(keys, vals, qk, qv)
    blk = memory_read(keys, vals, qk, qv, max_bytes=64 * T * h * w * 4)          # 64 query columns per block
    assert one.shape == blk.shape == (1024, h, w)
    assert float((one - blk).abs().max()) <= 2e-6
    f64 = memory_read(keys, vals, qk, qv, dtype=torch.float64, max_bytes=64 * T * h * w * 8)
    assert f64.dtype == torch.float32 and float((one - f64).abs().max()) <= 2e-5
    assert torch.equal(one[512:], qv) and torch.equal(f64[512:], qv)


@pytest.mark.parametrize("name", ["b2_s3_64x64", "b1_s4_64x96"])
def test_train_oracle_vs_reference_fixture(name):
    """oracle/train_oracle.py (training-mode forward + the FBA / trimap losses) against the outputs of the
    reference's own FullModel.forward (tests/golden/make_train_golden.py).  Batch 1 is bit-identical on the generating
    machine; with batch 2 the reference convolves both samples in one oneDNN call (another blocking than per sample)."""
    import numpy as np
    import torch
    from oracle.otvm_oracle import OtvmOracle
    from oracle.train_oracle import train_forward
    from otvm_amd.synth_data import train_batch
    from otvm_amd.synth_weights import synthetic_state_dict
    g = np.load(os.path.join(GOLDEN, "train_%s.npz" % name))
    B, S, H, W, seed = (int(g[k]) for k in ("B", "S", "H", "W", "seed"))
    a, fg, bg, tri = (torch.from_numpy(x) for x in train_batch(B, S, H, W, seed))
    r = train_forward(OtvmOracle(synthetic_state_dict(0)), a, fg, bg, tri)
    for k in ("loss1", "loss2", "loss3", "loss_trimap"):
        assert abs(float(r[k]) - float(g[k])) <= 2e-5 * max(1.0, abs(float(g[k]))), (k, float(r[k]), float(g[k]))
    for k in ("alphas", "comps", "Fs", "Bs", "preds_trimap"):
        assert float((r[k] - torch.from_numpy(g[k])).abs().max()) <= 2e-4, k
