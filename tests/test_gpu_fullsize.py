"""GPU (-m gpu): parity at BASELINE.json's full sizes.

The CPU oracle needs ~30 s per 1080p frame, so only a two-frame 1080p clip is compared directly; the rest are
size-independent properties that need no oracle: linearity of the convolution kernels on the real layer shapes,
slot-order invariance of the memory read, exactness properties of the distance transform, determinism and the
large-input schedule at 4K.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tests import gpu_util
    from otvm_amd import lib
    lib.load()
    return gpu_util


def _model(synth_sd, precision="f16x3", dk=12):
    from otvm_amd import helpers
    cfg = helpers.default_cfg()
    m = helpers.get_model_alpha(cfg, helpers.get_model_trimap(cfg, "Test", dk), "Test", dk)
    m.load_state_dict(synth_sd, strict=True)
    m.precision = precision
    return m.cuda().eval()


def _frame_vs_oracle(m, orc, a, fg, tg, t, kw, label, orc64=None):
    """One frame on the HIP path and on the oracle, with the tie-break protocol of tests/test_gpu_frame.py: when the
    HIP class map (the 3-class argmax feeding the distance transform, alpha/model.py:42) differs from the oracle's,
    every differing pixel must be a near-tie in the oracle (top-2 probability gap < 2e-3) and the oracle frame is re-run
    with the HIP tie-breaks -- so the 1e-3 alpha bound is ALWAYS asserted.  Returns (hip out, oracle out, alpha diff, ties).

    orc64 (optional): the same oracle evaluated in float64 with the same bank.  Then the frame is also compared with
    the EXACT value of the reference algorithm (bound 1e-3), and the bound against the fp32 oracle is widened by the
    fp32 oracle's own measured distance from that exact value on this frame -- two fp32 evaluations of the same
    algorithm differ by their summation orders, and with five memory slots the softmax over 40 800 memory positions
    amplifies a 1e-5 difference in the query key to ~1e-3 in the readout (tools/steady_diag.py: the exact-fp32 MFMA
    path and the f16x3 path are both ~1.0e-3 from the oneDNN oracle and 1.7e-4 from each other)."""
    out = m(a, fg, fg.clone(), tri_gt=tg, _frame_id=t, **kw)
    torch.cuda.synchronize()
    pl = m._engine.last_plan
    cls_h = pl.CLS.reshape(pl.Hp, pl.Wp).cpu().long()
    bank_before = list(orc.bank)
    cap = {}
    ref = orc.frame(a, fg, fg.clone(), tri_gt=tg, frame_id=t, capture=cap, **kw)
    ties = 0
    if not torch.equal(cls_h, cap["cls"]):
        diff = cls_h != cap["cls"]
        ties = int(diff.sum())
        top2 = torch.sort(cap["tri_in"][0], dim=0, descending=True)[0]
        gap = float((top2[0] - top2[1])[diff].max())
        assert gap < 2e-3, "%s: class map differs at a pixel that is not a near-tie (gap %g)" % (label, gap)
        orc.bank = bank_before
        cap = {}
        ref = orc.frame(a, fg, fg.clone(), tri_gt=tg, frame_id=t, capture=cap, class_override=cls_h, **kw)
    d = float((out[3].cpu() - ref[3]).abs().max())
    dt = float((out[1].cpu() - ref[1]).abs().max())
    print("%s frame %d: alpha max-abs %.3e, trimap max-abs %.3e, tie-breaks %d of %d" % (label, t, d, dt, ties, cls_h.numel()))
    slack = 0.0
    if orc64 is not None:
        ref64 = orc64.frame(a, fg, fg.clone(), tri_gt=tg, frame_id=t, class_override=cls_h if ties else None, **kw)
        d64 = float((out[3].cpu().double() - ref64[3]).abs().max())
        slack = float((ref[3].double() - ref64[3]).abs().max())
        print("%s frame %d: vs the float64 evaluation %.3e; the fp32 oracle's own distance from it %.3e" % (label, t, d64, slack))
        assert d64 <= 1e-3, "%s frame %d: alpha max-abs vs the exact (float64) evaluation %.3e" % (label, t, d64)
    assert d <= 1e-3 + slack, "%s frame %d: alpha max-abs %.3e (allowed 1e-3 + %.3e)" % (label, t, d, slack)
    assert dt <= 5e-3, "%s frame %d: trimap max-abs %.3e" % (label, t, dt)
    if "tri_in" in cap and not kw["first_frame"]:
        dp = float((pl.PROBS.reshape(1, 3, pl.Hp, pl.Wp).cpu() - cap["tri_in"]).abs().max())
        assert dp <= 1e-3, "%s frame %d: propagated trimap probabilities max-abs %.3e" % (label, t, dp)
    assert float((out[0].cpu() - ref[0]).abs().max()) <= 1e-6
    return out, ref, d, ties


def _clip_tensors(frames, tri, t, H, W):
    fg = torch.from_numpy(frames[t].astype(np.float32)).permute(2, 0, 1)[None, None].contiguous()
    return torch.ones(1, 1, 1, H, W), fg, torch.from_numpy(tri)[None, None]


def test_1080p_two_frames_vs_oracle(synth_sd):
    """BASELINE configs[2] geometry (1920x1080 -> padded 1088x1920): first frame + one propagated frame."""
    from oracle.otvm_oracle import OtvmOracle
    from otvm_amd.synth_data import synthetic_clip
    H, W, T = 1080, 1920, 2
    frames, tri = synthetic_clip(H, W, T, seed=21)
    m = _model(synth_sd)
    orc = OtvmOracle(synth_sd, dilate_kernel=12)
    for t in range(T):
        a, fg, tg = _clip_tensors(frames, tri, t, H, W)
        kw = dict(first_frame=(t == 0), last_frame=(t == T - 1), memorize=(t % 5 == 0), max_memory_num=5)
        out, ref, d, ties = _frame_vs_oracle(m, orc, a, fg, tg, t, kw, "1080p")
        assert out[3].shape == (1, 1, 1, H, W) and out[1].shape == (1, 1, 3, H, W)
        assert m.memories["frames"] == [b[2] for b in orc.bank]


@pytest.mark.parametrize("seed,full_f64", [(23, None), (41, False), (59, False)], ids=["seed23", "seed41", "seed59"])
def test_1080p_steady_state_frame_vs_oracle(synth_sd, seed, full_f64):
    """BASELINE configs[2] in its steady state (memory every 5, max 5 slots): the HIP path free-runs frames 0..20 -- all
    five slots filled, one eviction done (bank read by frame 21 = [0, 9, 14, 19, 20]) -- then frame 21 is
    compared with the oracle, whose bank is seeded from the device slots (a CPU frame is ~30 s at this size, so the
    oracle cannot free-run 21 of them).  Checks alpha (<= 1e-3 against the float64 evaluation of the reference algorithm,
    <= 1e-3 + the fp32 oracle's own rounding distance against the fp32 oracle), the propagated trimap, T_read = 5 and the
    bank after the frame's own update.

    Three clip seeds (the margin under 1e-3 was measured on one).  All three compare with the fp32 oracle whose
    Memory.forward alone is evaluated in float64; OTVM_TEST_FULL_F64=1 adds round 2's full float64 arbitration on seed 23
    (measured then: HIP 1.08e-4 from the float64 frame, the all-fp32 oracle 9.1e-4 from it).  The mixed evaluation:
    the fp32 oracle whose Memory.forward alone is evaluated in float64 (``read_dtype``: the
    softmax over 40 800 memory positions is the stage whose fp32 CPU evaluation is ~1e-3 from its exact value) and assert
    the plain 1e-3 bound with no slack."""
    from oracle.otvm_oracle import OtvmOracle
    from otvm_amd.synth_data import synthetic_clip
    import os
    if full_f64 is None:
        full_f64 = os.environ.get("OTVM_TEST_FULL_F64", "0") != "0"
    H, W, T, t_s = 1080, 1920, 24, 21
    frames, tri = synthetic_clip(H, W, t_s + 1, seed=seed)
    m = _model(synth_sd)
    flags = lambda t: dict(first_frame=(t == 0), last_frame=(t == T - 1), memorize=(t % 5 == 0), max_memory_num=5)
    for t in range(t_s):
        a, fg, tg = _clip_tensors(frames, tri, t, H, W)
        m(a.cuda(), fg.cuda(), fg.cuda(), tri_gt=tg.cuda(), _frame_id=t, **flags(t))
    eng = m._engine
    eng.flush()
    torch.cuda.synchronize()
    assert [s["frame"] for s in eng.bank] == [0, 9, 14, 19, 20]
    pl = eng.last_plan
    hw, h16, w16 = pl.hw, pl.Hp // 16, pl.Wp // 16
    orc = OtvmOracle(synth_sd, dilate_kernel=12, read_dtype=None if full_f64 else torch.float64)
    orc.bank = [(s["k"].t.reshape(hw, 128).t().reshape(128, h16, w16).cpu().contiguous(),
                 s["v"].t.reshape(hw, 512).t().reshape(512, h16, w16).cpu().contiguous(), s["frame"]) for s in eng.bank]
    bank0 = list(orc.bank)
    orc64 = None
    if full_f64:
        orc64 = OtvmOracle(synth_sd, dilate_kernel=12, dtype=torch.float64)
        orc64.bank = [(k.double(), v.double(), f) for k, v, f in orc.bank]
    a, fg, tg = _clip_tensors(frames, tri, t_s, H, W)
    out, ref, d, ties = _frame_vs_oracle(m, orc, a, fg, tg, t_s, flags(t_s),
                                         "1080p steady state (T_read=5, seed %d%s)" % (seed, "" if full_f64 else ", float64 memory read in the oracle"),
                                         orc64=orc64)
    print("1080p steady state seed %d: margin under the 1e-3 bound %.3e" % (seed, 1e-3 - d))
    if not full_f64:
        # The evaluation north_star names literally: the ALL-fp32 CPU forward.  Its Memory.forward -- a softmax over 40 800
        # positions in fp32 oneDNN arithmetic -- is itself ~9e-4 from the exact value, so the bound against it is
        # the contract value plus that evaluation's own measured distance from the float64-read oracle on THIS frame (both
        # terms computed here, every seed; VERDICT r4: the number was printed on one seed and nothing failed if it drifted).
        o32 = OtvmOracle(synth_sd, dilate_kernel=12)
        o32.bank = list(bank0)
        cls_h = pl.CLS.reshape(pl.Hp, pl.Wp).cpu().long()
        r32 = o32.frame(a, fg, fg.clone(), tri_gt=tg, frame_id=t_s, class_override=cls_h if ties else None, **flags(t_s))
        d32 = float((out[3].cpu() - r32[3]).abs().max())
        own32 = float((r32[3] - ref[3]).abs().max())
        print("1080p steady state seed %d: distance to the all-fp32 CPU oracle %.3e (that oracle's own distance from the "
              "float64-read oracle %.3e; HIP vs the float64-read oracle %.3e)" % (seed, d32, own32, d))
        assert d32 <= 1e-3 + own32, "seed %d: alpha vs the all-fp32 CPU forward %.3e (allowed 1e-3 + %.3e)" % (seed, d32, own32)
    assert m.memories["frames"] == [b[2] for b in orc.bank] == [0, 9, 14, 19, 21]


def test_480p_sequence_vs_oracle(synth_sd):
    """BASELINE configs[1] geometry (832x480, no padding): first frame, a propagated frame with the memory read and a
    second memorised frame -- the size at which the split-K route and the small-map tiles carry most layers."""
    from oracle.otvm_oracle import OtvmOracle
    from otvm_amd.synth_data import synthetic_clip
    H, W, T = 480, 832, 4
    frames, tri = synthetic_clip(H, W, T, seed=22)
    m = _model(synth_sd)
    orc = OtvmOracle(synth_sd, dilate_kernel=12)
    for t in range(T):
        a, fg, tg = _clip_tensors(frames, tri, t, H, W)
        kw = dict(first_frame=(t == 0), last_frame=(t == T - 1), memorize=(t % 2 == 0), max_memory_num=5)
        _frame_vs_oracle(m, orc, a, fg, tg, t, kw, "480p")
        assert m.memories["frames"] == [b[2] for b in orc.bank]


def test_480p_reference_generated_fixture(synth_sd):
    """The same geometry against a fixture produced by the REFERENCE itself (tests/golden/seq_c480_832x480_s3m3.npz, made by
    tests/golden/make_golden.py --c480 from the imported reference; skip 3 / max 3: frames 2 and 3 read two slots).  The
    HIP path free-runs the clip; every frame is compared with the oracle under the tie-break protocol AND -- as long as no
    tie-break has happened -- directly with the reference's alpha at the contract value."""
    import json
    import os
    from oracle.otvm_oracle import OtvmOracle
    from tests.common import GOLDEN, clip_inputs, frame_flags, load_golden
    meta = json.load(open(os.path.join(GOLDEN, "fullsize.json")))["c480_832x480_s3m3"]
    gold = load_golden("c480_832x480_s3m3")
    m = _model(synth_sd)
    orc = OtvmOracle(synth_sd, dilate_kernel=meta["dilate_kernel"])
    total_ties = 0
    for t, (a, fg, bg, tg) in enumerate(clip_inputs(meta)):
        out, ref, d, ties = _frame_vs_oracle(m, orc, a, fg, tg, t, frame_flags(meta, t), "480p reference fixture")
        total_ties += ties
        assert len(m.memories["frames"]) == gold["bank"][t] and m.memories["frames"] == [b[2] for b in orc.bank]
        do = float(np.abs(ref[3][0, 0, 0].numpy() - gold["alpha"][t]).max())
        dg = float(np.abs(out[3][0, 0, 0].cpu().numpy() - gold["alpha"][t]).max())
        print("480p reference fixture frame %d: HIP vs the reference's alpha %.3e; this box's oracle vs it %.3e; tie-breaks so far %d"
              % (t, dg, do, total_ties))
        if total_ties == 0:
            assert dg <= 1e-3, "frame %d: alpha max-abs vs the reference-generated fixture %.3e" % (t, dg)


def test_1080p_reference_generated_fixture(synth_sd):
    """BASELINE configs[2] geometry (1920x1080 -> padded 1088x1920, the only padded BASELINE size) against a fixture produced by
    the REFERENCE itself (tests/golden/seq_c1080_1920x1080_s5m5.npz: tests/golden/make_golden.py --c1080 from the imported
    reference; round 5): the first frame (anchored by the reference's per-row alpha sums) and one propagated
    frame with the memory read, whose alpha is compared at the contract value -- directly, not through the oracle."""
    import json
    import os
    from tests.common import GOLDEN, clip_inputs, frame_flags, load_golden
    meta = json.load(open(os.path.join(GOLDEN, "fullsize.json")))["c1080_1920x1080_s5m5"]
    gold = load_golden("c1080_1920x1080_s5m5")
    m = _model(synth_sd)
    outs, nflips = [], []
    for t, (a, fg, bg, tg) in enumerate(clip_inputs(meta)):
        out = m(a, fg, fg.clone(), tri_gt=tg, _frame_id=t, **frame_flags(meta, t))
        torch.cuda.synchronize()
        outs.append(out[3][0, 0, 0].cpu().numpy())
        cls = out[1][0, 0].argmax(0).cpu().numpy()               # class map of the OUTPUT trimap, as the fixture stores it
        flips = int((cls != gold["trimap_cls"][t]).sum())
        nflips.append(flips)
        print("1080p reference fixture frame %d: class map differs from the reference's at %d pixels (reference vs itself with "
              "another summation order: %d)" % (t, flips, meta["reference_self_noise_trimap_flips"][t]))
    # frame 0: per-row sums (1920 values each): a 1e-3 max-abs error on every pixel of a row would move its sum by 1.9
    d0 = float(np.abs(outs[0].astype(np.float64).sum(1) - gold["alpha0_rowsum"]).max())
    d1 = float(np.abs(outs[1] - gold["alpha1"]).max())
    print("1080p reference fixture: frame 0 row sums max-abs %.3e; frame 1 alpha max-abs vs the reference %.3e (reference self-noise "
          "%.1e)" % (d0, d1, meta["reference_self_noise_alpha_maxabs"][1]))
    assert d0 <= 0.25, d0
    # (the output trimap's class map flips at a handful of near-tie pixels under ANY other fp32 summation order -- the reference
    #  against itself: 3 / 4 pixels -- so it is bounded, not required to be equal)
    assert max(nflips) <= 64, nflips
    assert d1 <= 1e-3, "frame 1: alpha max-abs vs the reference-generated fixture %.3e" % d1
    assert m.memories["frames"] == [0]                         # (the last frame does not memorise, alpha/model.py:461)


def test_1080p_reference_generated_steady_fixture(synth_sd):
    """Round 6: the HIP path against the REFERENCE's own output through the steady memory read at BASELINE
    configs[2]'s geometry -- five 1920x1080 frames, memory every 3, at most 3 slots (frames 2 / 3 / 4 read 2 / 2 / 3 slots of 8160
    positions: alpha/model.py:472-493, STM.py:148-159), tests/golden/seq_c1080_1920x1080_s3m3.npz.  Alpha at the contract value
    (1e-3, directly against the reference -- not through the oracle) on the stored frames while no tie-break of the propagated
    trimap's argmax has occurred (a flipped class is a different, equally valid, trajectory: the frames behind it are compared
    through the oracle in the other tests); per-row sums on every frame."""
    import json
    import os
    from tests.common import GOLDEN, clip_inputs, frame_flags, load_golden
    meta = json.load(open(os.path.join(GOLDEN, "fullsize.json")))["c1080_1920x1080_s3m3"]
    gold = load_golden("c1080_1920x1080_s3m3")
    keep = [int(t) for t in gold["alpha_frames"]]
    m = _model(synth_sd)
    eng_model = m.module if hasattr(m, "module") else m
    flips_so_far, reads = 0, []
    for t, (a, fg, bg, tg) in enumerate(clip_inputs(meta)):
        out = m(a, fg, fg.clone(), tri_gt=tg, _frame_id=t, **frame_flags(meta, t))
        torch.cuda.synchronize()
        reads.append(eng_model._engine.last_T_read)
        alpha = out[3][0, 0, 0].cpu().numpy()
        cls = out[1][0, 0].argmax(0).cpu().numpy()
        flips = int((cls != gold["trimap_cls"][t]).sum())
        ds = float(np.abs(alpha.astype(np.float64).sum(1) - gold["alpha_rowsum"][t]).max())
        msg = "1080p steady reference fixture frame %d (T_read %d): row sums max-abs %.3e, class map differs at %d pixels (reference " \
              "vs itself: %d)" % (t, reads[-1], ds, flips, meta["reference_self_noise_trimap_flips"][t])
        if t in keep:
            d = float(np.abs(alpha - gold["alpha"][keep.index(t)]).max())
            msg += "; alpha max-abs vs the reference %.3e (reference self-noise %.1e)" % (d, meta["reference_self_noise_alpha_maxabs"][t])
            if flips_so_far <= 64:
                assert d <= 1e-3, "frame %d: alpha max-abs vs the reference-generated fixture %.3e" % (t, d)
        print(msg)
        assert ds <= 0.25 and flips <= 64, (t, ds, flips)
        flips_so_far += flips
    assert reads == [0, 1, 2, 2, 3], reads                      # slots read by each frame's memory read (bank after: 1 2 2 3 3)


def test_demo_dove_layout_1080p_clip_through_eval_cli(tmp_path, synth_sd):
    """BASELINE configs[0]: a clip laid out exactly like the reference's demo/dove (11 JPEG frames of 1920x1080 under
    <root>/dove/frames/00000.jpg.., ONE grayscale trimap <root>/dove/trimap/00000.png with the levels {0, 128, 254},
    dataset.py:887-893) through the eval.py-shaped command line with the reference's default schedule (memory every 10,
    max 5).  The reference's own images do not travel, so the JPEGs are generated here at the same size and coding.
    Checks: PNGs written == direct run_video_matte on the decoded frames (the CLI's decode-ahead / async-write pipeline
    changes nothing), the bank trajectory of SURVEY.md 3.3, frames 0 and 1 against the CPU oracle, --trimap wide."""
    import os
    from PIL import Image
    from oracle.otvm_oracle import OtvmOracle
    from otvm_amd import eval_cli
    from otvm_amd.datasets import Demo_Test, load_sequence
    from otvm_amd.synth_data import synthetic_clip
    from otvm_amd.video import run_video_matte
    H, W, T = 1080, 1920, 11
    frames_bgr, tri = synthetic_clip(H, W, T, seed=25)
    root = os.path.join(str(tmp_path), "demo")
    os.makedirs(os.path.join(root, "dove", "frames")); os.makedirs(os.path.join(root, "dove", "trimap"))
    for t in range(T):
        Image.fromarray(frames_bgr[t][..., ::-1].copy()).save(os.path.join(root, "dove", "frames", "%05d.jpg" % t), quality=92)
    g = (np.asarray(tri)[1] * 128 + np.asarray(tri)[2] * 254).astype(np.uint8)
    Image.fromarray(g).save(os.path.join(root, "dove", "trimap", "00000.png"))
    out_dir = os.path.join(str(tmp_path), "demo_results")
    s = eval_cli.main(["--demo", "--data", root, "--out", out_dir, "--synthetic-weights"])
    assert s["frames"] == T and s["sequences"] == [0]
    res_cli = s["outputs"][0]
    # bank read by frame t (ids resident after frame t-1's update): [0], [0,1], [0,2] ... [0,9]; the last frame does
    # not memorize (alpha/model.py:461)
    assert res_cli["bank_frames"] == [[0]] + [[0, t] for t in range(1, T - 1)] + [[0, T - 2]]
    d = load_sequence(next(iter(Demo_Test(root))))
    assert np.array_equal(d["trimap"], np.asarray(tri))                      # {0,128,254} -> the one-hot we started from
    m = _model(synth_sd)
    direct = run_video_matte(m, d["frames"], trimap=d["trimap"], skip=10, max_num=5)
    pred = os.path.join(out_dir, "alpha", "test", "s4_OTVM", "pred", "dove")
    for t in range(T):
        png = np.asarray(Image.open(os.path.join(pred, "%05d.png" % t)))
        assert png.shape == (H, W) and np.array_equal(png, direct["alpha_u8"][t].numpy())
    assert torch.equal(res_cli["alpha"].cpu(), direct["alpha"])
    # frames 0 and 1 of the decoded clip against the oracle
    orc = OtvmOracle(synth_sd, dilate_kernel=12)
    for t in (0, 1):
        fg = torch.from_numpy(d["frames"][t].astype(np.float32)).permute(2, 0, 1)[None, None].contiguous()
        ref = orc.frame(torch.ones(1, 1, 1, H, W), fg, fg.clone(), tri_gt=torch.from_numpy(d["trimap"])[None, None],
                        frame_id=t, first_frame=(t == 0), last_frame=False, memorize=(t % 10 == 0), max_memory_num=5)
        dd = float((direct["alpha"][t] - ref[3][0, 0, 0]).abs().max())
        print("dove-layout frame %d: alpha max-abs vs oracle %.3e" % (t, dd))
        assert dd <= 1e-3
    # --trimap wide (dilate kernel 20, eval.py:71-72) only changes the V108 flow; on the demo flow it must be inert
    s2 = eval_cli.main(["--demo", "--data", root, "--out", out_dir + "_wide", "--synthetic-weights", "--trimap", "wide",
                        "--max-frames", "2", "--sync-io"])
    assert torch.equal(s2["outputs"][0]["alpha"].cpu(), direct["alpha"][:2])


LAYERS = [  # (Cin, Cout, k, dil, H, W): real 1080p layer geometries
    (64, 64, 3, 1, 1088, 1920),      # refinement 64->64, full resolution (patch kernel)
    (80, 32, 3, 1, 1088, 1920),      # conv_up4.0 (16x32-pixel patch blocks)
    (256, 256, 3, 1, 272, 480),      # STM decoder RF2 (wide patch kernel)
    (512, 512, 3, 4, 136, 240),      # FBA layer4 dilated conv
    (1024, 256, 1, 1, 136, 240),     # FBA layer3 1x1
    (3072, 256, 3, 1, 136, 240),     # conv_up1.0 (K = 27648)
]


@pytest.mark.parametrize("shape", LAYERS, ids=lambda s: "c%d_%d_k%d_d%d_%dx%d" % s)
def test_conv_linearity_at_full_size(G, shape):
    """conv(a*x + b*y) == a*conv(x) + b*conv(y) on the real layer shapes (fp32-class tolerance)."""
    Cin, Cout, k, dil, H, W = shape
    g = torch.Generator(device="cpu").manual_seed(5)
    dev = G.DEV
    x = torch.randn(H * W * Cin, generator=g).to(dev)
    y = torch.randn(H * W * Cin, generator=g).to(dev)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    cw = G.pack_weight(w)
    pad = dil * (k - 1) // 2
    from otvm_amd.engine import Act

    def run(t):
        out = Act(torch.empty(H * W * Cout, device=dev), H, W, Cout)
        G.conv2d(Act(t, H, W, Cin), cw, out, pad=pad, dil=dil, precision=1)
        return out.t
    a_, b_ = 0.75, -1.5
    lhs = run(a_ * x + b_ * y)
    rhs = a_ * run(x) + b_ * run(y)
    scale = float(rhs.abs().max())
    assert torch.isfinite(lhs).all()
    assert float((lhs - rhs).abs().max()) <= 2e-5 * scale
    # spot-check 64 random output pixels against an fp64 direct evaluation
    idx = torch.randint(0, H * W, (64,), generator=g)
    xv = x.reshape(H, W, Cin).cpu().double()
    wv = w.double()
    o = run(x).reshape(H, W, Cout).cpu().double()
    for i in idx.tolist():
        py, px = divmod(i, W)
        acc = torch.zeros(Cout, dtype=torch.float64)
        for ky in range(k):
            for kx in range(k):
                iy, ix = py + (ky - k // 2) * dil, px + (kx - k // 2) * dil
                if 0 <= iy < H and 0 <= ix < W:
                    acc += wv[:, :, ky, kx] @ xv[iy, ix]
        assert float((o[py, px] - acc).abs().max()) <= 2e-5 * max(1.0, float(acc.abs().max()))


def test_memory_read_slot_order_invariance_1080p(G):
    """The readout sums over all memory positions: permuting the slots must not change it."""
    from otvm_amd import lib as L
    lib = L.load()
    hw, T = 68 * 120, 5
    g = torch.Generator().manual_seed(9)
    keys = [(torch.randn(hw, 128, generator=g) * 0.8).to(G.DEV) for _ in range(T)]
    vals = [torch.randn(hw, 512, generator=g).to(G.DEV) for _ in range(T)]
    q = (torch.randn(hw, 128, generator=g) * 0.8).to(G.DEV)
    slots = []
    for t in range(T):
        sl = torch.zeros(int(lib.otvm_bank_slot_bytes_f16x3(hw)), dtype=torch.uint8, device=G.DEV)
        L.check(lib.otvm_bank_pack_f16x3(keys[t].data_ptr(), vals[t].data_ptr(), hw, sl.data_ptr(), G.stream()))
        slots.append(sl)
    ws = torch.empty(int(lib.otvm_memory_read_ws_bytes(hw, T)), dtype=torch.uint8, device=G.DEV)
    outs = []
    for perm in ([0, 1, 2, 3, 4], [4, 2, 0, 3, 1]):
        out = torch.empty(hw, 512, device=G.DEV)
        sp = (C.c_void_p * T)(*[slots[i].data_ptr() for i in perm])
        L.check(lib.otvm_memory_read_f16x3(q.data_ptr(), 128, sp, T, hw, out.data_ptr(), 512, ws.data_ptr(), G.stream()))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.isfinite(outs[0]).all()
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-5 * float(outs[0].abs().max())
    # and against a direct fp64 softmax readout for a handful of queries
    K = torch.cat(keys).double().cpu()
    V = torch.cat(vals).double().cpu()
    for qi in (0, 4079, 8159):
        p = torch.softmax(K @ q[qi].double().cpu() / math.sqrt(128.0), 0)
        ref = p @ V
        assert float((outs[0][qi].double().cpu() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_distance_encoding_properties_1080p(G):
    """Exact-EDT properties at 1088x1920 without an oracle: value 1 exactly on the class, monotone in the three
    sigmas, and agreement with the brute-force definition on sampled pixels."""
    from tests.test_gpu_kernels import _encode
    H, W = 1088, 1920
    g = torch.Generator().manual_seed(3)
    logits = F.interpolate(torch.randn(1, 3, 17, 30, generator=g) * 4, size=(H, W), mode="bilinear")[0]
    probs = torch.softmax(logits, 0)
    enc, cls, _ = _encode(G, probs)
    for k, target in ((0, 0), (1, 2)):
        on = cls == target
        assert bool(on.any())
        e = enc[3 * k:3 * k + 3]
        assert torch.equal(e[:, on], torch.ones_like(e[:, on]))                  # d = 0 on the class itself
        assert bool((e[0] <= e[1] + 1e-7).all()) and bool((e[1] <= e[2] + 1e-7).all())
        ys, xs = torch.nonzero(on, as_tuple=True)
        pts = torch.stack([ys, xs], 1).double()
        for (py, px) in [(5, 7), (544, 960), (1087, 1919), (300, 1500), (900, 100)]:
            d2 = float(((pts - torch.tensor([py, px], dtype=torch.float64)) ** 2).sum(1).min())
            want = math.exp(-(np.float32(math.sqrt(d2)) ** 2) / (2 * (0.08 * 320) ** 2))
            assert abs(float(e[1, py, px]) - want) <= 2e-6


def test_4k_growing_bank_frame_vs_oracle(synth_sd):
    """BASELINE configs[4], the unbounded-bank stress variant (memory every frame, no eviction) at 3840x2160: the HIP path
    free-runs frames 0..2 (three memorised 2176x3840 slots, 97 920 memory positions), the oracle's bank is seeded from the
    device slots and frame 3 is compared (alpha <= 1e-3, tie-break protocol, bank ids).

    Round 3: runs in the default suite.  The oracle's Memory.forward evaluates the [97 920, 32 640] affinity matrix in
    blocks of query columns (oracle.memory_read: same arithmetic per column, < 1 GB live instead of 12.8 GB x several
    copies) and in float64 (``read_dtype``): with ~1e5 memory positions the fp32 bmm + softmax of the CPU evaluation is
    itself 1.6e-3 away from the exact value (profiles/r02_4k_growing_bank_vs_oracle.log), every other stage stays fp32.
    The bound against it is the plain 1e-3.  OTVM_TEST_4K_ORACLE=1 adds the full-float64 frame (~8 minutes of CPU)."""
    import os
    from oracle.otvm_oracle import OtvmOracle
    from otvm_amd.synth_data import synthetic_clip
    # (round 5: frame 3 is the clip's LAST frame -- no memorize on either side: the compared quantity is the frame that READS the
    #  three slots; the CPU oracle's Encoder_M pass at 4K was a fifth of this test's 217 s.  Memorize at 4K: the frames before.)
    H, W, T, t_s = 2160, 3840, 4, 3
    frames, tri = synthetic_clip(H, W, t_s + 1, seed=29)
    m = _model(synth_sd)
    flags = lambda t: dict(first_frame=(t == 0), last_frame=(t == T - 1), memorize=True, max_memory_num=64)
    for t in range(t_s):
        a, fg, tg = _clip_tensors(frames, tri, t, H, W)
        m(a.cuda(), fg.cuda(), fg.cuda(), tri_gt=tg.cuda(), _frame_id=t, **flags(t))
    eng = m._engine
    eng.flush()
    torch.cuda.synchronize()
    assert [s["frame"] for s in eng.bank] == [0, 1, 2]
    pl = eng.last_plan
    hw, h16, w16 = pl.hw, pl.Hp // 16, pl.Wp // 16
    assert (pl.Hp, pl.Wp, hw) == (2176, 3840, 32640)
    orc = OtvmOracle(synth_sd, dilate_kernel=12, read_dtype=torch.float64)
    orc.bank = [(s["k"].t.reshape(hw, 128).t().reshape(128, h16, w16).cpu().contiguous(),
                 s["v"].t.reshape(hw, 512).t().reshape(512, h16, w16).cpu().contiguous(), s["frame"]) for s in eng.bank]
    orc64 = None
    if os.environ.get("OTVM_TEST_4K_ORACLE", "0") != "0":
        orc64 = OtvmOracle(synth_sd, dilate_kernel=12, dtype=torch.float64)
        orc64.bank = [(k.double(), v.double(), f) for k, v, f in orc.bank]
    a, fg, tg = _clip_tensors(frames, tri, t_s, H, W)
    _frame_vs_oracle(m, orc, a, fg, tg, t_s, flags(t_s), "4K growing bank (T_read=3, float64 memory read in the oracle)", orc64=orc64)
    assert m.memories["frames"] == [b[2] for b in orc.bank] == [0, 1, 2]


def test_4k_large_input_runs_and_is_deterministic(synth_sd):
    """BASELINE configs[4] geometry (3840x2160 -> 2176x3840): large-input schedule (eval.py:184-187), finite output,
    bit-identical on a re-run."""
    from otvm_amd.synth_data import disc_trimap
    from otvm_amd.video import memory_schedule, run_video_matte
    H, W, T = 2160, 3840, 3
    assert memory_schedule(0, H, W, 10, 5) == (True, 2, True)
    m = _model(synth_sd)
    g = torch.Generator().manual_seed(1)
    lo = torch.rand(T, 3, H // 16, W // 16, generator=g)
    frames = (F.interpolate(lo, size=(H, W), mode="bilinear") * 255).floor().permute(0, 2, 3, 1).contiguous()
    tri = disc_trimap(H, W)
    r1 = run_video_matte(m, frames, trimap=tri, skip=10, max_num=5, keep_on_device=True)
    r2 = run_video_matte(m, frames, trimap=tri, skip=10, max_num=5, keep_on_device=True)
    assert torch.isfinite(r1["alpha"]).all() and r1["alpha"].shape == (T, H, W)
    assert float(r1["alpha"].min()) >= 0.0 and float(r1["alpha"].max()) <= 1.0
    assert torch.equal(r1["alpha"], r2["alpha"]) and torch.equal(r1["alpha_u8"], r2["alpha_u8"])
    assert m.memories["frames"] == [0, 1]


def test_1080p_free_running_clip_is_bit_reproducible(synth_sd):
    """Stream hazards at BASELINE size: a 1080p clip matted with the host running ahead of the device (query encoder of the
    next frame, resident-slot memory read, decoder skip branches and preprocess on side streams, otvm_amd/engine.py) must
    reproduce, bit for bit, the same clip matted with a device synchronisation after every frame -- including bench.py's
    cross-check flow (engine.flush() + sync before a frame, which then reads every slot on the side stream).
    tools/race_stress.py runs the same check for hundreds of repetitions."""
    from otvm_amd.synth_data import synthetic_clip
    H, W, T, t_flush = 1080, 1920, 8, 6
    frames, tri = synthetic_clip(H, W, T, seed=37)
    dev = torch.device("cuda:0")
    fr = [torch.from_numpy(frames[t]).to(dev) for t in range(T)]            # uint8 [H,W,3], resident
    tri_d = torch.from_numpy(tri)[None, None].to(dev)
    ones = torch.ones(1, 1, 1, H, W, device=dev)
    m = _model(synth_sd)
    torch.cuda.synchronize()

    def matte(ready, sync, flush):
        outs = []
        for t in range(T):
            if flush and t == t_flush:
                m._engine.flush()
                torch.cuda.synchronize()
            o = m(ones, fr[t], fr[t], tri_gt=tri_d, first_frame=(t == 0), last_frame=(t == T - 1), memorize=(t % 5 == 0),
                  max_memory_num=5, _inputs_ready=ready)
            outs.append(o[3])
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return torch.stack(outs)
    # (a flush changes how the bank is grouped into partial launches -- all slots in one group instead of resident + new --
    # and with it the fp32 summation order of the read: each flow is compared with its own synchronised run)
    ref = {False: matte(None, True, False), True: matte(None, True, True)}
    assert float((ref[True] - ref[False]).abs().max()) <= 1e-4
    for rep in range(3):
        for ready, flush in ((True, False), (None, False), (True, True)):
            assert torch.equal(matte(ready, False, flush), ref[flush]), (rep, ready, flush)
