"""GPU (-m gpu): every HIP kernel of libotvm_hip.so against the same op evaluated on the CPU
(torch fp32 ops / the oracle's functions), through the C ABI."""
import ctypes as C
import math
import os

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


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CONV_CASES = [
    # Cin, Cout, k, stride, pad, dil, H, W, bias, act, in_relu, residual
    (64, 64, 1, 1, 0, 1, 24, 40, False, 0, 0, False),
    (64, 256, 1, 1, 0, 1, 24, 40, True, 1, 0, True),
    (256, 512, 1, 2, 0, 1, 24, 40, False, 0, 0, False),
    (128, 128, 3, 2, 1, 1, 24, 40, False, 0, 0, False),
    (256, 256, 3, 1, 2, 2, 17, 30, False, 0, 0, False),
    (512, 512, 3, 1, 4, 4, 17, 30, False, 0, 0, False),
    (11, 64, 7, 2, 3, 1, 64, 96, False, 0, 0, False),
    (22, 64, 7, 2, 3, 1, 32, 64, True, 1, 0, False),
    (72, 32, 3, 1, 1, 1, 32, 48, True, 2, 0, False),
    (32, 16, 3, 1, 1, 1, 32, 48, True, 2, 0, False),
    (73, 64, 3, 1, 1, 1, 32, 48, True, 0, 0, False),
    (256, 3, 3, 1, 1, 1, 16, 24, True, 0, 1, False),
    (256, 256, 3, 1, 1, 1, 16, 24, True, 0, 1, True),
    (1024, 128, 3, 1, 1, 1, 6, 8, True, 0, 0, False),
    (320, 64, 3, 1, 1, 1, 40, 56, True, 0, 0, False),
    (2048, 256, 1, 1, 0, 1, 6, 6, True, 0, 0, False),
    (64, 64, 3, 1, 1, 1, 130, 258, False, 0, 0, False),     # > 256*128 pixels -> the 128x64 tile
    (512, 2048, 1, 1, 0, 1, 40, 70, False, 0, 0, False),    # many tiles -> the 128x128 tile
    (96, 256, 1, 1, 0, 1, 352, 353, True, 1, 1, True),      # large M, Cout 256: the two-stage 256-row tiles
    (40, 128, 3, 2, 1, 1, 704, 705, True, 2, 0, False),     # large M, generic K decode (Cin % 32 != 0), stride 2
    (64, 384, 1, 1, 0, 1, 351, 353, False, 0, 0, False),    # ragged M and N tiles on the big tiles
    (16, 32, 3, 1, 1, 1, 1030, 1031, True, 2, 0, False),    # >= 2^20 pixels, <= 32 filters: 16x32-pixel patch blocks, ragged edges
    (32, 32, 3, 1, 4, 4, 17, 12, False, 1, 0, False),       # dilated, <= 32 filters on the 64-channel patch tile (second tile: no weights)
]


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f16x3"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "c%d_%d_k%d_s%d_d%d_%dx%d" % (c[0], c[1], c[2], c[3], c[5], c[6], c[7]))
def test_conv2d(G, case, prec):
    Cin, Cout, k, stride, pad, dil, H, W, use_bias, act, in_relu, use_res = case
    x = rnd(1, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=1.0 / math.sqrt(Cin * k * k))
    b = rnd(Cout, seed=3) if use_bias else None
    xin = F.relu(x) if in_relu else x
    ref = F.conv2d(xin, w, b, stride, pad, dil)
    res = rnd(*ref.shape, seed=4) if use_res else None
    if use_res:
        ref = ref + res
    ref = F.relu(ref) if act == 1 else (F.leaky_relu(ref, 0.01) if act == 2 else ref)
    cw = G.pack_weight(w)
    xa = G.to_act(x, ld=cw.I_pad + 4, off=4)                 # read a channel slice of a wider buffer
    assert xa.C == cw.I_pad
    out = G.empty_act(ref.shape[2], ref.shape[3], max(4, (Cout + 3) // 4 * 4), ld=Cout + 8 - Cout % 4, off=4)
    ra = G.to_act(res) if use_res else None
    G.conv2d(xa, cw, out, None if b is None else b.to(G.DEV), stride, pad, dil, act, in_relu, ra, precision=prec)
    got = G.from_act(out, Cout)
    assert torch.isfinite(got).all()
    # same bound for the exact-fp32 MFMA and the split-fp16 (f16x3) path: both are fp32-class
    assert G.maxdiff(got, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f16x3"])
@pytest.mark.parametrize("Cin,Cout,H,W", [(64, 64, 50, 70), (32, 128, 33, 47), (128, 256, 20, 31), (64, 1024, 9, 13), (256, 2048, 6, 7)])
def test_conv_fused_groupnorm_stats(G, prec, Cin, Cout, H, W):
    """The conv epilogue accumulates the GroupNorm(32) sums of its output (replaces the separate stats pass)."""
    x = rnd(1, Cin, H, W, seed=70)
    w = rnd(Cout, Cin, 3, 3, seed=71, scale=1.0 / math.sqrt(Cin * 9))
    b = rnd(Cout, seed=72)
    ref = F.conv2d(x, w, b, 1, 1).double()
    g = ref.reshape(32, Cout // 32, H * W)
    want = torch.stack([g.sum((1, 2)), (g * g).sum((1, 2))], 1).flatten()
    stats = torch.zeros(64, dtype=torch.float64, device=G.DEV)
    out = G.empty_act(H, W, Cout)
    G.conv2d(G.to_act(x), G.pack_weight(w), out, b.to(G.DEV), pad=1, precision=prec, gn_stats=stats)
    got = stats.cpu()
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("Cin,Cout,k,stride,H,W,use_res,act,gn", [
    (2048, 256, 1, 1, 30, 52, True, 1, False),      # 480p OS16, 64 chunks: 26 tiles of 128x128 -> 4-way split
    (512, 64, 3, 1, 17, 23, False, 2, False),       # narrow output (128x64 tiles), 3x3 with the fragment weights withheld (see below)
    (256, 256, 3, 2, 24, 32, False, 0, True),       # strided 3x3 + fused GroupNorm sums (reduced after the finish pass)
    (2048, 128, 1, 1, 15, 26, True, 0, False),      # ragged M, K = 64 chunks
])
def test_conv_split_k(G, Cin, Cout, k, stride, H, W, use_res, act, gn):
    """Split-K route (otvm_conv_params.splitk_ws): same result as the single-pass kernel up to fp32 summation order,
    deterministic, bias / residual / activation / GroupNorm sums applied after the reduction."""
    pad = k // 2
    x = rnd(1, Cin, H, W, seed=90)
    w = rnd(Cout, Cin, k, k, seed=91, scale=1.0 / math.sqrt(Cin * k * k))
    b = rnd(Cout, seed=92)
    ref = F.conv2d(x, w, b, stride, pad)
    res = rnd(*ref.shape, seed=93) if use_res else None
    if use_res:
        ref = ref + res
    ref = F.relu(ref) if act == 1 else (F.leaky_relu(ref, 0.01) if act == 2 else ref)
    # the patch kernel would take 3x3 stride-1 layers before the split is considered: pack without fragment weights
    cw = G.pack_weight(w)
    cw.w_frag = None
    xa = G.to_act(x)
    ra = G.to_act(res) if use_res else None
    bd = b.to(G.DEV)
    ws = torch.empty(16 << 20, device=G.DEV)
    outs = []
    for use_ws in (ws, None, ws):
        out = G.empty_act(ref.shape[2], ref.shape[3], Cout)
        stats = torch.zeros(64, dtype=torch.float64, device=G.DEV) if gn else None
        G.conv2d(xa, cw, out, bd, stride, pad, 1, act, 0, ra, precision=1, gn_stats=stats, splitk_ws=use_ws)
        outs.append((G.from_act(out, Cout), None if stats is None else stats.cpu()))
    got, plain, again = outs
    assert G.maxdiff(got[0], ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert G.maxdiff(got[0], plain[0]) <= 1e-5 * max(1.0, float(ref.abs().max())) and not torch.equal(got[0], plain[0])
    assert torch.equal(got[0], again[0])                                   # fixed reduction order
    if gn:
        g = ref.double().reshape(32, Cout // 32, -1)
        want = torch.stack([g.sum((1, 2)), (g * g).sum((1, 2))], 1).flatten()
        assert float((got[1] - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_conv_fuzz_all_routes(G):
    """tools/conv_fuzz.py: randomised shapes / views / epilogues over every dispatch route (implicit-GEMM tiles, patch
    kernels, split-K, both precisions) against torch on the CPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "conv_fuzz.py"), "--n", "80", "--seed", "3"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "conv_fuzz: 80 cases" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _probes_lib():
    """libotvm_hip_probes.so (python otvm_amd/csrc/build.py --probes): the same kernels with -DOTVM_PROBES, i.e. the OTVM_*
    ablation switches readable from the environment; the shipping library has them compiled to their defaults (round 6)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "otvm_amd", "libotvm_hip_probes.so")
    if not os.path.exists(path):
        pytest.skip("libotvm_hip_probes.so is not built (python otvm_amd/csrc/build.py --probes)")
    return path


@pytest.mark.parametrize("m16", ["1", "0"], ids=["mfma16x16x32", "mfma32x32x16"])
def test_nine_tap_patch_tiles_in_either_matrix_core_form(G, m16):
    """The 64-filter patch tiles run on v_mfma_f32_16x16x32_f16 by default (round 5; planar LDS patch, tap pairs in the K = 32 of
    one instruction, permuted accumulator rows, its own fused GroupNorm sums); OTVM_PATCH_M16=0 (probes library) keeps the 32x32x16 form (A/B
    runs).  The switch is read once per process: both forms are checked in processes of their own on ragged blocks, every
    dilation, 16-channel stages that do not fill a 32-channel block, bias / residual / activation / fused statistics."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if m16 != "1":                                          # (the default form needs no switch: the shipping library)
        env.update(OTVM_HIP_LIB=_probes_lib(), OTVM_PATCH_M16=m16)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "conv_fuzz.py"), "--n", "40", "--seed", "9", "--patch64"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "conv_fuzz: 40 cases" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_patch_tiles_on_views_beyond_2_gib(G):
    """A full-resolution layer of a 4K frame is a 2.7 GB view (2176 x 3840 pixels x 80 floats): byte offsets use all 32 bits, there
    is no flag bit to spare (round 6: the lean staging of the nine-tap tiles first marked outside pixels with bit 31 and zeroed
    every NORMALISED pixel beyond 2 GiB -- found by the 4K frame test).  The 64-filter tile with fused input normalisation on such a
    view: its last rows must equal, bit for bit, the same convolution run on a small view of the same rows."""
    from otvm_amd import lib as L
    from otvm_amd.engine import Act, conv_params
    lib = L.load()
    H, W, Cin, ld, Cout = 2176, 3840, 64, 80, 64
    g = torch.Generator(device=G.DEV).manual_seed(5)
    x = Act(torch.randn(H * W * ld + 16, device=G.DEV, generator=g), H, W, Cin, ld, 0)
    assert H * W * ld * 4 > (1 << 31)
    w = rnd(Cout, Cin, 3, 3, seed=7, scale=1.0 / math.sqrt(Cin * 9))
    cw = G.pack_weight(w)
    sc, sh = (rnd(Cin, seed=8).abs() + 0.5).to(G.DEV), (rnd(Cin, seed=9) * 0.3).to(G.DEV)
    out = G.empty_act(H, W, Cout)
    p = conv_params(x, cw, out, None, 1, 1, 1, 0, 0, None, L.PREC_F16X3, (sc.data_ptr(), sh.data_ptr(), 2))
    L.check(lib.otvm_conv2d(C.byref(p), G.stream()), "patch conv on a 2.7 GB view")
    rows = 40                                                   # the last rows of the map, plus one row of context above them
    xs = Act(x.t, rows + 1, W, Cin, ld, (H - rows - 1) * W * ld)
    outs = G.empty_act(rows + 1, W, Cout)
    ps = conv_params(xs, cw, outs, None, 1, 1, 1, 0, 0, None, L.PREC_F16X3, (sc.data_ptr(), sh.data_ptr(), 2))
    L.check(lib.otvm_conv2d(C.byref(ps), G.stream()), "the same rows as a small view")
    torch.cuda.synchronize()
    big = out.torch()[H - rows:]
    small = outs.torch()[1:]                                    # (row 0 of the small view sees zero padding above: not comparable)
    assert torch.isfinite(big).all() and float(big.abs().max()) > 0.1
    assert torch.equal(big, small), float((big - small).abs().max())


def test_tile_walk_is_bit_identical(G):
    """csrc/common.h: the spatially tiled kernels (patch convs, stems, head conv, fused bottleneck) walk their tiles in XCD-aware
    bands by default (round 5: halo lines are fetched once per XCD L2 instead of once per neighbour).  The walk decides which
    workgroup computes which tile, never a tile's arithmetic: row-major (OTVM_TILE_WALK=0), every family in bands of 8 and bands of
    3 tile rows (short last bands, one and two channel tiles per position) give bit-identical outputs.  The switch is read once
    per process, so every walk runs in a process of its own (tools/tile_walk_check.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    probes = _probes_lib()
    for env in ({"OTVM_TILE_WALK": "0"}, {"OTVM_TILE_WALK": "15"}, {"OTVM_TILE_WALK": "15", "OTVM_TILE_BAND": "3"}, {}):
        if env:                                             # ({}: the shipping library and its compiled-in walk)
            env = dict(env, OTVM_HIP_LIB=probes)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "tile_walk_check.py")], capture_output=True, text=True,
                           timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0 and "tile_walk_check: 11 outputs" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        digests.append(r.stdout.strip().split()[-1])
    assert len(set(digests)) == 1, digests


def test_gn_table_tail_stress_short(G):
    """tools/gn_tail_stress.py, short form: the GroupNorm table written by a conv's LAST workgroup (common.h::
    otvm_gn_table_tail -- device-scope statistics atomics, a workgroup-scope fence, a ticket) against otvm_gn_table over the
    finished statistics, bit for bit, on five layer shapes x 150 launches.  The ordering argument rests on gfx950 behaviour
    (device-scope atomics are acknowledged past the XCD's L2 once vmcnt reaches 0), not on the HIP memory model: this test is
    the guard should a compiler or ISA change break it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gn_tail_stress.py"), "--reps", "150"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and "gn_tail_stress: 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_kernel_fuzz_non_conv(G):
    """tools/kernel_fuzz.py: GroupNorm (incl. group widths that are not multiples of 4), upsampling, pooling, PPM pooling,
    memory read and the distance encoding on randomised shapes against torch / the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_fuzz.py"), "--n", "24", "--seed", "7"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "kernel_fuzz: 24 rounds" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_fused_groupnorm_stats_reject_odd_group_width(G):
    """The in-tile group reduction needs 32 groups of a power-of-two number of channels: anything else fails loudly."""
    x, w = rnd(1, 32, 12, 16, seed=95), rnd(192, 32, 1, 1, seed=96)
    out = G.empty_act(12, 16, 192)
    stats = torch.zeros(64, dtype=torch.float64, device=G.DEV)
    with pytest.raises(RuntimeError):
        G.conv2d(G.to_act(x), G.pack_weight(w), out, precision=1, gn_stats=stats)


def test_conv_big_tile_fused_groupnorm_stats(G):
    """GroupNorm sums out of the two-stage 256-row tiles (the group sums live behind the epilogue patches in LDS)."""
    Cin, Cout, H, W = 64, 256, 352, 353
    x = rnd(1, Cin, H, W, seed=73)
    w = rnd(Cout, Cin, 1, 1, seed=74, scale=1.0 / math.sqrt(Cin))
    b = rnd(Cout, seed=75)
    ref = F.conv2d(x, w, b).double()
    g = ref.reshape(32, Cout // 32, H * W)
    want = torch.stack([g.sum((1, 2)), (g * g).sum((1, 2))], 1).flatten()
    stats = torch.zeros(64, dtype=torch.float64, device=G.DEV)
    out = G.empty_act(H, W, Cout)
    G.conv2d(G.to_act(x), G.pack_weight(w), out, b.to(G.DEV), precision=1, gn_stats=stats)
    assert G.maxdiff(G.from_act(out), ref.float()) <= 2e-5 * float(ref.abs().max())
    assert float((stats.cpu() - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("Cin,Cout,dil,H,W,act", [(64, 64, 1, 37, 70, 1), (64, 32, 1, 20, 33, 2), (256, 256, 2, 320, 320, 1)])
def test_conv_fused_input_groupnorm(G, Cin, Cout, dil, H, W, act):
    """GroupNorm apply of the producer folded into the conv's staging (otvm_conv_params.in_scale; these shapes take the patch
    kernel): equals otvm_gn_apply followed by the plain conv, including the zero padding of the NORMALISED tensor at the border."""
    from otvm_amd import lib as L
    lib = L.load()
    x = rnd(1, Cin, H, W, seed=80) * 1.7 + 0.3
    w = rnd(Cout, Cin, 3, 3, seed=81, scale=1.0 / math.sqrt(Cin * 9))
    b = rnd(Cout, seed=82)
    gamma, beta = rnd(Cin, seed=83).abs() + 0.5, rnd(Cin, seed=84) * 0.2
    xn = F.group_norm(x, 32, gamma, beta, 1e-5)
    xn = F.relu(xn) if act == 1 else F.leaky_relu(xn, 0.01)
    ref = F.conv2d(xn, w, b, 1, dil, dil)
    xa, cw = G.to_act(x), G.pack_weight(w)
    g_d, b_d, bias_d = gamma.to(G.DEV), beta.to(G.DEV), b.to(G.DEV)           # kept alive: raw pointers go to the library
    stats = torch.zeros(64, dtype=torch.float64, device=G.DEV)
    L.check(lib.otvm_gn_stats(xa.ptr, H * W, Cin, xa.ld, stats.data_ptr(), G.stream()))
    tab = torch.zeros(2 * Cin, device=G.DEV)
    L.check(lib.otvm_gn_table(stats.data_ptr(), H * W, Cin, g_d.data_ptr(), b_d.data_ptr(),
                              tab.data_ptr(), tab.data_ptr() + 4 * Cin, G.stream()))
    out = G.empty_act(H, W, max(4, Cout))
    from otvm_amd.engine import conv_params
    probe = conv_params(xa, cw, out, None, 1, dil, dil, 0, 0, None, 1, (tab.data_ptr(), tab.data_ptr() + 4 * Cin, act))
    assert lib.otvm_conv2d_accepts_input_norm(C.byref(probe)) == 1          # these shapes take the patch kernel
    G.conv2d(xa, cw, out, bias_d, pad=dil, dil=dil, precision=1, in_norm=(tab.data_ptr(), tab.data_ptr() + 4 * Cin, act))
    got = G.from_act(out, Cout)
    assert G.maxdiff(got, ref) <= 3e-5 * max(1.0, float(ref.abs().max()))
    # bit-identical to the two-pass route (same table arithmetic, same staging)
    xa2 = G.to_act(x)
    L.check(lib.otvm_gn_apply(xa2.ptr, H * W, Cin, xa2.ld, stats.data_ptr(), g_d.data_ptr(), b_d.data_ptr(), 0, 0, 0, 0, 0, act,
                              xa2.ptr, xa2.ld, G.stream()))
    out2 = G.empty_act(H, W, max(4, Cout))
    G.conv2d(xa2, cw, out2, bias_d, pad=dil, dil=dil, precision=1)
    assert torch.equal(G.from_act(out2, Cout), got)
    # combinations no kernel implements are rejected loudly: with in_relu, or in exact fp32
    w1 = rnd(Cout, Cin, 1, 1, seed=85)
    cw1 = G.pack_weight(w1)
    with pytest.raises(RuntimeError):
        G.conv2d(xa, cw1, out, None, precision=1, in_relu=1, in_norm=(tab.data_ptr(), tab.data_ptr() + 4 * Cin, act))
    with pytest.raises(RuntimeError):
        G.conv2d(xa, cw1, out, None, precision=0, in_norm=(tab.data_ptr(), tab.data_ptr() + 4 * Cin, act))


IGEMM_NORM_CASES = [
    # Cin, Cout, k, stride, dil, H, W, input act, residual, out act, fused statistics, tune code (0 = heuristic)
    (64, 256, 1, 1, 1, 37, 70, 1, False, 0, True, 0),                  # bn2 -> conv3 of a layer-1 bottleneck (+ statistics of bn3)
    (128, 512, 1, 1, 1, 20, 33, 1, False, 0, True, (1 + 1) * 16 + 1),  # 256x128 tile
    (512, 2048, 1, 1, 1, 24, 40, 1, False, 0, False, (0 + 1) * 16 + 1),   # 256x256 tile
    (128, 128, 3, 2, 1, 41, 57, 1, False, 0, True, 0),                 # 3x3 stride 2 (first block of a stage): zero padding of the NORMALISED tensor
    (256, 256, 3, 1, 2, 24, 40, 2, True, 1, False, (2 + 1) * 16 + 1),  # dilated 3x3 on the implicit-GEMM path, LeakyReLU input, residual + ReLU
    (1024, 256, 1, 1, 1, 17, 23, 1, False, 0, False, (3 + 1) * 16 + 4),   # 128x64 tile, K split over 4 workgroups
    (256, 128, 1, 1, 1, 30, 34, 0, False, 0, False, (10 + 1) * 16 + 1),   # pipelined 64x64 tile, no input activation
    (256, 256, 1, 1, 1, 33, 40, 1, False, 0, False, (8 + 1) * 16 + 1),    # 4-wave 128x256 tile
    # round 5: tile t with LDS-DMA weight stages = tile 32 + t
    (512, 2048, 1, 1, 1, 24, 40, 1, False, 0, True, (32 + 0 + 1) * 16 + 1),   # 256x256 + statistics
    (256, 256, 3, 1, 2, 24, 40, 2, True, 1, False, (32 + 1 + 1) * 16 + 1),    # 256x128: dilated 3x3 (zero padding of the NORMALISED tensor), residual + ReLU
    (1024, 256, 1, 1, 1, 17, 23, 1, False, 0, False, (32 + 1 + 1) * 16 + 4),  # 256x128, K split over 4 workgroups
    (128, 128, 3, 2, 1, 41, 57, 1, False, 0, True, (32 + 2 + 1) * 16 + 1),    # 128x128 (one activation stage, two weight stages): strided 3x3 + statistics
    (1024, 256, 1, 1, 1, 17, 23, 1, False, 0, False, (32 + 3 + 1) * 16 + 4),  # 128x64, K split over 4
    (256, 128, 1, 1, 1, 30, 34, 0, False, 0, False, (32 + 10 + 1) * 16 + 1),  # pipelined 64x64, no input activation
    (256, 256, 1, 1, 1, 33, 40, 1, False, 0, False, (32 + 8 + 1) * 16 + 1),   # 4-wave 128x256 (eight blocks per wave: two DMA bases)
    (64, 256, 1, 1, 1, 37, 70, 1, False, 0, True, (32 + 7 + 1) * 16 + 1),     # 4-wave 256x128, two chunks of K
    # ... and tile 64 + t: the same tile multiplying with v_mfma_f32_16x16x32_f16 (another accumulator layout: epilogue + statistics)
    (512, 2048, 1, 1, 1, 24, 40, 1, False, 0, True, (64 + 0 + 1) * 16 + 1),   # 256x256 + statistics (64 channels per group)
    (256, 256, 3, 1, 2, 24, 40, 2, True, 1, False, (64 + 1 + 1) * 16 + 1),    # 256x128: dilated 3x3, residual + ReLU
    (1024, 256, 1, 1, 1, 17, 23, 1, False, 0, True, (64 + 1 + 1) * 16 + 1),   # 256x128 + statistics (8 channels per group)
    (1024, 256, 1, 1, 1, 17, 23, 1, False, 0, False, (64 + 1 + 1) * 16 + 4),  # 256x128, K split over 4 workgroups
    (128, 128, 3, 2, 1, 41, 57, 1, False, 0, True, (64 + 2 + 1) * 16 + 1),    # 128x128: strided 3x3 + statistics (4 channels per group)
    (128, 512, 1, 1, 1, 20, 33, 1, False, 0, True, (64 + 4 + 1) * 16 + 1),    # 64x64 + statistics (16 channels per group)
    (1024, 256, 1, 1, 1, 17, 23, 1, False, 0, False, (64 + 3 + 1) * 16 + 4),  # 128x64, K split over 4
    (256, 128, 1, 1, 1, 30, 34, 0, False, 0, False, (64 + 10 + 1) * 16 + 1),  # pipelined 64x64, no input activation
    (256, 256, 1, 1, 1, 33, 40, 1, False, 0, False, (64 + 8 + 1) * 16 + 1),   # 4-wave 128x256
    (64, 256, 1, 1, 1, 37, 70, 1, False, 0, True, (64 + 7 + 1) * 16 + 1),     # 4-wave 256x128, two chunks of K
    (64, 64, 1, 1, 1, 37, 70, 1, False, 0, True, (64 + 5 + 1) * 16 + 1),      # 256x64 + statistics (2 channels per group)
]


@pytest.mark.parametrize("case", IGEMM_NORM_CASES, ids=lambda c: "c%d_%d_k%d_s%d_d%d_t%d" % (c[0], c[1], c[2], c[3], c[4], c[11]))
def test_conv_fused_input_groupnorm_implicit_gemm(G, case):
    """The same fusion on the implicit-GEMM kernels (whole 32-channel chunks: every GroupNorm'd tensor): bit-identical to
    otvm_gn_apply followed by the plain conv in the same configuration, batched launch == single launches."""
    from otvm_amd import lib as L
    from otvm_amd.engine import Act, conv_params
    lib = L.load()
    Cin, Cout, k, stride, dil, H, W, iact, use_res, act, gn, tune = case
    pad = dil * (k - 1) // 2
    B = 2
    xs = [rnd(1, Cin, H, W, seed=90 + b) * 1.7 + 0.3 for b in range(B)]
    w = rnd(Cout, Cin, k, k, seed=81, scale=1.0 / math.sqrt(Cin * k * k))
    bias = rnd(Cout, seed=82).to(G.DEV)
    gamma, beta = rnd(Cin, seed=83).abs() + 0.5, rnd(Cin, seed=84) * 0.2
    g_d, b_d = gamma.to(G.DEV), beta.to(G.DEV)
    cw = G.pack_weight(w)
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    xb = _batched_act(G, xs)
    rs = [rnd(1, Cout, Ho, Wo, seed=60 + b) for b in range(B)]
    rb = _batched_act(G, rs) if use_res else None
    stats_in = torch.zeros(B * 64, dtype=torch.float64, device=G.DEV)
    L.check(lib.otvm_gn_stats_b(xb.ptr, H * W, Cin, xb.ld, stats_in.data_ptr(), B, xb.bs, 64, G.stream()))
    tab = torch.zeros(B * 2 * Cin, device=G.DEV)
    L.check(lib.otvm_gn_table_b(stats_in.data_ptr(), H * W, Cin, g_d.data_ptr(), b_d.data_ptr(), tab.data_ptr(),
                                tab.data_ptr() + 4 * Cin, B, 64, 2 * Cin, G.stream()))
    ws = torch.empty(8 << 20, device=G.DEV)
    ob = Act(torch.full((B * (Ho * Wo * Cout + 64) + 16,), float("nan"), device=G.DEV), Ho, Wo, Cout, Cout, 0, B=B, bs=Ho * Wo * Cout + 64)
    st_b = torch.zeros(B * 128, dtype=torch.float64, device=G.DEV)
    p = conv_params(xb, cw, ob, bias, stride, pad, dil, act, 0, rb, 1, (tab.data_ptr(), tab.data_ptr() + 4 * Cin, iact, 2 * Cin), ws)
    p.tune = tune
    if gn:
        p.gn_stats, p.gn_bs = st_b.data_ptr(), 128
    assert lib.otvm_conv2d_accepts_input_norm(C.byref(p)) == 1
    L.check(lib.otvm_conv2d(C.byref(p), G.stream()), "conv with fused input normalisation")
    torch.cuda.synchronize()
    for b in range(B):
        # two-pass route on image b: apply pass into a copy, then the plain conv in the same configuration
        xa = G.to_act(xs[b])
        L.check(lib.otvm_gn_apply(xa.ptr, H * W, Cin, xa.ld, stats_in.data_ptr() + 8 * 64 * b, g_d.data_ptr(), b_d.data_ptr(), 0, 0, 0, 0, 0,
                                  iact, xa.ptr, xa.ld, G.stream()))
        o1 = G.empty_act(Ho, Wo, Cout)
        st1 = torch.zeros(64, dtype=torch.float64, device=G.DEV)
        p1 = conv_params(xa, cw, o1, bias, stride, pad, dil, act, 0, None if rb is None else rb.img(b), 1, None, ws)
        p1.tune = tune
        if gn:
            p1.gn_stats = st1.data_ptr()
        L.check(lib.otvm_conv2d(C.byref(p1), G.stream()), "plain conv")
        torch.cuda.synchronize()
        assert torch.equal(ob.torch(b), o1.torch()), "image %d" % b
        if gn:
            got = st_b[b * 128:b * 128 + 64]
            assert float((got - st1).abs().max()) <= 1e-9 * float(st1.abs().max())
            # ... and the sums themselves against float64 sums of the written tensor (the 16x16x32 tiles have their own reduction
            # over the accumulator layout: every group size of the frame is among the cases)
            if act == 0 and not use_res:
                yg = o1.torch().double().reshape(-1, 32, Cout // 32)
                want_s = torch.stack([yg.sum((0, 2)), (yg * yg).sum((0, 2))], 1).reshape(64)
                assert float((st1 - want_s).abs().max()) <= 2e-5 * float(want_s.abs().max()), "fused GroupNorm sums"
        xn = F.group_norm(xs[b], 32, gamma, beta, 1e-5)
        xn = F.relu(xn) if iact == 1 else (F.leaky_relu(xn, 0.01) if iact == 2 else xn)
        want = F.conv2d(xn, w, bias.cpu(), stride, pad, dil)
        if use_res:
            want = want + rs[b]
        want = F.relu(want) if act == 1 else want
        gotc = ob.torch(b).permute(2, 0, 1)[None].cpu()
        assert float((gotc - want).abs().max()) <= 3e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("Cin,Cout,k,dil,H,W,tune", [
    (256, 256, 3, 1, 40, 56, 0),                    # heuristic: patch or an LDS-DMA tile
    (256, 256, 3, 1, 40, 56, (32 + 0 + 1) * 16 + 1),    # 256x256 LDS-DMA tile (hi blocks only in the weight stage)
    (512, 128, 1, 1, 33, 47, (32 + 2 + 1) * 16 + 1),    # 128x128 LDS-DMA tile
    (512, 128, 1, 1, 33, 47, (3 + 1) * 16 + 2),         # register-staged 128x64, K split over 2
    (128, 512, 1, 1, 33, 47, (32 + 8 + 1) * 16 + 1),    # 4-wave 128x256 LDS-DMA tile (two DMA bases)
    (64, 64, 3, 1, 33, 47, 0),                      # narrow patch tile
    (80, 32, 3, 1, 50, 33, 0),                      # 32-channel patch tile, 5 stages
    (24, 64, 7, 2, 40, 64, (3 + 1) * 16 + 1),           # generic K decode (7x7 stem shape on an implicit-GEMM tile)
])
def test_conv_single_pass_f16_mode(G, Cin, Cout, k, dil, H, W, tune):
    """precision 2 (OTVM_PREC_F16, round 5): the labelled reduced-precision mode -- ONE MFMA pass on fp16-rounded operands in the
    implicit-GEMM and patch kernels.  Against a float64 convolution: the error of an fp16-operand product (2^-11 per operand,
    averaged over K), i.e. far above f16x3's and far below a wrong result -- and exactly the float64 convolution of the
    fp16-ROUNDED operands up to fp32 accumulation (the mode computes what it says)."""
    from otvm_amd import lib as L
    from otvm_amd.engine import conv_params
    lib = L.load()
    pad = dil * (k - 1) // 2
    stride = 2 if k == 7 else 1
    x = rnd(1, Cin, H, W, seed=410)
    w = rnd(Cout, Cin, k, k, seed=411, scale=1.0 / math.sqrt(Cin * k * k))
    b = rnd(Cout, seed=412)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride, pad, dil)
    cw, xa, bd = G.pack_weight(w), G.to_act(x), b.to(G.DEV)
    out = G.empty_act(ref.shape[2], ref.shape[3], Cout)
    ws = torch.empty(8 << 20, device=G.DEV)
    err = {}
    for prec in (L.PREC_F16X3, L.PREC_F16):
        out.t.fill_(float("nan"))
        p = conv_params(xa, cw, out, bd, stride, pad, dil, 0, 0, None, prec, None, ws)
        p.tune = tune
        L.check(lib.otvm_conv2d(C.byref(p), G.stream()), "conv precision %d" % prec)
        torch.cuda.synchronize()
        got = G.from_act(out, Cout).double()
        assert torch.isfinite(got).all()
        err[prec] = float((got - ref).abs().max()) / float(ref.abs().max())
    # the operands as the kernel rounds them: weights per filter scaled by a power of two (exact), round to nearest
    xh = x.half().double()
    wh = w.half().double()
    ref16 = F.conv2d(xh, wh, b.double(), stride, pad, dil)
    e16 = float((got - ref16).abs().max()) / float(ref.abs().max())
    print("   f16x3 %.2e, f16 %.2e of max|y| against float64; f16 against the float64 convolution of the rounded operands %.2e"
          % (err[L.PREC_F16X3], err[L.PREC_F16], e16))
    assert err[L.PREC_F16X3] <= 2e-5 and 1e-5 < err[L.PREC_F16] <= 4e-3, err
    assert e16 <= 3e-5, e16


def test_f16x3_is_fp32_class(G):
    """Error of the split-fp16 path vs an fp64 reference, next to the exact-fp32 MFMA path, on operands
    spanning 1e-3 .. 30 (the dropped lo*lo term is 2^-22 relative)."""
    x = rnd(1, 256, 24, 32, seed=60)
    x[:, ::7] *= 1e-3
    x[:, ::11] *= 30
    w = rnd(128, 256, 3, 3, seed=61, scale=0.02)
    w[::5] *= 1e-2
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    cw = G.pack_weight(w)
    errs = []
    for prec in (0, 1):
        out = G.empty_act(24, 32, 128)
        G.conv2d(G.to_act(x), cw, out, pad=1, precision=prec)
        errs.append(float((G.from_act(out).double() - ref).abs().max() / ref.abs().max()))
    print("relative error vs fp64: fp32-MFMA %.2e, f16x3 %.2e" % tuple(errs))
    assert errs[0] < 2e-6 and errs[1] < 5e-6


def test_weight_standardisation_and_bn_fold(G):
    from oracle.otvm_oracle import standardise_weight
    from otvm_amd import lib as L
    w = rnd(64, 24, 3, 3, seed=5, scale=0.05) + 0.02
    x = rnd(1, 24, 20, 28, seed=6)
    cw = G.pack_weight(w, ws=True)
    out = G.empty_act(20, 28, 64)
    ref = F.conv2d(x, standardise_weight(w), None, 1, 1)
    for prec in (0, 1):
        G.conv2d(G.to_act(x), cw, out, pad=1, precision=prec)
        assert G.maxdiff(G.from_act(out), ref) <= 3e-5 * float(ref.abs().max())
    # BatchNorm(eval) folded into the conv (scale into weights, shift into bias)
    g_, b_, m_, v_ = rnd(64, seed=7).abs() + 0.5, rnd(64, seed=8), rnd(64, seed=9), rnd(64, seed=10).abs() + 0.5
    dv = [t.to(G.DEV) for t in (g_, b_, m_, v_)]
    scale = torch.empty(64, device=G.DEV)
    bias = torch.empty(64, device=G.DEV)
    L.check(L.load().otvm_fold_bn(dv[0].data_ptr(), dv[1].data_ptr(), dv[2].data_ptr(), dv[3].data_ptr(), 1e-5, 64,
                                  scale.data_ptr(), bias.data_ptr(), G.stream()))
    cw = G.pack_weight(w, scale=scale.cpu())
    ref = F.relu(F.batch_norm(F.conv2d(x, w, None, 1, 1), m_, v_, g_, b_, False, 0.0, 1e-5))
    for prec in (0, 1):
        G.conv2d(G.to_act(x), cw, out, bias, pad=1, act=1, precision=prec)
        assert G.maxdiff(G.from_act(out), ref) <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("Cc,H,W,act,use_res", [(64, 33, 47, 1, False), (128, 16, 20, 2, False), (256, 9, 11, 1, True),
                                                (2048, 5, 7, 0, False), (1024, 6, 6, 1, True), (256, 1, 1, 2, False),
                                                (256, 2, 2, 2, False), (64, 128, 160, 2, True),
                                                (64, 300, 400, 1, True)])            # > 4096 blocks: the grid-stride loop
def test_groupnorm(G, Cc, H, W, act, use_res):
    from otvm_amd import lib as L
    lib = L.load()
    x = rnd(1, Cc, H, W, seed=11) * 3 + 1.5
    gamma, beta = rnd(Cc, seed=12) + 1, rnd(Cc, seed=13)
    res = rnd(1, Cc, H, W, seed=14) if use_res else None
    ref = F.group_norm(x, 32, gamma, beta, 1e-5)
    if use_res:
        ref = ref + res
    ref = F.relu(ref) if act == 1 else (F.leaky_relu(ref, 0.01) if act == 2 else ref)
    xa = G.to_act(x, ld=Cc + 4, off=4)
    out = G.empty_act(H, W, Cc)
    stats = torch.zeros(64, dtype=torch.float64, device=G.DEV)
    ra = G.to_act(res) if use_res else None
    gd, bd = gamma.to(G.DEV), beta.to(G.DEV)
    L.check(lib.otvm_gn_stats(xa.ptr, H * W, Cc, xa.ld, stats.data_ptr(), G.stream()))
    L.check(lib.otvm_gn_apply(xa.ptr, H * W, Cc, xa.ld, stats.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                              0 if ra is None else ra.ptr, 0 if ra is None else ra.ld, 0, 0, 0, act, out.ptr, out.ld, G.stream()))
    torch.cuda.synchronize()
    assert G.maxdiff(G.from_act(out), ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    if use_res:
        # the residual is itself a raw GroupNorm input whose apply pass is folded in (res_scale / res_shift / res_act):
        # out = act(GN(x) + leaky(GN_r(r))), as the refinement's first BasicBlock reads the un-normalised conv1 output
        g2, b2 = rnd(Cc, seed=15) + 1, rnd(Cc, seed=16)
        rn = F.leaky_relu(F.group_norm(res, 32, g2, b2, 1e-5), 0.01)
        ref2 = F.group_norm(x, 32, gamma, beta, 1e-5) + rn
        ref2 = F.relu(ref2) if act == 1 else (F.leaky_relu(ref2, 0.01) if act == 2 else ref2)
        st2 = torch.zeros(64, dtype=torch.float64, device=G.DEV)
        g2d, b2d = g2.to(G.DEV), b2.to(G.DEV)
        tab = torch.zeros(2 * Cc, device=G.DEV)
        L.check(lib.otvm_gn_stats(ra.ptr, H * W, Cc, ra.ld, st2.data_ptr(), G.stream()))
        L.check(lib.otvm_gn_table(st2.data_ptr(), H * W, Cc, g2d.data_ptr(), b2d.data_ptr(), tab.data_ptr(), tab.data_ptr() + 4 * Cc,
                                  G.stream()))
        out.t.fill_(float("nan"))
        L.check(lib.otvm_gn_apply(xa.ptr, H * W, Cc, xa.ld, stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), ra.ptr, ra.ld,
                                  tab.data_ptr(), tab.data_ptr() + 4 * Cc, 2, act, out.ptr, out.ld, G.stream()))
        torch.cuda.synchronize()
        assert G.maxdiff(G.from_act(out), ref2) <= 2e-5 * max(1.0, float(ref2.abs().max()))


def test_maxpool_upsample_ppm(G):
    from otvm_amd import lib as L
    lib = L.load()
    x = rnd(1, 64, 34, 50, seed=15)
    out = G.empty_act(17, 25, 64)
    xa = G.to_act(x)
    L.check(lib.otvm_maxpool3x3s2(xa.ptr, 34, 50, 64, xa.ld, out.ptr, out.ld, G.stream()))
    torch.cuda.synchronize()
    assert G.maxdiff(G.from_act(out), F.max_pool2d(x, 3, 2, 1)) == 0
    # x2 upsample with add, and arbitrary-size upsample (PPM 3x3 -> 17x30)
    add = rnd(1, 64, 68, 100, seed=16)
    out = G.empty_act(68, 100, 64, ld=80, off=8)
    aa = G.to_act(add)
    L.check(lib.otvm_upsample_bilinear(xa.ptr, 34, 50, 64, xa.ld, 0, 0, 0, aa.ptr, aa.ld, out.ptr, 68, 100, out.ld, G.stream()))
    torch.cuda.synchronize()
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) + add
    assert G.maxdiff(G.from_act(out), ref) <= 1e-5
    # GroupNorm apply + LeakyReLU folded into the resampling (FBA decoder: GN, LeakyReLU, then x2 upsample)
    gam, bet = rnd(64, seed=18) + 1, rnd(64, seed=19)
    gd, bd = gam.to(G.DEV), bet.to(G.DEV)
    st = torch.zeros(64, dtype=torch.float64, device=G.DEV)
    tab = torch.zeros(128, device=G.DEV)
    L.check(lib.otvm_gn_stats(xa.ptr, 34 * 50, 64, xa.ld, st.data_ptr(), G.stream()))
    L.check(lib.otvm_gn_table(st.data_ptr(), 34 * 50, 64, gd.data_ptr(), bd.data_ptr(), tab.data_ptr(), tab.data_ptr() + 256, G.stream()))
    out.t.fill_(float("nan"))
    L.check(lib.otvm_upsample_bilinear(xa.ptr, 34, 50, 64, xa.ld, tab.data_ptr(), tab.data_ptr() + 256, 2, 0, 0, out.ptr, 68, 100,
                                       out.ld, G.stream()))
    torch.cuda.synchronize()
    refn = F.interpolate(F.leaky_relu(F.group_norm(x, 32, gam, bet, 1e-5), 0.01), scale_factor=2, mode="bilinear", align_corners=False)
    assert G.maxdiff(G.from_act(out), refn) <= 2e-5 * max(1.0, float(refn.abs().max()))
    for s in (1, 2, 3, 6):
        y = rnd(1, 256, s, s, seed=17 + s)
        ya = G.to_act(y)
        out = G.empty_act(17, 30, 256)
        L.check(lib.otvm_upsample_bilinear(ya.ptr, s, s, 256, ya.ld, 0, 0, 0, 0, 0, out.ptr, 17, 30, out.ld, G.stream()))
        torch.cuda.synchronize()
        ref = F.interpolate(y, size=(17, 30), mode="bilinear", align_corners=False)
        assert G.maxdiff(G.from_act(out), ref) <= 1e-5
    # adaptive average pooling bins 1,2,3,6 on a 17x30 map
    z = rnd(1, 512, 17, 30, seed=30)
    za = G.to_act(z)
    pool = torch.empty(50 * 512, device=G.DEV)
    pws = torch.empty(int(lib.otvm_ppm_pool_ws_bytes(17, 512)), dtype=torch.uint8, device=G.DEV)
    L.check(lib.otvm_ppm_pool(za.ptr, 17, 30, 512, za.ld, pool.data_ptr(), pws.data_ptr(), G.stream()))
    torch.cuda.synchronize()
    base = 0
    for s in (1, 2, 3, 6):
        ref = F.adaptive_avg_pool2d(z, s)[0].permute(1, 2, 0).reshape(s * s, 512)
        got = pool[base * 512:(base + s * s) * 512].reshape(s * s, 512).cpu()
        assert G.maxdiff(got, ref) <= 1e-5
        base += s * s


@pytest.mark.parametrize("T,h,w", [(1, 5, 7), (2, 8, 12), (5, 9, 13), (3, 16, 20), (19, 6, 11)])
def test_memory_read(G, T, h, w):
    from oracle.otvm_oracle import memory_read
    from otvm_amd import lib as L
    lib = L.load()
    hw = h * w
    mk, mv = rnd(128, T, h, w, seed=40, scale=2.5), rnd(512, T, h, w, seed=41)
    qk, qv = rnd(128, h, w, seed=42, scale=2.5), rnd(512, h, w, seed=43)
    mk[:, 0, 0, 0] = qk[:, 1, 1] * 3            # a dominant match: forces large running-max jumps
    ref = memory_read(mk, mv, qk, qv)[:512].reshape(512, hw).t()
    keys = [mk[:, t].reshape(128, hw).t().contiguous().to(G.DEV) for t in range(T)]
    vals = [mv[:, t].reshape(512, hw).t().contiguous().to(G.DEV) for t in range(T)]
    q = qk.reshape(128, hw).t().contiguous().to(G.DEV)
    out = torch.full((hw, 1024), float("nan"), device=G.DEV)
    ws = torch.empty(int(lib.otvm_memory_read_ws_bytes(hw, T)), dtype=torch.uint8, device=G.DEV)
    kp = (C.c_void_p * T)(*[k.data_ptr() for k in keys])
    vp = (C.c_void_p * T)(*[v.data_ptr() for v in vals])
    L.check(lib.otvm_memory_read(q.data_ptr(), 128, kp, vp, T, hw, out.data_ptr(), 1024, ws.data_ptr(), G.stream()))
    torch.cuda.synchronize()
    got = out[:, :512].cpu()
    assert torch.isfinite(got).all()
    assert G.maxdiff(got, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    # f16x3 variant over the packed (split, fragment-major) bank
    slots = []
    for t in range(T):
        sl = torch.zeros(int(lib.otvm_bank_slot_bytes_f16x3(hw)), dtype=torch.uint8, device=G.DEV)
        L.check(lib.otvm_bank_pack_f16x3(keys[t].data_ptr(), vals[t].data_ptr(), hw, sl.data_ptr(), G.stream()))
        slots.append(sl)
    out.fill_(float("nan"))
    sp = (C.c_void_p * T)(*[s_.data_ptr() for s_ in slots])
    L.check(lib.otvm_memory_read_f16x3(q.data_ptr(), 128, sp, T, hw, out.data_ptr(), 1024, ws.data_ptr(), G.stream()))
    torch.cuda.synchronize()
    got = out[:, :512].cpu()
    assert torch.isfinite(got).all()
    assert G.maxdiff(got, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))


def _encode(G, probs, override=None):
    from otvm_amd import lib as L
    lib = L.load()
    _, Hp, Wp = probs.shape
    P = Hp * Wp
    pd = probs.contiguous().to(G.DEV)
    x11 = torch.zeros(P * 12, device=G.DEV)
    d80 = torch.zeros(P * 80, device=G.DEV)
    cls = torch.empty(P, dtype=torch.uint8, device=G.DEV)
    ws = torch.empty(int(lib.otvm_trimap_encode_ws_bytes(Hp, Wp)), dtype=torch.uint8, device=G.DEV)
    ov = None if override is None else override.to(torch.uint8).contiguous().to(G.DEV)
    L.check(lib.otvm_trimap_encode(pd.data_ptr(), Hp, Wp, 0 if ov is None else ov.data_ptr(), cls.data_ptr(),
                                   x11.data_ptr(), 12, d80.data_ptr(), 80, ws.data_ptr(), G.stream()))
    torch.cuda.synchronize()
    x = x11.reshape(Hp, Wp, 12).cpu()
    return x[..., 3:11].permute(2, 0, 1), cls.reshape(Hp, Wp).cpu(), d80.reshape(Hp, Wp, 80)[..., 70:72].cpu()


def test_trimap_encode(G):
    from oracle.otvm_oracle import make_trimap8, class_map
    from otvm_amd.synth_data import disc_trimap
    cases = []
    H, W = 64, 96
    cases.append(torch.from_numpy(disc_trimap(H, W)))
    allbg = torch.zeros(3, H, W); allbg[0] = 1
    cases.append(allbg)                                          # fg class empty -> zero triple
    allfg = torch.zeros(3, H, W); allfg[2] = 1
    cases.append(allfg)
    single = allbg.clone(); single[0, 20, 30] = 0; single[2, 20, 30] = 1
    cases.append(single)
    g = torch.Generator().manual_seed(5)
    cases.append(torch.softmax(torch.randn(3, H, W, generator=g) * 3, 0))        # noisy masks
    smooth = F.interpolate(torch.randn(1, 3, 8, 12, generator=g) * 4, size=(H, W), mode="bilinear")[0]
    cases.append(torch.softmax(smooth, 0))
    cases.append(torch.softmax(F.interpolate(torch.randn(1, 3, 5, 7, generator=g) * 4, size=(160, 224), mode="bilinear")[0], 0))
    for probs in cases:
        got, cls, tri2 = _encode(G, probs)
        ref = make_trimap8(probs)
        assert torch.equal(cls.long(), class_map(probs))
        assert G.maxdiff(got, ref) <= 2e-6, G.maxdiff(got, ref)
        assert torch.equal(tri2[..., 0], probs[0]) and torch.equal(tri2[..., 1], probs[2])
    # class-map override (tie-break synchronisation used by the sequence parity test)
    probs = cases[-2]
    ov = class_map(probs).clone()
    ov[10:14, 10:14] = 2
    got, cls, _ = _encode(G, probs, ov)
    assert torch.equal(cls.long(), ov)
    assert G.maxdiff(got, make_trimap8(probs, ov)) <= 2e-6


def test_fba_head(G):
    from oracle.otvm_oracle import fba_fusion
    from otvm_amd import lib as L
    lib = L.load()
    P = 37 * 41
    for n_out in (7, 10):
        hid = rnd(1, 16, 37, 41, seed=50)
        w, b = rnd(n_out, 16, 1, 1, seed=51, scale=0.4), rnd(n_out, seed=52, scale=0.3)
        img = torch.rand(1, 3, 37, 41, generator=torch.Generator().manual_seed(53))
        out = F.conv2d(hid, w, b)
        al = torch.clamp(out[:, 0:1], 0, 1)
        al, _, _ = fba_fusion(al, img, torch.sigmoid(out[:, 1:4]), torch.sigmoid(out[:, 4:7]))
        ha = G.to_act(hid, c_pad=16, ld=24, off=0)
        ia = G.to_act(img, c_pad=4, ld=8, off=0)
        alpha = torch.full((P * 2,), float("nan"), device=G.DEV)
        tri = torch.full((3 * P,), float("nan"), device=G.DEV)
        sm = torch.zeros(P * 24, device=G.DEV)
        wd, bd = w.reshape(n_out, 16).contiguous().to(G.DEV), b.to(G.DEV)
        L.check(lib.otvm_fba_head(ha.ptr, ha.ld, wd.data_ptr(), bd.data_ptr(), n_out, ia.ptr, ia.ld, P, alpha.data_ptr(), 2,
                                  tri.data_ptr() if n_out == 10 else 0, sm.data_ptr() + 64 if n_out == 10 else 0, 24,
                                  G.stream()))
        torch.cuda.synchronize()
        assert G.maxdiff(alpha[::2].cpu(), al.flatten()) <= 2e-6
        if n_out == 10:
            p = torch.softmax(out[:, 7:10], 1)[0].reshape(3, P)
            assert G.maxdiff(tri.reshape(3, P).cpu(), p) <= 2e-6
            s = sm.reshape(P, 24).cpu()
            assert G.maxdiff(s[:, 19], p[1]) <= 2e-6 and G.maxdiff(s[:, 20], p[2]) <= 2e-6
            assert G.maxdiff(s[:, 21], al.flatten()) <= 2e-6


@pytest.mark.parametrize("wide16", [False, True, "generic"], ids=["tile32", "tile16", "tile16-relu-nobias"])
@pytest.mark.parametrize("n_out,H,W,write_hid", [(7, 40, 64, False), (7, 37, 45, True), (10, 24, 96, True), (10, 19, 33, True)])
def test_conv_with_head_epilogue(G, n_out, H, W, write_hid, wide16):
    """otvm_conv2d_head: conv3x3 32 -> 16 + LeakyReLU with the 1x1 head + fba_fusion (+ softmax of the trimap logits) in its
    epilogue (FBA/models.py:383-388, 425-432) == otvm_conv2d followed by otvm_fba_head: the hidden state bit for bit (same
    tiles, same epilogue arithmetic), the head's outputs to fp32 rounding; interior and edge tiles, with and without the
    hidden state written.  "tile16-relu-nobias": the 16-wide kernel's generic instantiation (any activation, optional bias; the frame
    only issues LeakyReLU + bias, the compile-time fast path)."""
    from otvm_amd import lib as L
    from otvm_amd.engine import conv_params
    lib, st = L.load(), G.stream()
    P = H * W
    generic = wide16 == "generic"
    act = 1 if generic else 2
    x = rnd(1, 32, H, W, seed=80)
    w, b = rnd(16, 32, 3, 3, seed=81, scale=1.0 / math.sqrt(32 * 9)), rnd(16, seed=82, scale=0.2)
    hw, hb = rnd(n_out, 16, seed=83, scale=0.4).contiguous().to(G.DEV), rnd(n_out, seed=84, scale=0.3).to(G.DEV)
    img = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(85))
    xa, ia = G.to_act(x), G.to_act(img, c_pad=4, ld=8)
    cw = G.pack_weight(w)
    bd = None if generic else b.to(G.DEV)
    if generic:
        b = torch.zeros_like(b)
    # reference route
    hid0 = G.empty_act(H, W, 16, ld=24)
    G.conv2d(xa, cw, hid0, bd, pad=1, act=act, precision=1)
    a0 = torch.full((2 * P,), float("nan"), device=G.DEV)
    t0 = torch.full((3 * P,), float("nan"), device=G.DEV)
    sm0 = torch.zeros(P * 24, device=G.DEV)
    L.check(lib.otvm_fba_head(hid0.ptr, hid0.ld, hw.data_ptr(), hb.data_ptr(), n_out, ia.ptr, ia.ld, P, a0.data_ptr(), 2,
                              t0.data_ptr() if n_out == 10 else 0, sm0.data_ptr() + 64 if n_out == 10 else 0, 24, st))
    # fused
    hid1 = G.empty_act(H, W, 16, ld=24)
    a1 = torch.full((2 * P,), float("nan"), device=G.DEV)
    t1 = torch.full((3 * P,), float("nan"), device=G.DEV)
    sm1 = torch.zeros(P * 24, device=G.DEV)
    p = conv_params(xa, cw, hid1, bd, 1, 1, 1, act, 0, None, 1)
    if not write_hid:
        p.out, p.out_ld = 0, 0
    h = L.HeadParams()
    h.w, h.b, h.n_out, h.img, h.img_ld, h.P = hw.data_ptr(), hb.data_ptr(), n_out, ia.ptr, ia.ld, P
    h.alpha_out, h.alpha_stride = a1.data_ptr(), 2
    if n_out == 10:
        h.tri_out, h.sm, h.sm_ld = t1.data_ptr(), sm1.data_ptr() + 64, 24
    if wide16:                       # the same layer on v_mfma_f32_16x16x32_f16 (another summation order: fp32 rounding apart)
        assert cw.w16 is not None
        h.w16 = cw.w16.data_ptr()
    L.check(lib.otvm_conv2d_head(C.byref(p), C.byref(h), st), "conv2d_head")
    torch.cuda.synchronize()
    if write_hid and not wide16:
        assert torch.equal(G.from_act(hid1), G.from_act(hid0))
    if write_hid:
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        ref = (torch.relu(ref) if generic else F.leaky_relu(ref, 0.01)).float()
        assert G.maxdiff(G.from_act(hid1), ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert G.maxdiff(a1[::2].cpu(), a0[::2].cpu()) <= (1e-6 if not wide16 else 2e-5)
    if n_out == 10:
        assert G.maxdiff(t1.cpu(), t0.cpu()) <= (1e-6 if not wide16 else 2e-5)
        assert G.maxdiff(sm1.cpu(), sm0.cpu()) <= (1e-6 if not wide16 else 2e-5)
    # a layer it cannot take is refused
    p.Cout = 32
    assert lib.otvm_conv2d_head(C.byref(p), C.byref(h), st) != 0


def test_glue_kernels(G):
    from otvm_amd import lib as L
    from otvm_amd.synth_data import soft_alpha
    lib = L.load()
    H, W = 50, 70
    # first-frame trimap from GT alpha, narrow / medium / wide kernels (eval.py:67-72: 5 / 12 / 20 -> 11, 25 and 41
    # pixel windows; alpha/model.py:342-362)
    a = torch.from_numpy(soft_alpha(H, W, 0))
    for r in (5, 12, 20):
        trimask = ((a > 0) & (a < 1)).float()[None, None]
        tm = F.max_pool2d(trimask, 2 * r + 1, 1, r)[0, 0]
        t1 = torch.where(tm > 0.5, torch.ones_like(a), 2 * a).long()
        ref = F.one_hot(t1, 3).permute(2, 0, 1).float()
        out = torch.empty(3 * H * W, device=G.DEV)
        ws = torch.empty(H * W, dtype=torch.uint8, device=G.DEV)
        ad = a.contiguous().to(G.DEV)
        L.check(lib.otvm_trimap_from_alpha(ad.data_ptr(), H, W, r, out.data_ptr(), ws.data_ptr(), G.stream()))
        torch.cuda.synchronize()
        assert torch.equal(out.reshape(3, H, W).cpu(), ref)
    # pad / crop round trip and u8 truncation
    tri = torch.rand(3, H, W)
    Hp, Wp, lh, lw = 64, 96, 7, 13
    pd = torch.empty(3 * Hp * Wp, device=G.DEV)
    td = tri.contiguous().to(G.DEV)
    L.check(lib.otvm_pad_trimap(td.data_ptr(), H, W, pd.data_ptr(), Hp, Wp, lh, lw, G.stream()))
    ref = torch.cat([F.pad(tri[:1], (lw, Wp - W - lw, lh, Hp - H - lh), value=1.0),
                     F.pad(tri[1:], (lw, Wp - W - lw, lh, Hp - H - lh), value=0.0)])
    torch.cuda.synchronize()
    assert torch.equal(pd.reshape(3, Hp, Wp).cpu(), ref)
    alpha_p = torch.rand(Hp * Wp, device=G.DEV)
    al, au8, tr = torch.empty(H * W, device=G.DEV), torch.empty(H * W, dtype=torch.uint8, device=G.DEV), torch.empty(3 * H * W, device=G.DEV)
    L.check(lib.otvm_crop_outputs(alpha_p.data_ptr(), pd.data_ptr(), Hp, Wp, H, W, lh, lw, al.data_ptr(), au8.data_ptr(),
                                  tr.data_ptr(), G.stream()))
    torch.cuda.synchronize()
    ac = alpha_p.reshape(Hp, Wp)[lh:lh + H, lw:lw + W].cpu()
    assert torch.equal(al.reshape(H, W).cpu(), ac)
    assert torch.equal(au8.reshape(H, W).cpu(), (ac * 255).byte())
    assert torch.equal(tr.reshape(3, H, W).cpu(), tri)
    # one-hot of argmax
    oh = torch.empty(3 * H * W, device=G.DEV)
    L.check(lib.otvm_onehot_argmax3(td.data_ptr(), H * W, oh.data_ptr(), G.stream()))
    torch.cuda.synchronize()
    assert torch.equal(oh.reshape(3, H, W).cpu(), F.one_hot(tri.max(0)[1], 3).permute(2, 0, 1).float())


def test_matting_metrics(G):
    """On-device SAD / MSE / dtSSD (integer-exact sums) against the metrics oracle and the reference-generated fixture."""
    import os
    from oracle import metrics_oracle as M
    from otvm_amd.video import ClipMetrics
    from tests.common import GOLDEN
    ops = np.load(os.path.join(GOLDEN, "ops.npz"))
    p, t, m = (torch.from_numpy(ops[k]) for k in ("met_pred", "met_target", "met_mask"))
    cm = ClipMetrics(G.DEV)
    for i in range(p.shape[0]):
        cm.add(p[i].to(torch.uint8).to(G.DEV), t[i].to(torch.uint8).to(G.DEV), m[i].to(torch.uint8).to(G.DEV))
    torch.cuda.synchronize()
    r = cm.result()
    assert abs(r["sad_sum"] - float(ops["met_sad"].sum())) <= 1e-6 * float(ops["met_sad"].sum())
    assert abs(r["sad_sum"] - float(M.sad(p, t, m).double().sum())) <= 1e-6
    # MSE / dtSSD are per-frame ratios in the reference; the device accumulates their exact numerators/denominators
    num = float(((p - t).double().pow(2) * m.double()).sum()) / 255.0 ** 2
    assert abs(r["mse_num"] - num) <= 1e-9 * num and r["mask_sum"] == float(m.sum())
    e, n = M.dtssd(p, t, m)
    assert abs(r["dt_err2_sum"] - float(e.double().pow(2).sum())) <= 1e-5 * r["dt_err2_sum"]
    assert r["dt_mask_sum"] == float(m[:-1].sum())
    # per-frame values as the reference's BatchMetric methods return them (fixture: met_sad / met_mse / met_dtssd)
    assert np.allclose(r["sad_per_frame"], ops["met_sad"], rtol=1e-5, atol=1e-9)
    assert np.allclose(r["mse_per_frame"], ops["met_mse"], rtol=1e-5, atol=1e-12)
    assert np.allclose(r["dtssd_per_pair"], ops["met_dt_err"], rtol=1e-5, atol=1e-9)
    assert np.allclose(r["dtssd_num_per_pair"], ops["met_dt_num"], rtol=0, atol=0)
    # the reference's default mask (no mask given): the unknown band of the ground truth
    cm2 = ClipMetrics(G.DEV, capacity=2)                     # also exercises the row-buffer growth
    for i in range(p.shape[0]):
        cm2.add(p[i].to(torch.uint8).to(G.DEV), t[i].to(torch.uint8).to(G.DEV), "unknown")
    unk = ((t > 0) & (t < 255)).float()
    assert np.allclose(cm2.result()["sad_per_frame"], M.sad(p, t, unk).numpy(), rtol=1e-5, atol=1e-9)


# ---- the reference-held vectors of tests/golden/ops.npz (outputs of the imported reference's own functions,
# tests/golden/make_golden.py) fed straight to the HIP kernels through the C ABI
def _ops():
    import os
    from tests.common import GOLDEN
    return np.load(os.path.join(GOLDEN, "ops.npz"))


@pytest.mark.parametrize("T", [1, 2, 5])
def test_reference_vectors_memory_read(G, T):
    """Memory.forward (STM.py:144-163) vectors mem{1,2,5}_* -> otvm_memory_read / otvm_memory_read_f16x3."""
    from otvm_amd import lib as L
    lib = L.load()
    ops = _ops()
    mk, mv = torch.from_numpy(ops["mem%d_mk" % T][0]), torch.from_numpy(ops["mem%d_mv" % T][0])      # [C,T,h,w]
    qk = torch.from_numpy(ops["mem%d_qk" % T][0])
    want = torch.from_numpy(ops["mem%d_out" % T][0])                                                   # [1024,h,w]
    h, w = qk.shape[-2:]
    hw = h * w
    ref = want[:512].reshape(512, hw).t()
    keys = [mk[:, t].reshape(128, hw).t().contiguous().to(G.DEV) for t in range(T)]
    vals = [mv[:, t].reshape(512, hw).t().contiguous().to(G.DEV) for t in range(T)]
    q = qk.reshape(128, hw).t().contiguous().to(G.DEV)
    ws = torch.empty(int(lib.otvm_memory_read_ws_bytes(hw, T)), dtype=torch.uint8, device=G.DEV)
    out = torch.full((hw, 512), float("nan"), device=G.DEV)
    kp = (C.c_void_p * T)(*[k.data_ptr() for k in keys])
    vp = (C.c_void_p * T)(*[v.data_ptr() for v in vals])
    L.check(lib.otvm_memory_read(q.data_ptr(), 128, kp, vp, T, hw, out.data_ptr(), 512, ws.data_ptr(), G.stream()))
    torch.cuda.synchronize()
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    assert G.maxdiff(out.cpu(), ref) <= tol
    slots = []
    for t in range(T):
        sl = torch.zeros(int(lib.otvm_bank_slot_bytes_f16x3(hw)), dtype=torch.uint8, device=G.DEV)
        L.check(lib.otvm_bank_pack_f16x3(keys[t].data_ptr(), vals[t].data_ptr(), hw, sl.data_ptr(), G.stream()))
        slots.append(sl)
    out.fill_(float("nan"))
    sp = (C.c_void_p * T)(*[s_.data_ptr() for s_ in slots])
    L.check(lib.otvm_memory_read_f16x3(q.data_ptr(), 128, sp, T, hw, out.data_ptr(), 512, ws.data_ptr(), G.stream()))
    torch.cuda.synchronize()
    assert G.maxdiff(out.cpu(), ref) <= tol
    # the second half of the reference output is the query value passed through (STM.py:161)
    assert torch.equal(want[512:], torch.from_numpy(ops["mem%d_qv" % T][0]))


def test_reference_vectors_trimap_transform(G):
    """trimap_transform (utils/utils.py:25-39) vectors tt_masks -> tt_out through otvm_trimap_encode: the hard
    (bg, fg) masks become one-hot probabilities (unknown = neither), whose argmax is the same class map."""
    ops = _ops()
    for m, ref in zip(ops["tt_masks"], ops["tt_out"]):
        m = torch.from_numpy(m)
        probs = torch.stack([m[0], 1.0 - m[0] - m[1], m[1]])
        got, cls, _ = _encode(G, probs)
        assert G.maxdiff(got[:6], torch.from_numpy(ref)) <= 2e-6
    assert float(np.abs(ops["tt_out"][0][3:]).max()) == 0.0            # empty class -> zero triple on both sides


def test_reference_vectors_ws_conv_groupnorm(G):
    """layers_WS.Conv2d + GroupNorm(32) vector (layers_WS.py:6-27): weight standardisation at pack time, dilated 3x3
    on both precisions, fused statistics + apply."""
    from otvm_amd import lib as L
    lib = L.load()
    ops = _ops()
    x, w = torch.from_numpy(ops["ws_x"]), torch.from_numpy(ops["ws_w"])
    ref = torch.from_numpy(ops["ws_out"])
    _, Cin, H, W = x.shape
    Cout = w.shape[0]
    for prec in (0, 1):
        cw = G.pack_weight(w, ws=True)
        xa = G.to_act(x)
        raw = G.empty_act(H, W, Cout)
        stats = torch.zeros(64, dtype=torch.float64, device=G.DEV)
        G.conv2d(xa, cw, raw, bias=torch.from_numpy(ops["ws_b"]).to(G.DEV), pad=2, dil=2, precision=prec, gn_stats=stats)
        gam, bet = torch.from_numpy(ops["ws_g"]).to(G.DEV), torch.from_numpy(ops["ws_be"]).to(G.DEV)
        out = G.empty_act(H, W, Cout)
        L.check(lib.otvm_gn_apply(raw.ptr, H * W, Cout, raw.ld, stats.data_ptr(), gam.data_ptr(), bet.data_ptr(), 0, 0, 0, 0, 0, 0,
                                  out.ptr, out.ld, G.stream()))
        torch.cuda.synchronize()
        assert G.maxdiff(G.from_act(out), ref) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_reference_vectors_fba_fusion(G):
    """fba_fusion (FBA/models.py:279-288) vector ff_* through otvm_fba_head: an identity 1x1 head hands the kernel
    alpha and the logits of F, B, so what is compared is the fusion itself (incl. the F-then-B update order)."""
    from otvm_amd import lib as L
    lib = L.load()
    ops = _ops()
    a, img, Fg, Bg = (torch.from_numpy(ops[k]) for k in ("ff_a", "ff_img", "ff_F", "ff_B"))
    want = torch.from_numpy(ops["ff_out"])
    _, _, H, W = a.shape
    P = H * W
    logit = lambda p: torch.log(p.double() / (1 - p.double())).float()
    hid = torch.zeros(1, 16, H, W)
    hid[:, 0:1], hid[:, 1:4], hid[:, 4:7] = a, logit(Fg), logit(Bg)
    wd = torch.eye(16)[:7].contiguous().to(G.DEV)
    bd = torch.zeros(7, device=G.DEV)
    ha, ia = G.to_act(hid, c_pad=16), G.to_act(img, c_pad=4)
    alpha = torch.full((P,), float("nan"), device=G.DEV)
    L.check(lib.otvm_fba_head(ha.ptr, ha.ld, wd.data_ptr(), bd.data_ptr(), 7, ia.ptr, ia.ld, P, alpha.data_ptr(), 1, 0, 0, 0,
                              G.stream()))
    torch.cuda.synchronize()
    assert G.maxdiff(alpha.cpu(), want[0, 0].flatten()) <= 5e-6


@pytest.mark.parametrize("Cin,Cout,k,stride,dil,H,W,use_res,act,gn", [
    (256, 256, 3, 1, 1, 40, 56, True, 1, False),        # patch (wide) + all five tiles + K splits
    (96, 512, 3, 1, 1, 35, 70, False, 0, True),         # patch (wide): two channel tiles, six K stages (odd: both weight buffers end a run), ragged rows / columns, fused GroupNorm sums
    (1024, 512, 3, 1, 1, 17, 30, False, 0, False),      # deep, small map: the shape class the tuner moves to 256x256 / S
    (2048, 256, 1, 1, 1, 6, 6, False, 0, True),         # PPM 1x1 with fused GroupNorm sums: split-K + statistics pass
    (64, 64, 3, 1, 1, 33, 47, False, 2, False),         # narrow output: patch, 256x64, 128x64, 64x64
    (64, 32, 3, 1, 1, 37, 70, True, 1, False),          # 32 channels, ragged 8-row / 32-column blocks
    (80, 64, 3, 1, 1, 50, 33, False, 0, True),          # 80 input channels (5 stages), fused GroupNorm sums on the patch kernel
    (128, 128, 3, 2, 1, 34, 50, False, 1, False),       # strided 3x3
    (24, 64, 7, 2, 1, 40, 64, False, 1, False),         # stem: the 7x7 stem kernel and the generic K decode on every tile
    (12, 64, 7, 2, 1, 37, 51, False, 0, True),          # FBA stem: odd sizes, fused GroupNorm sums
])
def test_conv_every_tunable_configuration(G, Cin, Cout, k, stride, dil, H, W, use_res, act, gn):
    """otvm_conv2d_candidates / otvm_conv_params.tune: every configuration the plan-time autotuner may pick for a layer
    (patch kernel, the implicit-GEMM tiles, K splits) computes the same convolution; a code that is not legal for the
    layer is refused."""
    from otvm_amd import lib as L
    from otvm_amd.engine import conv_params
    lib = L.load()
    pad = dil * (k - 1) // 2
    x = rnd(1, Cin, H, W, seed=90)
    w = rnd(Cout, Cin, k, k, seed=91, scale=1.0 / math.sqrt(Cin * k * k))
    b = rnd(Cout, seed=92)
    ref = F.conv2d(x, w, b, stride, pad, dil)
    res = rnd(*ref.shape, seed=93) if use_res else None
    if use_res:
        ref = ref + res
    ref = F.relu(ref) if act == 1 else (F.leaky_relu(ref, 0.01) if act == 2 else ref)
    cw = G.pack_weight(w)
    xa = G.to_act(x)
    ra = G.to_act(res) if use_res else None
    bd = b.to(G.DEV)
    ws = torch.empty(16 << 20, device=G.DEV)
    out = G.empty_act(ref.shape[2], ref.shape[3], Cout)
    stats = torch.zeros(64, dtype=torch.float64, device=G.DEV) if gn else None
    p = conv_params(xa, cw, out, bd, stride, pad, dil, act, 0, ra, L.PREC_F16X3, None, ws)
    if gn:
        p.gn_stats = stats.data_ptr()
    codes = (C.c_int * 128)()
    n = int(lib.otvm_conv2d_candidates(C.byref(p), codes, 128))
    assert n >= 3 and len(set(codes[:n])) == n
    if Cin % 32 == 0 and Cout > 32:
        assert any(c // 16 - 1 == 9 for c in codes[:n]), "the one-wave 64x64 tile must be a candidate of a whole-chunk layer"
        assert any(c // 16 - 1 in (10, 42, 74) for c in codes[:n]) and any(c // 16 - 1 in (11, 43, 75) for c in codes[:n]), \
            "pipelined small tiles (32 + t: LDS-DMA form, 64 + t: on v_mfma_f32_16x16x32_f16)"
    seen_split = False
    for c in list(codes[:n]) + [0]:
        out.t.fill_(float("nan"))
        if gn:
            stats.zero_()
        p.tune = c
        L.check(lib.otvm_conv2d(C.byref(p), G.stream()), "tune %d" % c)
        torch.cuda.synchronize()
        got = G.from_act(out, Cout)
        assert torch.isfinite(got).all(), c
        assert G.maxdiff(got, ref) <= 2e-5 * max(1.0, float(ref.abs().max())), c
        seen_split |= (c & 15) > 1
        if gn:
            g = ref.double().reshape(32, Cout // 32, -1)
            want = torch.stack([g.sum((1, 2)), (g * g).sum((1, 2))], 1).flatten()
            assert float((stats.cpu() - want).abs().max()) <= 1e-5 * float(want.abs().max()), c
    assert seen_split or Cin * k * k < 512
    p.tune = (15 + 1) * 16 + 1                                 # no such tile
    assert lib.otvm_conv2d(C.byref(p), G.stream()) != 0
    p.tune = (32 + 9 + 1) * 16 + 1                             # the one-wave tile has no LDS-DMA form
    assert lib.otvm_conv2d(C.byref(p), G.stream()) != 0
    if Cout < 256:
        p.tune = (0 + 1) * 16 + 1                              # 256x256 needs Cout >= 256
        assert lib.otvm_conv2d(C.byref(p), G.stream()) != 0
    torch.cuda.synchronize()


def test_ppm_head_matches_conv_groupnorm_leakyrelu(G):
    """otvm_ppm_head: the four PPM heads (FBA/models.py:298-307) in one launch == 1x1 conv + bias -> GroupNorm(32) ->
    LeakyReLU of every pooled map, with weight standardisation applied by the packer as for every FBA conv."""
    from oracle.otvm_oracle import standardise_weight
    from otvm_amd import lib as L
    lib = L.load()
    pooled = rnd(50, 2048, seed=300)
    hp = L.PpmHeadParams()
    keep, want, outs = [], [], []
    base = 0
    for i, s in enumerate((1, 2, 3, 6)):
        w = rnd(256, 2048, 1, 1, seed=301 + i, scale=0.05)
        b = rnd(256, seed=311 + i)
        gamma, beta = 1.0 + 0.1 * rnd(256, seed=321 + i), 0.1 * rnd(256, seed=331 + i)
        cw = G.pack_weight(w, ws=True)
        wstd = standardise_weight(w.double())
        x = pooled[base:base + s * s].t().reshape(1, 2048, s, s).double()
        y = F.leaky_relu(F.group_norm(F.conv2d(x, wstd, b.double()), 32, gamma.double(), beta.double(), 1e-5), 0.01)
        want.append(y[0].permute(1, 2, 0).reshape(s * s, 256))
        out = torch.full((s * s, 256), float("nan"), device=G.DEV)
        t = [b.to(G.DEV), gamma.to(G.DEV), beta.to(G.DEV)]
        keep += [cw, out] + t
        hp.w[i], hp.bias[i], hp.gamma[i], hp.beta[i], hp.out[i] = cw.w.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), out.data_ptr()
        hp.K_pad = cw.K_pad
        outs.append(out)
        base += s * s
    pd = pooled.to(G.DEV)
    hp.pooled, hp.C, hp.Cout, hp.out_ld, hp.act = pd.data_ptr(), 2048, 256, 256, 2
    L.check(lib.otvm_ppm_head(C.byref(hp), G.stream()), "ppm_head")
    torch.cuda.synchronize()
    for i in range(4):
        d = float((outs[i].cpu().double() - want[i]).abs().max())
        # the 1x1 map normalises 8 values per group: a near-zero variance amplifies rounding, hence the looser bound there
        assert d <= (2e-3 if i == 0 else 2e-5) * max(1.0, float(want[i].abs().max())), (i, d)
    hp.C = 1024
    assert lib.otvm_ppm_head(C.byref(hp), G.stream()) != 0


# ---------------------------------------------------------------------------------------------- batched launches (ABI 11)
def _batched_act(G, xs, c_pad=None):
    """list of [1,C,H,W] CPU tensors -> one batched device Act (image b lies bs elements behind image 0) + the single views"""
    from otvm_amd.engine import Act, _rup
    _, Cc, H, W = xs[0].shape
    c_pad = _rup(Cc, 4) if c_pad is None else c_pad
    bs = H * W * c_pad + 32                                           # (a stride that is not the tensor size)
    buf = torch.zeros(len(xs) * bs + 16, dtype=torch.float32)
    for b, x in enumerate(xs):
        torch.as_strided(buf, (H, W, Cc), (W * c_pad, c_pad, 1), b * bs).copy_(x[0].permute(1, 2, 0))
    return Act(buf.to(G.DEV), H, W, c_pad, c_pad, 0, B=len(xs), bs=bs)


BATCH_CONV_CASES = [
    # Cin, Cout, k, stride, dil, H, W, residual, act, gn stats, forced tune code (0 = heuristic)
    (128, 256, 1, 1, 1, 20, 30, True, 1, False, 0),          # implicit GEMM 1x1 + residual + ReLU
    (256, 256, 3, 1, 1, 16, 24, False, 0, True, 0),          # 3x3 + fused GroupNorm statistics (per image)
    (64, 64, 3, 1, 1, 40, 56, True, 0, False, 0),            # patch kernel + residual
    (24, 64, 7, 2, 1, 64, 96, False, 1, False, 0),           # stem kernel
    (1024, 128, 3, 1, 1, 12, 16, False, 0, False, (3 + 1) * 16 + 4),   # 128x64 tile, K split over 4 workgroups
    (512, 128, 1, 1, 1, 17, 23, False, 0, True, (2 + 1) * 16 + 2),     # 128x128 / S2 + statistics pass behind the reduction
    (40, 128, 3, 2, 1, 33, 47, False, 2, False, 0),          # generic K decode, stride 2
    (256, 256, 3, 1, 1, 16, 24, True, 1, False, (9 + 1) * 16 + 1),     # one-wave 64x64 tile + residual + ReLU
    (1024, 128, 1, 1, 1, 12, 16, False, 0, True, (9 + 1) * 16 + 4),    # one-wave tile, K split over 4, statistics pass
    (256, 256, 3, 1, 1, 16, 24, True, 1, False, (10 + 1) * 16 + 2),    # pipelined 64x64 tile, K split over 2
    (512, 128, 1, 1, 1, 17, 23, False, 0, True, (11 + 1) * 16 + 1),    # pipelined 128x64 tile + fused statistics
    (256, 256, 3, 1, 1, 32, 40, True, 1, False, (13 + 1) * 16 + 1),    # 4-wave 256x256 tile (4x4 accumulator tiles per wave): interior tiles
    (256, 256, 1, 1, 1, 33, 40, False, 0, True, (13 + 1) * 16 + 1),    # same, edge tile + fused statistics
    (256, 256, 3, 1, 1, 32, 40, True, 1, False, (32 + 0 + 1) * 16 + 1),    # round 5, LDS-DMA weight stages (tile 32 + t): 256x256, interior tiles
    (128, 512, 1, 1, 1, 33, 40, False, 0, True, (32 + 1 + 1) * 16 + 1),    # 256x128: edge tiles, four channel tiles, fused statistics
    (256, 256, 3, 1, 1, 16, 24, True, 1, False, (32 + 4 + 1) * 16 + 2),    # 64x64, K split over 2, residual + ReLU
    (512, 128, 1, 1, 1, 17, 23, False, 0, True, (32 + 11 + 1) * 16 + 1),   # pipelined 128x64 + fused statistics
]


def test_wide_patch_tile_weight_stages_are_race_free(G):
    """The 256-channel patch tile copies its weight stages global -> LDS by LDS-DMA into alternating buffers (round 4): what
    orders that data for the fragment reads is a counted wait + a barrier, nothing else -- an early read would pass a
    tolerance check whenever the DMA happens to land first.  So: the forced patch configuration (tune 241) on a layer with
    six channel stages and two channel tiles, 40 launches back to back while another stream keeps the memory system busy,
    every result bit-identical to the first and within fp32 rounding of the reference."""
    from otvm_amd import lib as L
    from otvm_amd.engine import conv_params
    lib = L.load()
    Cin, Cout, H, W = 96, 512, 70, 101
    x = rnd(1, Cin, H, W, seed=190)
    w = rnd(Cout, Cin, 3, 3, seed=191, scale=1.0 / math.sqrt(Cin * 9))
    ref = F.conv2d(x, w, None, 1, 1, 1)
    cw, xa = G.pack_weight(w), G.to_act(x)
    out = G.empty_act(H, W, Cout)
    p = conv_params(xa, cw, out, None, 1, 1, 1, 0, 0, None, L.PREC_F16X3, None, None)
    codes = (C.c_int * 128)()
    n = int(lib.otvm_conv2d_candidates(C.byref(p), codes, 128))
    assert 241 in list(codes[:n]), "the patch kernel must be a candidate of a 3x3 layer with 512 filters"
    p.tune = 241
    side = torch.cuda.Stream(device=G.DEV)
    junk = torch.empty(64 << 20, device=G.DEV)
    first = None
    for it in range(40):
        with torch.cuda.stream(side):
            junk.add_(1.0)                                  # HBM / L2 traffic next to the launch
        out.t.fill_(float("nan"))
        L.check(lib.otvm_conv2d(C.byref(p), G.stream()), "patch")
        torch.cuda.synchronize()
        got = G.from_act(out, Cout)
        if first is None:
            first = got
            assert G.maxdiff(got, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
        else:
            assert torch.equal(got, first), it


@pytest.mark.parametrize("fam", [32, 64], ids=["mfma32x32x16", "mfma16x16x32"])
@pytest.mark.parametrize("tile,Cin,Cout,k,H,W", [(0, 256, 512, 3, 70, 101), (1, 96, 384, 3, 70, 101), (0, 2048, 256, 1, 68, 120),
                                                 (1, 64, 128, 1, 130, 200), (2, 256, 384, 3, 50, 61), (3, 512, 192, 1, 40, 50), (4, 320, 128, 1, 30, 40),
                                                 (7, 128, 256, 1, 70, 90), (8, 128, 512, 1, 70, 90), (10, 256, 128, 3, 30, 40), (11, 256, 192, 1, 60, 50)],
                         ids=["256x256_3x3", "256x128_3x3", "256x256_1x1", "256x128_short_k", "128x128_3x3", "128x64", "64x64", "256x128w4", "128x256w4",
                              "64x64D_3x3", "128x64D"])
def test_igemm_lds_dma_weight_stages_are_race_free(G, tile, Cin, Cout, k, H, W, fam):
    """The implicit-GEMM tiles with LDS-DMA weight stages (round 5) order the copied weights for the fragment reads with a
    hand-counted `s_waitcnt vmcnt(N)` + the chunk's barrier; the compiler does not know the copy exists.  An early read would
    pass a tolerance check whenever the DMA happens to land first, so: the forced configuration, 30 launches back to back
    while another stream keeps the memory system busy, every result bit-identical to the first, the first within fp32
    rounding of the reference AND bit-identical to the register-staged tile of the same shape (same summation order) -- the
    16x16x32 form of the tile (fam = 64) adds a chunk's 32 products inside one instruction instead of two 16-deep steps: the same
    products, within 2e-6 of the staged tile."""
    from otvm_amd import lib as L
    from otvm_amd.engine import conv_params
    lib = L.load()
    pad = (k - 1) // 2
    x = rnd(1, Cin, H, W, seed=290)
    w = rnd(Cout, Cin, k, k, seed=291, scale=1.0 / math.sqrt(Cin * k * k))
    ref = F.conv2d(x, w, None, 1, pad, 1)
    cw, xa = G.pack_weight(w), G.to_act(x)
    out = G.empty_act(H, W, Cout)
    p = conv_params(xa, cw, out, None, 1, pad, 1, 0, 0, None, L.PREC_F16X3, None, None)
    codes = (C.c_int * 128)()
    n = int(lib.otvm_conv2d_candidates(C.byref(p), codes, 128))
    # (OTVM_IGEMM_M16 = 2, the default: the 16x16x32 form replaces the 32x32x16 form in the list; that one stays legal when forced)
    assert (64 + tile + 1) * 16 + 1 in list(codes[:n]), "the LDS-DMA tile (16x16x32 form) must be a candidate of a whole-chunk layer"
    assert (tile + 1) * 16 + 1 not in list(codes[:n]), "... in place of the register-staged form"
    p.tune = (tile + 1) * 16 + 1                            # the register-staged tile of the same shape (still legal when forced)
    L.check(lib.otvm_conv2d(C.byref(p), G.stream()), "register-staged tile")
    torch.cuda.synchronize()
    staged = G.from_act(out, Cout)
    p.tune = (fam + tile + 1) * 16 + 1
    side = torch.cuda.Stream(device=G.DEV)
    junk = torch.empty(64 << 20, device=G.DEV)
    first = None
    for it in range(30):
        with torch.cuda.stream(side):
            junk.add_(1.0)                                  # HBM / L2 traffic next to the launch
        out.t.fill_(float("nan"))
        L.check(lib.otvm_conv2d(C.byref(p), G.stream()), "LDS-DMA tile")
        torch.cuda.synchronize()
        got = G.from_act(out, Cout)
        if first is None:
            first = got
            assert G.maxdiff(got, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
            if fam == 32:
                assert torch.equal(got, staged), "the two forms of the tile add the same products in the same order"
            else:
                assert G.maxdiff(got, staged) <= 2e-6 * max(1.0, float(ref.abs().max())), "the two forms of the tile add the same products"
        else:
            assert torch.equal(got, first), it


@pytest.mark.parametrize("case", BATCH_CONV_CASES, ids=lambda c: "c%d_%d_k%d_s%d_%dx%d_t%d" % (c[0], c[1], c[2], c[3], c[5], c[6], c[10]))
def test_conv_batched_launch_equals_single_launches(G, case):
    """otvm_conv_params.batch: B images through one launch == B single launches, bit for bit (same tiles, same summation
    order per image), on every kernel route; the fused GroupNorm statistics land in per-image blocks."""
    from otvm_amd import lib as L
    from otvm_amd.engine import Act, conv_params
    Cin, Cout, k, stride, dil, H, W, use_res, act, gn, tune = case
    B = 3
    pad = 3 if k == 7 else dil * (k - 1) // 2
    lib = L.load()
    xs = [rnd(1, Cin, H, W, seed=10 + b) for b in range(B)]
    w = rnd(Cout, Cin, k, k, seed=5, scale=1.0 / math.sqrt(Cin * k * k))
    cw = G.pack_weight(w, i_pad=24 if Cin == 24 else None)
    bias = rnd(Cout, seed=6).to(G.DEV)
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    xb = _batched_act(G, xs, cw.I_pad)
    rs = [rnd(1, Cout, Ho, Wo, seed=20 + b) for b in range(B)]
    rb = _batched_act(G, rs) if use_res else None
    ob = Act(torch.full((B * (Ho * Wo * Cout + 64) + 16,), float("nan"), device=G.DEV), Ho, Wo, Cout, Cout, 0, B=B, bs=Ho * Wo * Cout + 64)
    ws = torch.empty(8 << 20, device=G.DEV)
    stats_b = torch.zeros(B * 128, dtype=torch.float64, device=G.DEV)
    p = conv_params(xb, cw, ob, bias, stride, pad, dil, act, 0, rb, 1, None, ws)
    p.tune = tune
    if gn:
        p.gn_stats, p.gn_bs = stats_b.data_ptr(), 128
    L.check(lib.otvm_conv2d(C.byref(p), G.stream()), "batched conv")
    torch.cuda.synchronize()
    for b in range(B):
        o1 = G.empty_act(Ho, Wo, Cout)
        st1 = torch.zeros(64, dtype=torch.float64, device=G.DEV)
        p1 = conv_params(xb.img(b), cw, o1, bias, stride, pad, dil, act, 0, None if rb is None else rb.img(b), 1, None, ws)
        p1.tune = tune
        if gn:
            p1.gn_stats = st1.data_ptr()
        L.check(lib.otvm_conv2d(C.byref(p1), G.stream()), "single conv")
        torch.cuda.synchronize()
        assert torch.equal(ob.torch(b), o1.torch()), "image %d" % b
        if gn:
            got = stats_b[b * 128:b * 128 + 64]
            assert float((got - st1).abs().max()) <= 1e-9 * float(st1.abs().max()), "statistics of image %d" % b
    # against torch for one image
    want = F.conv2d(xs[1], w, bias.cpu(), stride, pad, dil)
    if use_res:
        want = want + rs[1]
    want = F.relu(want) if act == 1 else (F.leaky_relu(want, 0.01) if act == 2 else want)
    got = ob.torch(1).permute(2, 0, 1)[None].cpu()
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


def test_groupnorm_and_resampling_batched_equal_single(G):
    """otvm_gn_stats_b / otvm_gn_table_b / otvm_gn_apply_b / otvm_upsample_bilinear_b / otvm_maxpool3x3s2_b: per-image results
    identical to the single-image calls (per-image statistics and tables, shared gamma / beta)."""
    from otvm_amd import lib as L
    from otvm_amd.engine import Act
    lib, st = L.load(), G.stream()
    B, Cc, H, W = 3, 128, 18, 26
    xs = [rnd(1, Cc, H, W, seed=30 + b, scale=1.0 + b) for b in range(B)]
    rs = [rnd(1, Cc, H, W, seed=40 + b) for b in range(B)]
    xb, rb = _batched_act(G, xs), _batched_act(G, rs)
    gamma, beta = rnd(Cc, seed=1).to(G.DEV), rnd(Cc, seed=2).to(G.DEV)
    stats = torch.zeros(B * 96, dtype=torch.float64, device=G.DEV)
    L.check(lib.otvm_gn_stats_b(xb.ptr, xb.P, Cc, xb.ld, stats.data_ptr(), B, xb.bs, 96, st))
    tab = torch.zeros(B * 2 * Cc, device=G.DEV)
    L.check(lib.otvm_gn_table_b(stats.data_ptr(), xb.P, Cc, gamma.data_ptr(), beta.data_ptr(), tab.data_ptr(), tab.data_ptr() + 4 * Cc,
                                B, 96, 2 * Cc, st))
    ob = Act(torch.zeros(B * (H * W * Cc + 8), device=G.DEV), H, W, Cc, Cc, 0, B=B, bs=H * W * Cc + 8)
    q = L.GnApplyParams()
    q.x, q.P, q.C, q.ld, q.stats, q.gamma, q.beta = xb.ptr, xb.P, Cc, xb.ld, stats.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    q.residual, q.res_ld, q.res_scale, q.res_shift, q.res_act = rb.ptr, rb.ld, tab.data_ptr(), tab.data_ptr() + 4 * Cc, 2
    q.act, q.out, q.out_ld = 1, ob.ptr, ob.ld
    q.batch, q.x_bs, q.res_bs, q.out_bs, q.stats_bs, q.norm_bs = B, xb.bs, rb.bs, ob.bs, 96, 2 * Cc
    L.check(lib.otvm_gn_apply_b(C.byref(q), st))
    # x2 upsampling with the folded normalisation + add, general upsampling, max-pool
    ub = Act(torch.zeros(B * 4 * H * W * Cc, device=G.DEV), 2 * H, 2 * W, Cc, Cc, 0, B=B, bs=4 * H * W * Cc)
    addb = _batched_act(G, [rnd(1, Cc, 2 * H, 2 * W, seed=50 + b) for b in range(B)])
    L.check(lib.otvm_upsample_bilinear_b(xb.ptr, H, W, Cc, xb.ld, tab.data_ptr(), tab.data_ptr() + 4 * Cc, 2, addb.ptr, addb.ld, ub.ptr,
                                         2 * H, 2 * W, ub.ld, B, xb.bs, addb.bs, ub.bs, 2 * Cc, st))
    gb = Act(torch.zeros(B * 31 * 45 * Cc, device=G.DEV), 31, 45, Cc, Cc, 0, B=B, bs=31 * 45 * Cc)
    L.check(lib.otvm_upsample_bilinear_b(xb.ptr, H, W, Cc, xb.ld, 0, 0, 0, 0, 0, gb.ptr, 31, 45, gb.ld, B, xb.bs, 0, gb.bs, 0, st))
    mb = Act(torch.zeros(B * 9 * 13 * Cc, device=G.DEV), 9, 13, Cc, Cc, 0, B=B, bs=9 * 13 * Cc)
    L.check(lib.otvm_maxpool3x3s2_b(xb.ptr, H, W, Cc, xb.ld, mb.ptr, mb.ld, B, xb.bs, mb.bs, st))
    torch.cuda.synchronize()
    for b in range(B):
        x1, r1, a1 = xb.img(b), rb.img(b), addb.img(b)
        s1 = torch.zeros(64, dtype=torch.float64, device=G.DEV)
        L.check(lib.otvm_gn_stats(x1.ptr, x1.P, Cc, x1.ld, s1.data_ptr(), st))
        t1 = torch.zeros(2 * Cc, device=G.DEV)
        L.check(lib.otvm_gn_table(s1.data_ptr(), x1.P, Cc, gamma.data_ptr(), beta.data_ptr(), t1.data_ptr(), t1.data_ptr() + 4 * Cc, st))
        o1 = G.empty_act(H, W, Cc)
        L.check(lib.otvm_gn_apply(x1.ptr, x1.P, Cc, x1.ld, s1.data_ptr(), gamma.data_ptr(), beta.data_ptr(), r1.ptr, r1.ld,
                                  t1.data_ptr(), t1.data_ptr() + 4 * Cc, 2, 1, o1.ptr, o1.ld, st))
        u1, g1, m1 = G.empty_act(2 * H, 2 * W, Cc), G.empty_act(31, 45, Cc), G.empty_act(9, 13, Cc)
        L.check(lib.otvm_upsample_bilinear(x1.ptr, H, W, Cc, x1.ld, t1.data_ptr(), t1.data_ptr() + 4 * Cc, 2, a1.ptr, a1.ld, u1.ptr,
                                           2 * H, 2 * W, u1.ld, st))
        L.check(lib.otvm_upsample_bilinear(x1.ptr, H, W, Cc, x1.ld, 0, 0, 0, 0, 0, g1.ptr, 31, 45, g1.ld, st))
        L.check(lib.otvm_maxpool3x3s2(x1.ptr, H, W, Cc, x1.ld, m1.ptr, m1.ld, st))
        torch.cuda.synchronize()
        assert float((stats[b * 96:b * 96 + 64] - s1).abs().max()) <= 1e-9 * float(s1.abs().max())
        assert torch.equal(tab[b * 2 * Cc:(b + 1) * 2 * Cc], t1) or float((tab[b * 2 * Cc:(b + 1) * 2 * Cc] - t1).abs().max()) <= 1e-6
        assert float((ob.torch(b) - o1.torch()).abs().max()) <= 1e-5
        assert float((ub.torch(b) - u1.torch()).abs().max()) <= 1e-5
        assert torch.equal(gb.torch(b), g1.torch()) and torch.equal(mb.torch(b), m1.torch())
    want = F.group_norm(xs[2], 32, gamma.cpu(), beta.cpu(), 1e-5)
    tabr = F.leaky_relu(F.group_norm(rs[2], 32, gamma.cpu(), beta.cpu(), 1e-5), 0.01)
    # (the residual table is image 2's OWN table of x, applied to r: check the plain normalisation only)
    got = ob.torch(2).permute(2, 0, 1)[None].cpu()
    sc, sh = tab[2 * 2 * Cc:2 * 2 * Cc + Cc].cpu(), tab[2 * 2 * Cc + Cc:3 * 2 * Cc].cpu()
    res = F.leaky_relu(rs[2] * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), 0.01)
    assert float((got - F.relu(want + res)).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


def test_ppm_conv_algebra_matches_conv_over_upsampled_maps(G):
    """otvm_ppm_conv_z + otvm_ppm_conv_add: the PPM channels' share of conv_up1.0 (FBA/models.py:358-365) computed from the
    50 pooled pixels == F.conv2d over the four bilinearly upsampled maps (zero padding, align_corners=False), odd map size."""
    from otvm_amd import lib as L
    lib, st = L.load(), G.stream()
    H, W = 17, 29
    g = torch.Generator().manual_seed(8)
    ys = [torch.randn(1, 256, s, s, generator=g) for s in (1, 2, 3, 6)]
    w = torch.randn(256, 1024, 3, 3, generator=g) / math.sqrt(1024 * 9)
    ups = [F.interpolate(y, size=(H, W), mode="bilinear", align_corners=False) for y in ys]
    base = torch.randn(1, 256, H, W, generator=g)
    want = base + F.conv2d(torch.cat(ups, 1), w, None, 1, 1)
    ya = [G.to_act(y) for y in ys]
    yp = (C.c_void_p * 4)(*[a.ptr for a in ya])
    w_ppm = w.reshape(256, 4, 256, 9).permute(1, 3, 2, 0).contiguous().to(G.DEV)      # [scale][tap][c][o]
    Z = torch.zeros(9 * 50 * 256, device=G.DEV)
    out = G.to_act(base)
    L.check(lib.otvm_ppm_conv_z(yp, ya[0].ld, w_ppm.data_ptr(), Z.data_ptr(), st), "ppm_conv_z")
    stats = torch.zeros(64, dtype=torch.float64, device=G.DEV)
    L.check(lib.otvm_ppm_conv_add(Z.data_ptr(), H, W, out.ptr, out.ld, stats.data_ptr(), st), "ppm_conv_add")
    torch.cuda.synchronize()
    got = G.from_act(out)
    assert G.maxdiff(got, want) <= 2e-5 * max(1.0, float(want.abs().max()))
    gr = want.double().reshape(32, 8, -1)                              # fused GroupNorm(32) sums of the final tensor
    ws = torch.stack([gr.sum((1, 2)), (gr * gr).sum((1, 2))], 1).flatten()
    assert float((stats.cpu() - ws).abs().max()) <= 1e-5 * float(ws.abs().max())


# ---------------------------------------------------------------------------------------------- identity inside the fused input normalisation (ABI 17)
@pytest.mark.parametrize("Cout,H,W", [(32, 40, 64), (32, 37, 45), (64, 24, 96)])
def test_conv_input_norm_with_identity(G, Cout, H, W):
    """otvm_conv_params.in_res: conv3x3(relu(x * scale + shift + identity)) in ONE launch (the refinement's last BasicBlock:
    bn2 -> += identity -> ReLU -> pred.0, FBA/models.py:425-432 / resnet_GN_WS.py:36-48) == otvm_gn_apply(residual) followed by
    the plain convolution, bit for bit (same arithmetic per element, same tiles), interior and edge tiles."""
    from otvm_amd import lib as L
    from otvm_amd.engine import conv_params
    lib, st = L.load(), G.stream()
    Cin = 64
    x, idt = rnd(1, Cin, H, W, seed=70), rnd(1, Cin, H, W, seed=71).clamp_min(0)
    w, b = rnd(Cout, Cin, 3, 3, seed=72, scale=1.0 / math.sqrt(Cin * 9)), rnd(Cout, seed=73, scale=0.2)
    gam, bet = rnd(Cin, seed=74) * 0.2 + 1, rnd(Cin, seed=75) * 0.3
    xa, ia = G.to_act(x), G.to_act(idt)
    cw = G.pack_weight(w)
    bd, gd, btd = b.to(G.DEV), gam.to(G.DEV), bet.to(G.DEV)
    stats = torch.zeros(64, dtype=torch.float64, device=G.DEV)
    tab = torch.zeros(2 * Cin, device=G.DEV)
    L.check(lib.otvm_gn_stats(xa.ptr, H * W, Cin, xa.ld, stats.data_ptr(), st))
    L.check(lib.otvm_gn_table(stats.data_ptr(), H * W, Cin, gd.data_ptr(), btd.data_ptr(), tab.data_ptr(), tab.data_ptr() + 4 * Cin, st))
    # reference route: apply pass (+ identity, ReLU), then the convolution
    xn = G.empty_act(H, W, Cin, fill=0.0)
    L.check(lib.otvm_gn_apply(xa.ptr, H * W, Cin, xa.ld, stats.data_ptr(), gd.data_ptr(), btd.data_ptr(), ia.ptr, ia.ld, 0, 0, 0, 1,
                              xn.ptr, xn.ld, st))
    want = G.empty_act(H, W, Cout)
    G.conv2d(xn, cw, want, bd, pad=1, act=2, precision=1)
    got = G.empty_act(H, W, Cout)
    p = conv_params(xa, cw, got, bd, 1, 1, 1, 2, 0, None, 1, (tab.data_ptr(), tab.data_ptr() + 4 * Cin, 1))
    p.in_res, p.in_res_ld = ia.ptr, ia.ld
    assert lib.otvm_conv2d_accepts_input_residual(C.byref(p)) == 1
    L.check(lib.otvm_conv2d(C.byref(p), st), "conv (in_res)")
    torch.cuda.synchronize()
    assert torch.equal(G.from_act(got), G.from_act(want))
    ref = F.leaky_relu(F.conv2d(F.relu(F.group_norm(x, 32, gam, bet, 1e-5) + idt), w, b, padding=1), 0.01)
    assert G.maxdiff(G.from_act(got), ref) <= 2e-5 * max(1.0, float(ref.abs().max()))
    # a layer the patch kernel does not take must refuse the field
    p.dil, p.pad = 2, 2
    assert lib.otvm_conv2d_accepts_input_residual(C.byref(p)) == 0


def _predict_once(G, raw, P, planes, cw, mp, v, gamma, beta, passes, diag):
    """otvm_gram_f16 (no input normalisation) + otvm_gn_predict on the [P][planes] tensor ``raw``; returns (mean[32], rstd[32])."""
    from otvm_amd import lib as L
    lib, st = L.load(), G.stream()
    C4 = gamma.numel()
    nk = int(lib.otvm_gram_chunks(P, planes, None))
    ent = int(lib.otvm_gram_entries(planes))
    gpart = torch.empty(nk * ent, device=G.DEV)
    spart = torch.empty(nk * planes, device=G.DEV)
    q = L.GramParams()
    q.x, q.P, q.C, q.ld = raw.data_ptr(), P, planes, planes
    q.gpart, q.spart, q.passes, q.batch = gpart.data_ptr(), spart.data_ptr(), passes, 1
    q.diag = diag.data_ptr()
    L.check(lib.otvm_gram_f16(C.byref(q), st), "gram")
    pws = torch.empty(int(lib.otvm_gn_predict_ws_bytes()), dtype=torch.uint8, device=G.DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=G.DEV)
    eff = torch.full((2 * C4,), float("nan"), device=G.DEV)
    stat = torch.zeros(64, device=G.DEV)
    r = L.GnPredictParams()
    r.gpart, r.spart, r.P, r.C, r.Cout = gpart.data_ptr(), spart.data_ptr(), P, planes, C4
    r.Mp, r.v, r.ws, r.counter = mp.data_ptr(), v.data_ptr(), pws.data_ptr(), cnt.data_ptr()
    r.wscale, r.gamma, r.beta = cw.w_scale.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    r.scale_eff, r.bias_eff, r.stat_out, r.batch = eff.data_ptr(), eff.data_ptr() + 4 * C4, stat.data_ptr(), 1
    r.diag = diag.data_ptr()
    L.check(lib.otvm_gn_predict(C.byref(r), st), "gn_predict")
    torch.cuda.synchronize()
    return stat.view(32, 2)[:, 0].double().cpu(), stat.view(32, 2)[:, 1].double().cpu()


def test_gn_predict_conditioning_guard(G):
    """VERDICT r4 / ADVICE r4: var = E[y^2] - mean^2 from a single-pass fp16 Gram matrix cancels when a group's |mean| is much
    larger than its std.  Adversarial input: one output group whose mean is ~30x its std (kappa = mean^2 / var ~ 900), the others
    ordinary.  Checked: (a) the prediction kernel REPORTS the conditioning (diag[0] = max kappa, within 5 % of the float64 value),
    which is what the engine's automatic fall-back acts on (tests/test_gpu_frame.py: f16x3 Gram matrix above kappa = 4, the
    accumulated route above 256); (b) the error model behind those thresholds: relative error of rstd ~ (1 + kappa) x 2.8e-7 for
    the fp16 pass (measured 4.6e-4 at kappa = 1667) and ~ (1 + kappa) x 4.9e-9 for the f16x3 pass (8.2e-6), i.e. <= 1.5e-6 inside
    each regime the engine uses them in; the ordinary groups are at fp32 level either way; (c) operands beyond fp16's range are
    clamped -- finite statistics -- and flagged (diag[1] bit 1)."""
    from otvm_amd import lib as L
    from otvm_amd.engine import gram_tables
    lib = L.load()
    planes, C4, P = 64, 256, 40000
    cg = C4 // 32
    gen = torch.Generator().manual_seed(77)
    mu = torch.zeros(planes)
    mu[:32] = 1.0
    x = (mu[None, :] + 0.1 * torch.randn(P, planes, generator=gen)).float()
    w = torch.randn(C4, planes, generator=gen) * 0.2
    w[:cg] = (mu / 32.0)[None, :] + 0.003 * torch.randn(cg, planes, generator=gen)      # group 0: every filter ~ the mean detector
    cw = G.pack_weight(w.view(C4, planes, 1, 1).contiguous(), ws=False)
    mp, v = gram_tables(lib, cw)
    O_pad = cw.w.numel() // cw.K_pad
    wd = cw.w.view(O_pad, cw.K_pad)[:C4, :planes].double().cpu()
    y = x.double() @ wd.t()
    yg = y.view(P, 32, cg)
    mean64 = yg.mean((0, 2))
    var64 = (yg * yg).mean((0, 2)) - mean64 * mean64
    rstd64 = 1.0 / torch.sqrt(var64 + 1e-5)
    kappa64 = float((mean64 * mean64 / var64).max())
    assert 400 < kappa64 < 3000 and int((mean64 * mean64 / var64).argmax()) == 0, kappa64
    raw = x.to(G.DEV).contiguous()
    gamma, beta = torch.ones(C4, device=G.DEV), torch.zeros(C4, device=G.DEV)
    import struct
    err = {}
    for passes in (1, 3):
        diag = torch.zeros(2, dtype=torch.int32, device=G.DEV)
        mean_p, rstd_p = _predict_once(G, raw, P, planes, cw, mp, v, gamma, beta, passes, diag)
        kd = struct.unpack("f", struct.pack("i", int(diag[0].item())))[0]
        assert int(diag[1].item()) == 0 and abs(kd / kappa64 - 1.0) <= 0.05, (passes, kd, kappa64)
        rel = (rstd_p - rstd64).abs() / rstd64
        err[passes] = (float(rel[0]), float(rel[1:].max()))
        assert float(((mean_p - mean64).abs() / torch.maximum(mean64.abs(), 1.0 / rstd64)).max()) <= 2e-6
    print("   kappa %.0f: rstd relative error of the ill-conditioned group / worst ordinary group: fp16 Gram %.2e / %.2e, f16x3 Gram %.2e / %.2e"
          % (kappa64, err[1][0], err[1][1], err[3][0], err[3][1]))
    k1 = 1.0 + kappa64
    assert err[1][0] <= k1 * 6e-7 and err[3][0] <= k1 * 1.2e-8, (err, k1)      # the error model (2x margin on the measured constants)
    assert err[3][0] < 0.1 * err[1][0]
    assert err[1][1] <= 5e-6 and err[3][1] <= 1e-6, err        # ordinary groups: fp32 level with either pass
    # ---- (c) saturation: clamped, finite, flagged
    raw2 = raw.clone()
    raw2[123, 5] = 1.0e6
    raw2[4567, 40] = float("inf")
    diag = torch.zeros(2, dtype=torch.int32, device=G.DEV)
    mean_p, rstd_p = _predict_once(G, raw2, P, planes, cw, mp, v, gamma, beta, 1, diag)
    assert bool(torch.isfinite(mean_p).all()) and bool(torch.isfinite(rstd_p).all())
    assert int(diag[1].item()) & 2, "a clamped operand must be reported"


# ---------------------------------------------------------------------------------------------- predicted GroupNorm statistics (ABI 17)
PREDICT_CASES = [(64, 256, 272, 480), (128, 512, 136, 240), (256, 1024, 136, 240), (512, 2048, 136, 240), (192, 512, 37, 53)]


@pytest.mark.parametrize("passes", [1, 3])
@pytest.mark.parametrize("planes,C4,H,W", PREDICT_CASES, ids=["%d_%d_%dx%d" % c for c in PREDICT_CASES])
def test_gn_predict_matches_accumulated_statistics(G, planes, C4, H, W, passes):
    """otvm_gram_f16 + otvm_gn_predict (csrc/gram.hip): mean / rstd of GroupNorm(conv3(x')) predicted from the channel sums and
    the Gram matrix of conv3's INPUT x' = relu(GroupNorm(raw)) on the real 1080p bottleneck shapes of the FBA encoder
    (resnet_GN_WS.py:66-86) -- against the float64 statistics of the float64 convolution AND against the sums conv3's own
    epilogue accumulates (round 3's route); then the block tail itself: conv3 with the predicted scale / shift in its epilogue
    + identity + ReLU == conv3 -> otvm_gn_apply(+ identity, ReLU)."""
    from otvm_amd import lib as L
    from otvm_amd.engine import Act, conv_params, gram_tables
    lib, st = L.load(), G.stream()
    P = H * W
    g = torch.Generator(device=G.DEV).manual_seed(planes + passes)
    rawbuf = torch.zeros(P * planes + 16, device=G.DEV)
    raw = rawbuf[:P * planes].view(P, planes)
    raw.copy_(torch.randn(P, planes, generator=g, device=G.DEV) * 1.7 + 0.3)
    sc = (torch.rand(planes, generator=g, device=G.DEV) + 0.5).contiguous()
    sh = (torch.randn(planes, generator=g, device=G.DEV) * 0.4).contiguous()
    tab = torch.cat([sc, sh]).contiguous()
    w = torch.randn(C4, planes, 1, 1, generator=torch.Generator().manual_seed(5)) * 0.2 + 0.05
    cw = G.pack_weight(w, ws=True)
    mp, v = gram_tables(lib, cw)
    gamma = (torch.rand(C4, generator=g, device=G.DEV) + 0.5).contiguous()
    beta = (torch.randn(C4, generator=g, device=G.DEV) * 0.2).contiguous()
    # ---- float64 reference of the statistics
    xp = torch.relu(raw.double() * sc.double() + sh.double())
    O_pad = cw.w.numel() // cw.K_pad
    wstd = cw.w.view(O_pad, cw.K_pad)[:C4, :planes].double()
    y = xp @ wstd.t()
    cg = C4 // 32
    yg = y.view(P, 32, cg)
    mean64 = yg.mean((0, 2))
    var64 = (yg * yg).mean((0, 2)) - mean64 * mean64
    rstd64 = 1.0 / torch.sqrt(var64 + 1e-5)
    # ---- prediction
    nk = int(lib.otvm_gram_chunks(P, planes, None))
    ent = int(lib.otvm_gram_entries(planes))
    gpart = torch.empty(nk * ent, device=G.DEV)
    spart = torch.empty(nk * planes, device=G.DEV)
    q = L.GramParams()
    q.x, q.P, q.C, q.ld = raw.data_ptr(), P, planes, planes
    q.in_scale, q.in_shift, q.in_act = tab.data_ptr(), tab.data_ptr() + 4 * planes, 1
    q.gpart, q.spart, q.passes, q.batch = gpart.data_ptr(), spart.data_ptr(), passes, 1
    L.check(lib.otvm_gram_f16(C.byref(q), st), "gram")
    pws = torch.empty(int(lib.otvm_gn_predict_ws_bytes()), dtype=torch.uint8, device=G.DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=G.DEV)
    eff = torch.full((2 * C4,), float("nan"), device=G.DEV)
    stat = torch.zeros(64, device=G.DEV)
    r = L.GnPredictParams()
    r.gpart, r.spart, r.P, r.C, r.Cout = gpart.data_ptr(), spart.data_ptr(), P, planes, C4
    r.Mp, r.v, r.ws, r.counter = mp.data_ptr(), v.data_ptr(), pws.data_ptr(), cnt.data_ptr()
    r.wscale, r.gamma, r.beta = cw.w_scale.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    r.scale_eff, r.bias_eff, r.stat_out, r.batch = eff.data_ptr(), eff.data_ptr() + 4 * C4, stat.data_ptr(), 1
    L.check(lib.otvm_gn_predict(C.byref(r), st), "gn_predict")
    torch.cuda.synchronize()
    assert int(cnt.item()) == 0                                             # re-armed
    stat1 = stat.clone()
    L.check(lib.otvm_gram_f16(C.byref(q), st), "gram")
    L.check(lib.otvm_gn_predict(C.byref(r), st), "gn_predict")
    torch.cuda.synchronize()
    assert torch.equal(stat, stat1)                                         # no atomics on the way: bit-reproducible
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(3):
        L.check(lib.otvm_gram_f16(C.byref(q), st), "gram")
        L.check(lib.otvm_gn_predict(C.byref(r), st), "gn_predict")
    ev[0].record()
    for _ in range(10):
        L.check(lib.otvm_gram_f16(C.byref(q), st), "gram")
    ev[1].record()
    for _ in range(10):
        L.check(lib.otvm_gn_predict(C.byref(r), st), "gn_predict")
    ev[2].record()
    torch.cuda.synchronize()
    print("   timing: gram %.1f us, gn_predict %.1f us (block %d, %d chunks)" % (100 * ev[0].elapsed_time(ev[1]), 100 * ev[1].elapsed_time(ev[2]),
                                                                          int(lib.otvm_gram_block(planes)), nk))
    mean_p, rstd_p = stat.view(32, 2)[:, 0].double(), stat.view(32, 2)[:, 1].double()
    std64 = 1.0 / rstd64
    e_mean = float(((mean_p - mean64).abs() / torch.maximum(mean64.abs(), std64)).max())
    e_rstd = float(((rstd_p - rstd64).abs() / rstd64).max())
    # ---- round 3's route: the sums conv3's epilogue accumulates
    xa = Act(rawbuf, H, W, planes)
    t3 = G.empty_act(H, W, C4, fill=0.0)
    stats = torch.zeros(64, dtype=torch.float64, device=G.DEV)
    G.conv2d(xa, cw, t3, None, precision=1, gn_stats=stats, in_norm=(tab.data_ptr(), tab.data_ptr() + 4 * planes, 1))
    cntf = float(P * cg)
    sv = stats.view(32, 2)
    mean_a = sv[:, 0] / cntf
    rstd_a = 1.0 / torch.sqrt((sv[:, 1] / cntf - mean_a * mean_a).clamp_min(0) + 1e-5)
    a_mean = float(((mean_a - mean64).abs() / torch.maximum(mean64.abs(), std64)).max())
    a_rstd = float(((rstd_a - rstd64).abs() / rstd64).max())
    d_mean = float(((mean_p - mean_a).abs() / torch.maximum(mean_a.abs(), 1.0 / rstd_a)).max())
    d_rstd = float(((rstd_p - rstd_a).abs() / rstd_a).max())
    print("gn_predict %d->%d %dx%d passes %d: predicted vs float64 mean %.2e rstd %.2e | accumulated vs float64 mean %.2e rstd %.2e | "
          "predicted vs accumulated mean %.2e rstd %.2e" % (planes, C4, H, W, passes, e_mean, e_rstd, a_mean, a_rstd, d_mean, d_rstd))
    # (mean / rstd leave the kernel as fp32: 6e-8 of rounding each.  One fp16 pass: the operands' rounding noise averages out
    # over the P * cg values of a group -- 5e-7 .. 9e-7 measured on the 1080p shapes, 4e-6 on the 37 x 53 map)
    tol = 2.5e-7 if passes == 3 else (1e-6 if P >= 30000 else 1e-5)
    assert e_mean <= tol and e_rstd <= tol, (e_mean, e_rstd)
    assert d_mean <= tol and d_rstd <= tol, (d_mean, d_rstd)
    # ---- the block tail: predicted normalisation in conv3's epilogue == conv3 -> gn_apply(+ identity, ReLU)
    idt = G.empty_act(H, W, C4, fill=0.0)
    idt.t[:P * C4].copy_(torch.randn(P * C4, generator=g, device=G.DEV).clamp_min(0))
    want = G.empty_act(H, W, C4, fill=0.0)
    L.check(lib.otvm_gn_apply(t3.ptr, P, C4, t3.ld, stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), idt.ptr, idt.ld, 0, 0, 0, 1,
                              want.ptr, want.ld, st))
    got = G.empty_act(H, W, C4)
    p3 = conv_params(xa, cw, got, None, 1, 0, 1, 1, 0, idt, 1, (tab.data_ptr(), tab.data_ptr() + 4 * planes, 1))
    p3.w_scale, p3.bias = eff.data_ptr(), eff.data_ptr() + 4 * C4
    L.check(lib.otvm_conv2d(C.byref(p3), st), "conv3 (predicted normalisation)")
    torch.cuda.synchronize()
    gv, wv = got.t[:P * C4], want.t[:P * C4]
    d = float((gv - wv).abs().max())
    print("   block tail max-abs %.2e (range %.1f)" % (d, float(wv.abs().max())))
    assert d <= 2e-5 * max(1.0, float(wv.abs().max())), d
    # ---- a projection block: the identity is a raw GroupNorm input, scaled per channel in the epilogue, its shift in the bias
    rs_tab = torch.cat([torch.rand(C4, generator=g, device=G.DEV) + 0.5, torch.randn(C4, generator=g, device=G.DEV) * 0.3]).contiguous()
    want2 = G.empty_act(H, W, C4, fill=0.0)
    L.check(lib.otvm_gn_apply(t3.ptr, P, C4, t3.ld, stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), idt.ptr, idt.ld,
                              rs_tab.data_ptr(), rs_tab.data_ptr() + 4 * C4, 0, 1, want2.ptr, want2.ld, st))
    L.check(lib.otvm_gram_f16(C.byref(q), st), "gram")
    r.res_shift = rs_tab.data_ptr() + 4 * C4
    L.check(lib.otvm_gn_predict(C.byref(r), st), "gn_predict")
    got2 = G.empty_act(H, W, C4)
    p3.out, p3.res_scale = got2.ptr, rs_tab.data_ptr()
    L.check(lib.otvm_conv2d(C.byref(p3), st), "conv3 (predicted normalisation, scaled identity)")
    torch.cuda.synchronize()
    d2 = float((got2.t[:P * C4] - want2.t[:P * C4]).abs().max())
    assert d2 <= 2e-5 * max(1.0, float(want2.t[:P * C4].abs().max())), d2


# ---------------------------------------------------------------------------------------------- fused bottleneck (ABI 15)
@pytest.mark.parametrize("Cin,H,W", [(256, 8, 32), (256, 19, 45), (64, 16, 64), (64, 27, 70), (256, 68, 120)])
def test_stm_bottleneck_fused_kernel(G, Cin, H, W):
    """otvm_stm_bottleneck_f16x3 == relu(conv3(relu(conv2(relu(conv1 x)))) + identity) of the three (four) conv launches,
    with BatchNorm folded (STM.py:79-87 / torchvision Bottleneck): against fp64 torch, and image-wise identical in a
    batched launch; sizes that are not multiples of the 8x32 block exercise the halo and the edge masks."""
    from otvm_amd import lib as L
    from otvm_amd.engine import Act
    lib = L.load()
    B = 2
    proj = Cin == 64
    xs = [rnd(1, Cin, H, W, seed=30 + b).clamp_min(0) * 1.5 for b in range(B)]          # a ReLU output, as in the network
    w1 = rnd(64, Cin, 1, 1, seed=1, scale=1.0 / math.sqrt(Cin))
    w2 = rnd(64, 64, 3, 3, seed=2, scale=1.0 / math.sqrt(64 * 9))
    w3 = rnd(256, 64, 1, 1, seed=3, scale=1.0 / math.sqrt(64))
    wd = rnd(256, Cin, 1, 1, seed=4, scale=1.0 / math.sqrt(Cin))
    sc = [rnd(n, seed=40 + i).abs() + 0.5 for i, n in enumerate((64, 64, 256, 256))]  # folded BatchNorm scales
    bi = [rnd(n, seed=50 + i) * 0.3 for i, n in enumerate((64, 64, 256, 256))]
    c1 = G.pack_weight(w1, scale=sc[0])
    c2 = G.pack_weight(w2, scale=sc[1])
    if proj:
        c3 = G.pack_weight(torch.cat([w3 * sc[2][:, None, None, None], wd * sc[3][:, None, None, None]], dim=1))
        b3 = (bi[2] + bi[3]).to(G.DEV)
    else:
        c3 = G.pack_weight(w3, scale=sc[2])
        b3 = bi[2].to(G.DEV)
    b1, b2 = bi[0].to(G.DEV), bi[1].to(G.DEV)
    xb = _batched_act(G, xs)
    bs_o = H * W * 256 + 64
    ob = Act(torch.full((B * bs_o + 16,), float("nan"), device=G.DEV), H, W, 256, 256, 0, B=B, bs=bs_o)
    q = L.StmBottleneckParams(xb.ptr, H, W, Cin, xb.ld, ob.ptr, ob.ld, c1.w_wfrag.data_ptr(), c2.w_wfrag.data_ptr(),
                              c3.w_wfrag.data_ptr(), c1.w_scale.data_ptr(), c2.w_scale.data_ptr(), c3.w_scale.data_ptr(),
                              b1.data_ptr(), b2.data_ptr(), b3.data_ptr(), B, xb.bs, ob.bs, 0)
    L.check(lib.otvm_stm_bottleneck_f16x3(C.byref(q), G.stream()), "fused bottleneck")
    torch.cuda.synchronize()
    for b in range(B):
        x = xs[b].double()
        t = F.relu(F.conv2d(x, w1.double() * sc[0].double()[:, None, None, None], bi[0].double()))
        t = F.relu(F.conv2d(t, w2.double() * sc[1].double()[:, None, None, None], bi[1].double(), padding=1))
        y = F.conv2d(t, w3.double() * sc[2].double()[:, None, None, None], bi[2].double())
        idt = F.conv2d(x, wd.double() * sc[3].double()[:, None, None, None], bi[3].double()) if proj else x
        want = F.relu(y + idt)
        got = ob.torch(b).permute(2, 0, 1)[None].cpu().double()
        assert torch.isfinite(got).all()
        d = float((got - want).abs().max())
        assert d <= 1e-5 * max(1.0, float(want.abs().max())), (b, d)
        # a single-image launch gives the same bits
        o1 = G.empty_act(H, W, 256)
        q1 = L.StmBottleneckParams(xb.img(b).ptr, H, W, Cin, xb.ld, o1.ptr, o1.ld, c1.w_wfrag.data_ptr(), c2.w_wfrag.data_ptr(),
                                   c3.w_wfrag.data_ptr(), c1.w_scale.data_ptr(), c2.w_scale.data_ptr(), c3.w_scale.data_ptr(),
                                   b1.data_ptr(), b2.data_ptr(), b3.data_ptr(), 1, 0, 0, 0)
        L.check(lib.otvm_stm_bottleneck_f16x3(C.byref(q1), G.stream()), "fused bottleneck, one image")
        torch.cuda.synchronize()
        assert torch.equal(o1.torch(), ob.torch(b))
    q.Cin = 128
    assert lib.otvm_stm_bottleneck_f16x3(C.byref(q), G.stream()) != 0          # loud refusal, no fallback


@pytest.mark.parametrize("tile", [1, 2, 3, 0], ids=["8x16", "8x8", "4x8", "auto"])
@pytest.mark.parametrize("H,W", [(5, 7), (8, 16), (9, 33), (19, 45), (34, 60), (68, 120)])
def test_stm_bottleneck128_fused_kernel(G, H, W, tile):
    """otvm_stm_bottleneck_f16x3 with Cin = 512 (ABI 19, csrc/bottleneck128_f16x3.hip): an identity bottleneck of the STM encoders'
    1/8-resolution stage (planes 128: STM.py:43-51,79-87 with torchvision's Bottleneck, BatchNorm folded) as ONE launch ==
    relu(conv3(relu(conv2(relu(conv1 x)))) + x) against fp64 torch, on every pixel tile; sizes that are not multiples of the tiles
    exercise the halo and the edge masks; a batched launch is image-wise identical; all tiles give the same bits."""
    from otvm_amd import lib as L
    from otvm_amd.engine import Act
    lib = L.load()
    B = 2
    xs = [rnd(1, 512, H, W, seed=60 + b).clamp_min(0) * 1.5 for b in range(B)]          # a ReLU output, as in the network
    w1 = rnd(128, 512, 1, 1, seed=1, scale=1.0 / math.sqrt(512))
    w2 = rnd(128, 128, 3, 3, seed=2, scale=1.0 / math.sqrt(128 * 9))
    w3 = rnd(512, 128, 1, 1, seed=3, scale=1.0 / math.sqrt(128))
    sc = [rnd(n, seed=40 + i).abs() + 0.5 for i, n in enumerate((128, 128, 512))]      # folded BatchNorm scales
    bi = [rnd(n, seed=50 + i) * 0.3 for i, n in enumerate((128, 128, 512))]
    c1, c2, c3 = G.pack_weight(w1, scale=sc[0]), G.pack_weight(w2, scale=sc[1]), G.pack_weight(w3, scale=sc[2])
    assert all(c.w_wfrag is not None for c in (c1, c2, c3))
    b1, b2, b3 = (b_.to(G.DEV) for b_ in bi)
    xb = _batched_act(G, xs)
    bs_o = H * W * 512 + 64
    ob = Act(torch.full((B * bs_o + 16,), float("nan"), device=G.DEV), H, W, 512, 512, 0, B=B, bs=bs_o)

    def params(x, o, nb):
        return L.StmBottleneckParams(x.ptr, H, W, 512, x.ld, o.ptr, o.ld, c1.w_wfrag.data_ptr(), c2.w_wfrag.data_ptr(),
                                     c3.w_wfrag.data_ptr(), c1.w_scale.data_ptr(), c2.w_scale.data_ptr(), c3.w_scale.data_ptr(),
                                     b1.data_ptr(), b2.data_ptr(), b3.data_ptr(), nb, x.bs if nb > 1 else 0, o.bs if nb > 1 else 0, tile)
    q = params(xb, ob, B)
    L.check(lib.otvm_stm_bottleneck_f16x3(C.byref(q), G.stream()), "fused bottleneck (planes 128)")
    torch.cuda.synchronize()
    for b in range(B):
        x = xs[b].double()
        t = F.relu(F.conv2d(x, w1.double() * sc[0].double()[:, None, None, None], bi[0].double()))
        t = F.relu(F.conv2d(t, w2.double() * sc[1].double()[:, None, None, None], bi[1].double(), padding=1))
        want = F.relu(F.conv2d(t, w3.double() * sc[2].double()[:, None, None, None], bi[2].double()) + x)
        got = ob.torch(b).permute(2, 0, 1)[None].cpu().double()
        assert torch.isfinite(got).all()
        d = float((got - want).abs().max())
        print("planes-128 fused bottleneck %dx%d tile %d image %d: max-abs vs fp64 %.2e of %.2e" % (H, W, tile, b, d, float(want.abs().max())))
        assert d <= 1e-5 * max(1.0, float(want.abs().max())), (b, d)
        # a single-image launch gives the same bits -- on THIS tile and on the 8x8 tile (the tiles differ in who computes what,
        # not in any summation order)
        for t2_ in (tile, 2):
            o1 = G.empty_act(H, W, 512)
            q1 = params(xb.img(b), o1, 1)
            q1.tile = t2_
            L.check(lib.otvm_stm_bottleneck_f16x3(C.byref(q1), G.stream()), "fused bottleneck (planes 128), one image")
            torch.cuda.synchronize()
            assert torch.equal(o1.torch(), ob.torch(b)), (b, t2_)
    q.tile = 9
    assert lib.otvm_stm_bottleneck_f16x3(C.byref(q), G.stream()) != 0          # loud refusal, no fallback


# ---------------------------------------------------------------------------------------------- table by the last workgroup (ABI 16)
TAIL_CASES = [
    # Cin, Cout, k, stride, H, W, tune code (0 = heuristic)
    (64, 256, 1, 1, 37, 70, 0),                       # implicit GEMM, many tiles
    (256, 64, 1, 1, 40, 56, 0),                       # Cout = 64: two channels per group
    (64, 64, 3, 1, 40, 56, 0),                        # patch kernel
    (24, 64, 7, 2, 64, 96, 0),                        # stem kernel
    (1024, 128, 1, 1, 12, 16, (3 + 1) * 16 + 4),      # K split over 4 workgroups: statistics pass + internal table launch
    (256, 256, 3, 1, 16, 24, (9 + 1) * 16 + 1),       # one-wave tile
    (512, 128, 1, 1, 17, 23, (11 + 1) * 16 + 1),      # pipelined 128x64 tile
    (128, 2048, 1, 1, 33, 31, (0 + 1) * 16 + 1),      # 256x256 tile (no static LDS to spare): 64 channels per group
]


@pytest.mark.parametrize("case", TAIL_CASES, ids=lambda c: "c%d_%d_k%d_s%d_t%d" % (c[0], c[1], c[2], c[3], c[6]))
def test_conv_writes_groupnorm_table_of_its_output(G, case):
    """otvm_conv_params.gn_scale_out: the last workgroup of the conv writes the scale / shift table otvm_gn_table would
    compute from the finished statistics -- same bits, per image, and the ticket counter re-arms itself (second launch)."""
    from otvm_amd import lib as L
    from otvm_amd.engine import Act, conv_params
    lib = L.load()
    Cin, Cout, k, stride, H, W, tune = case
    pad = 3 if k == 7 else (k - 1) // 2
    B = 2
    xs = [rnd(1, Cin, H, W, seed=110 + b) for b in range(B)]
    w = rnd(Cout, Cin, k, k, seed=5, scale=1.0 / math.sqrt(Cin * k * k))
    cw = G.pack_weight(w, i_pad=24 if Cin == 24 else None)
    bias = rnd(Cout, seed=6).to(G.DEV)
    gamma, beta = (rnd(Cout, seed=7).abs() + 0.5).to(G.DEV), (rnd(Cout, seed=8) * 0.2).to(G.DEV)
    Ho, Wo = (H + 2 * pad - (k - 1) - 1) // stride + 1, (W + 2 * pad - (k - 1) - 1) // stride + 1
    xb = _batched_act(G, xs, cw.I_pad)
    ob = Act(torch.zeros(B * (Ho * Wo * Cout + 64) + 16, device=G.DEV), Ho, Wo, Cout, Cout, 0, B=B, bs=Ho * Wo * Cout + 64)
    ws = torch.empty(8 << 20, device=G.DEV)
    counter = torch.zeros(B, dtype=torch.int32, device=G.DEV)
    for rep in range(2):
        stats = torch.zeros(B * 128, dtype=torch.float64, device=G.DEV)
        tab = torch.full((B * (2 * Cout + 8),), float("nan"), device=G.DEV)
        p = conv_params(xb, cw, ob, bias, stride, pad, 1, 0, 0, None, 1, None, ws)
        p.tune = tune
        p.gn_stats, p.gn_bs = stats.data_ptr(), 128
        p.gn_gamma, p.gn_beta = gamma.data_ptr(), beta.data_ptr()
        p.gn_scale_out, p.gn_shift_out = tab.data_ptr(), tab.data_ptr() + 4 * Cout
        p.gn_counter, p.gn_tab_bs = counter.data_ptr(), 2 * Cout + 8
        L.check(lib.otvm_conv2d(C.byref(p), G.stream()), "conv with table")
        torch.cuda.synchronize()
        want = torch.full_like(tab, float("nan"))
        L.check(lib.otvm_gn_table_b(stats.data_ptr(), Ho * Wo, Cout, gamma.data_ptr(), beta.data_ptr(), want.data_ptr(),
                                    want.data_ptr() + 4 * Cout, B, 128, 2 * Cout + 8, G.stream()))
        torch.cuda.synchronize()
        for b in range(B):
            o = b * (2 * Cout + 8)
            assert torch.equal(tab[o:o + 2 * Cout], want[o:o + 2 * Cout]), "image %d, launch %d" % (b, rep)
            assert torch.isfinite(tab[o:o + 2 * Cout]).all()
        assert int(counter.abs().sum()) == 0                                   # re-armed
    # the statistics it was computed from are the conv output's: table == GroupNorm of the output (image 1)
    y = F.conv2d(xs[1], w, bias.cpu(), stride, pad)
    var, mean = torch.var_mean(y.double().reshape(32, -1), dim=1, unbiased=False)
    rstd = (1.0 / torch.sqrt(var + 1e-5)).float().repeat_interleave(Cout // 32)
    o = 2 * Cout + 8
    assert float((tab[o:o + Cout].cpu() - rstd * gamma.cpu()).abs().max()) <= 2e-4 * float((rstd * gamma.cpu()).abs().max())
    # gn_scale_out without the statistics is refused
    p.gn_stats = 0
    assert lib.otvm_conv2d(C.byref(p), G.stream()) != 0
