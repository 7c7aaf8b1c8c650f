"""CPU: host-side logic of the product (no GPU compute), the C ABI surface, and the sharded runner."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    """include/otvm_hip.h is the contract: every function it declares must be exported by the .so (and nothing
    the binding uses may be missing from the header)."""
    import __graft_entry__ as g
    path = g.build()
    header = open(os.path.join(ROOT, "include", "otvm_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(otvm_[a-z0-9_]+)\s*\(", header))
    declared -= {"otvm_conv_params", "otvm_preprocess_params"}
    assert len(declared) >= 20
    lib = ctypes.CDLL(path)
    for sym in sorted(declared):
        assert hasattr(lib, sym), "declared in the header but not exported: " + sym
    from otvm_amd import lib as L
    assert set(L.EXPORTED) == declared, (set(L.EXPORTED) ^ declared)
    lib.otvm_abi_version.restype = ctypes.c_int
    raw = open(os.path.join(ROOT, "include", "otvm_hip.h")).read()
    assert lib.otvm_abi_version() == L.ABI_VERSION == int(re.search(r"#define OTVM_ABI_VERSION (\d+)", raw).group(1))


def test_ctypes_struct_matches_c_layout():
    """sizeof(otvm_conv_params) / otvm_preprocess_params / otvm_ppm_head_params as compiled by gcc == the ctypes mirrors."""
    src = ('#include <stdio.h>\n#include "otvm_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(otvm_conv_params), '
           'sizeof(otvm_preprocess_params), sizeof(otvm_ppm_head_params), sizeof(otvm_gn_apply_params), sizeof(otvm_gram_params), '
           'sizeof(otvm_gn_predict_params), sizeof(otvm_stm_bottleneck_params));return 0;}\n')
    exe = os.path.join(ROOT, "otvm_amd", "csrc", "build", "abi_sizes")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    a, b, c, d, e, f, g = (int(v) for v in subprocess.check_output([exe]).split())
    from otvm_amd import lib as L
    assert ctypes.sizeof(L.ConvParams) == a and ctypes.sizeof(L.PreprocessParams) == b and ctypes.sizeof(L.PpmHeadParams) == c
    assert ctypes.sizeof(L.GnApplyParams) == d
    assert ctypes.sizeof(L.GramParams) == e and ctypes.sizeof(L.GnPredictParams) == f      # (ABI 18: + diag)
    assert ctypes.sizeof(L.StmBottleneckParams) == g                                         # (ABI 19: + tile)


def test_bank_policy_engine_equals_oracle():
    from otvm_amd.engine import bank_update as eng_update
    from oracle.otvm_oracle import bank_update as orc_update
    rng = np.random.Generator(np.random.PCG64(0))
    for max_num in (0, 1, 2, 3, 5):
        for skip in (2, 3, 5):
            eb, ob = [], []
            for t in range(40):
                mem = (t % skip == 0) if skip > 2 else False
                slot = dict(frame=t)
                eb, released = eng_update(eb, slot, t == 0, mem, max_num)
                ob = orc_update(ob, (None, None, t), t == 0, mem, max_num)
                assert [s["frame"] for s in eb] == [b[2] for b in ob]
                assert all(not any(r is k for k in eb) for r in released)
                assert len(eb) <= max(1, max_num)


def test_pad_amounts_and_schedule():
    from otvm_amd.engine import pad_amounts
    from otvm_amd.video import memory_schedule
    assert pad_amounts(1080, 1920, 32) == (0, 0, 4, 4)          # 1080p -> 1088, pad (0,0,4,4)
    assert pad_amounts(100, 150, 32) == (5, 5, 14, 14)
    assert pad_amounts(33, 65, 32) == (15, 16, 15, 16)          # odd remainder: extra pixel bottom/right
    # eval.py:184-189: large input doubles the skip and halves the bank
    assert memory_schedule(10, 1080, 1920) == (True, 5, False)
    assert memory_schedule(10, 2160, 3840) == (False, 2, True)
    assert memory_schedule(20, 2160, 3840) == (True, 2, True)
    assert memory_schedule(4, 480, 832, skip=2) == (False, 5, False)


def test_trimap_file_to_onehot():
    from otvm_amd.video import trimap_file_to_onehot
    g = np.array([[0, 128, 254], [254, 0, 128]], np.uint8)      # demo/dove levels {0,128,254}
    oh = trimap_file_to_onehot(g)
    assert oh.shape == (3, 2, 3)
    np.testing.assert_array_equal(oh[0], [[1, 0, 0], [0, 1, 0]])
    np.testing.assert_array_equal(oh[1], [[0, 1, 0], [0, 0, 1]])
    np.testing.assert_array_equal(oh[2], [[0, 0, 1], [1, 0, 0]])
    two = np.array([[0, 255]], np.uint8)                        # two-level quirk (dataset.py:890-893)
    oh = trimap_file_to_onehot(two)
    np.testing.assert_array_equal(oh[1], [[1, 1]])


def test_shard_sequences_partitions():
    from otvm_amd.dist import shard_sequences
    for world in (1, 2, 4, 8):
        for n in (1, 5, 48):
            lengths = [(7 * i) % 13 + 1 for i in range(n)]
            seen = []
            for r in range(world):
                seen += shard_sequences(n, r, world, lengths)
            assert sorted(seen) == list(range(n))
            assert sorted(sum((shard_sequences(n, r, world) for r in range(world)), [])) == list(range(n))
    loads = [sum([10, 9, 8, 1, 1, 1][i] for i in shard_sequences(6, r, 2, [10, 9, 8, 1, 1, 1])) for r in range(2)]
    assert abs(loads[0] - loads[1]) <= 4 and sum(loads) == 30      # LPT greedy: 13 vs 17


def _seq_lengths(n):
    """Ragged clip lengths: 5 sequences = the original case; 48 = BASELINE configs[3] (VideoMatting108 val: 48 clips of unequal
    length -- reference dataset.py:959-1017 -- sharded one sequence per GPU)."""
    return [3 + i for i in range(n)] if n <= 5 else [3 + (7 * i) % 11 for i in range(n)]


def _worker(rank, world, port, q, nseq=5):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from otvm_amd.dist import run_sharded
    seqs = [dict(frames=list(range(n)), id=i) for i, n in enumerate(_seq_lengths(nseq))]

    def matte(seq):                         # stand-in for the GPU path: the sharding/reduction logic is what is tested
        n = len(seq["frames"])
        met = dict(frames=n, sad_sum=1.0 * n, mse_num=2.0 * n, mask_sum=10.0 * n, dt_err2_sum=0.5, dt_mask_sum=10.0 * (n - 1),
                   mse_per_frame=[0.25 * (seq["id"] + 1)] * n, dtssd_per_pair=[0.125] * (n - 1), dtssd_num_per_pair=[2.0] * (n - 1))
        return dict(alpha=torch.full((n, 4, 4), float(seq["id"])), metrics=met)

    def ref(seq):
        return torch.full((len(seq["frames"]), 4, 4), float(seq["id"]) + 0.5)
    out = run_sharded(seqs, matte, rank=rank, world=world, reference_fn=ref)
    q.put((rank, out["frames"], out["sad"], out["max_abs"], out["sequences"], out["gt_metrics"]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nseq", [(2, 5), (2, 48), (8, 48)], ids=["2ranks-5seqs", "2ranks-48seqs", "8ranks-48seqs"])
def test_sharded_runner_ranks_gloo(world, nseq):
    """run_sharded over real process groups (gloo on CPU): 2 ranks on the small case, and BASELINE configs[3]'s shape -- 48 ragged
    sequences -- over 2 and over 8 ranks (the 8-GPU node's rank count; reference eval.py:42,80 runs one device per process).  Every
    rank ends with the same SUM / MAX-reduced totals, the shards partition the sequences, and the longest-first assignment
    keeps the ranks' frame loads within one longest clip of each other."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world * 31 + nseq) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, nseq)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
    lengths = _seq_lengths(nseq)
    total_frames = sum(lengths)
    for rank, frames, sad, max_abs, mine, gm in res:
        assert frames == total_frames                               # SUM all-reduce
        assert abs(sad - 0.5 * 16 * total_frames / 1000.0) < 1e-9
        assert abs(max_abs - 0.5) < 1e-12                           # MAX all-reduce
        # ground-truth metrics: pooled ratios AND the reference's per-frame / per-pair means (utils/tmp/metric.py:184-189,
        # 252-264), reduced over all ranks
        assert gm["frames"] == total_frames and abs(gm["sad"] - 1.0) < 1e-12 and abs(gm["mse"] - 0.2) < 1e-12
        want_mse = sum(0.25 * (i + 1) * n for i, n in enumerate(lengths)) / total_frames
        assert abs(gm["mse_mean"] - want_mse) < 1e-9
        assert abs(gm["dtssd_mean"] - 0.125) < 1e-12 and abs(gm["dtssd_norm_mean"] - 0.0625) < 1e-12
    assert sorted(sum((r[4] for r in res), [])) == list(range(nseq))
    loads = [sum(lengths[i] for i in r[4]) for r in res]
    assert max(loads) - min(loads) <= max(lengths), loads


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under otvm_amd/ may import it (or the reference)."""
    for dp, _, files in os.walk(os.path.join(ROOT, "otvm_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f
                assert "/root/reference" not in src, f


# ---- dataset enumeration / decoding / viz (reference dataset.py:959-1070, eval.py:96-115)

def _write_rgb(path, arr):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


def test_demo_enumeration_follows_reference_layout(tmp_path):
    from otvm_amd.datasets import Demo_Test, load_sequence
    rng = np.random.default_rng(0)
    img = lambda: rng.integers(0, 255, (12, 16, 3), dtype=np.uint8)
    root = str(tmp_path)
    for n in ("0001.png", "0002.png", "0003.png"):
        _write_rgb(os.path.join(root, "seqA", "frames", n), img())
    tri = np.zeros((12, 16), np.uint8); tri[3:9, 4:12] = 128; tri[5:7, 6:10] = 255
    _write_rgb(os.path.join(root, "seqA", "trimap", "0001.png"), tri)
    for n in ("a.png", "b.png"):
        _write_rgb(os.path.join(root, "seqB", "frames", n), img())
    _write_rgb(os.path.join(root, "seqB", "trimap", "b.png"), tri)        # trimap only for the 2nd frame
    items = list(Demo_Test(root))
    assert [it[6] for it in items] == ["seqA", "seqB"] and all(it[0] == "demo" and it[1] == root for it in items)
    assert items[0][2] == [os.path.join("seqA", "frames", n) for n in ("0001.png", "0002.png", "0003.png")]
    # the most recent existing trimap path is carried forward; '' before the first one (dataset.py:1041-1050)
    assert items[0][5] == [os.path.join("seqA", "trimap", "0001.png")] * 3
    assert items[1][5] == ["", os.path.join("seqB", "trimap", "b.png")]
    assert items[0][3] is None and items[0][4] is None
    d = load_sequence(items[0], max_frames=2)
    assert d["names"] == ["0001", "0002"] and len(d["frames"]) == 2 and d["frames"][0].shape == (12, 16, 3)
    # frames come back in cv2 channel order (BGR)
    from PIL import Image
    rgb = np.asarray(Image.open(os.path.join(root, "seqA", "frames", "0001.png")))
    assert np.array_equal(d["frames"][0], rgb[..., ::-1])
    assert tuple(d["trimap"].shape) == (3, 12, 16) and float(d["trimap"].sum()) == 12 * 16
    assert float(d["trimap"][2].sum()) == 2 * 4 and float(d["trimap"][1].sum()) == 6 * 8 - 2 * 4
    assert d["frame_paths"] == [os.path.join(root, "seqA", "frames", n) for n in ("0001.png", "0002.png")]
    assert "frames" not in load_sequence(items[0], decode_frames=False)
    # a clip whose FIRST frame has no trimap cannot be evaluated (the reference would hand cv2.imread('') -> None to the
    # model, dataset.py:879); a later frame's trimap is not silently promoted to frame 0
    with pytest.raises(FileNotFoundError):
        load_sequence(items[1])


def test_trimap_files_decode_like_cv2(tmp_path):
    """dataset.py:879-893 reads the trimap with cv2.IMREAD_UNCHANGED: grayscale {0, mid, max} levels (the shipped
    demo/dove file is {0,128,254}), or a colour-coded file whose R channel (BGR index 2) marks unknown and G marks
    foreground; palette PNGs are expanded to colour by OpenCV."""
    from PIL import Image
    from otvm_amd.datasets import read_trimap_unchanged
    from otvm_amd.video import trimap_file_to_onehot
    un = np.zeros((6, 8), bool); un[1:5, 1:7] = True
    fg = np.zeros((6, 8), bool); fg[2:4, 3:5] = True
    un &= ~fg
    want = np.stack([~(un | fg), un, fg]).astype(np.float32)
    p = str(tmp_path)
    g = np.zeros((6, 8), np.uint8); g[un] = 128; g[fg] = 254                       # the dove levels
    Image.fromarray(g).save(os.path.join(p, "gray.png"))
    rgb = np.zeros((6, 8, 3), np.uint8); rgb[un, 0] = 255; rgb[fg, 1] = 255         # R = unknown, G = foreground
    Image.fromarray(rgb).save(os.path.join(p, "rgb.png"))
    Image.fromarray(rgb).convert("P", palette=Image.ADAPTIVE, colors=4).save(os.path.join(p, "pal.png"))
    Image.fromarray(np.concatenate([rgb, np.full((6, 8, 1), 255, np.uint8)], -1)).save(os.path.join(p, "rgba.png"))
    for name in ("gray.png", "rgb.png", "pal.png", "rgba.png"):
        got = trimap_file_to_onehot(read_trimap_unchanged(os.path.join(p, name)))
        assert np.array_equal(got, want), name
    assert read_trimap_unchanged(os.path.join(p, "gray.png")).ndim == 2
    assert read_trimap_unchanged(os.path.join(p, "rgb.png"))[..., 2].max() == 255      # BGR: R is channel 2


def test_v108_enumeration_and_decoding(tmp_path):
    import json
    from otvm_amd.datasets import VideoMatting108_Test, load_sequence
    rng = np.random.default_rng(1)
    root = os.path.join(str(tmp_path), "VideoMatting108")
    corr = {"vidB/clip0/00002.png": "bg1/0002.jpg", "vidB/clip0/00001.png": "bg1/0001.jpg",
            "vidA/clip1/00001.png": "bg0/0001.jpg", "vidA/clip10/00001.png": "bg2/0001.jpg"}
    os.makedirs(root)
    json.dump(corr, open(os.path.join(root, "frame_corr.json"), "w"))
    open(os.path.join(root, "val_videos.txt"), "w").write("vidB/clip0\nvidA/clip1\n")
    open(os.path.join(root, "val_videos_subset.txt"), "w").write("vidA/clip1\n")
    rgba = {}
    for k in corr:
        a = rng.integers(0, 255, (10, 14, 4), dtype=np.uint8)
        rgba[k] = a
        _write_rgb(os.path.join(root, "FG_done", k), a)
    bgs = {}
    for k, v in corr.items():
        b = rng.integers(0, 255, (10, 14, 3), dtype=np.uint8)
        bgs[v] = b
        # the listed .jpg does not exist for bg1/0002: the loader falls back to .png (dataset.py:896-899)
        _write_rgb(os.path.join(root, "BG_done2", os.path.splitext(v)[0] + ".png"), b)
    ds = VideoMatting108_Test(str(tmp_path), mode="val")
    items = list(ds)
    assert len(ds) == 2 and [it[6] for it in items] == ["vidB/clip0", "vidA/clip1"]      # order of the set file
    assert items[0][0] == "V108" and items[0][1] == root and items[0][4] is None and items[0][5] is None
    # frames of a video: sorted frame_corr keys whose dirname is the video (clip10 must not leak into clip1)
    assert items[0][2] == [os.path.join("FG_done", "vidB/clip0/00001.png"), os.path.join("FG_done", "vidB/clip0/00002.png")]
    assert items[0][3] == [os.path.join("BG_done2", "bg1/0001.jpg"), os.path.join("BG_done2", "bg1/0002.jpg")]
    assert items[1][2] == [os.path.join("FG_done", "vidA/clip1/00001.png")]
    assert len(VideoMatting108_Test(str(tmp_path), use_subset=True)) == 1
    d = load_sequence(items[0])
    src = rgba["vidB/clip0/00001.png"]
    assert np.array_equal(d["frames"][0], src[..., 2::-1]) and np.array_equal(d["gt_alpha_u8"][0], src[..., 3])
    assert np.allclose(d["alphas"][0], src[..., 3].astype(np.float32) / 255.0)
    assert np.array_equal(d["backgrounds"][1], bgs["bg1/0002.jpg"][..., ::-1])


def test_viz_grid_layout_and_rounding():
    """Six half-resolution panels, two per row, padding 2 (torchvision make_grid defaults), x*255+0.5 truncated."""
    from otvm_amd.viz import make_grid_u8, viz_panels
    h, w = 8, 12
    img = torch.rand(1, 1, 3, h, w)
    tri_pred = torch.softmax(torch.randn(1, 1, 3, h, w), 2)
    tri_gt = torch.zeros(1, 1, 3, h, w); tri_gt[:, :, 1] = 1
    alpha = torch.rand(1, 1, 1, h, w)
    gt = torch.rand(1, 1, 1, h, w)
    panels = viz_panels((img, tri_pred, tri_gt, alpha, gt))
    assert tuple(panels.shape) == (6, 3, h // 2, w // 2)
    down = lambda x: torch.nn.functional.interpolate(x.reshape(1, -1, h, w), size=(h // 2, w // 2), mode="bilinear",
                                                     align_corners=False)[0]
    green = torch.zeros(3, h, w); green[1] = 1
    comp = img[0, 0] * alpha[0, 0] + green * (1 - alpha[0, 0])
    want = [img[0, 0], comp, tri_gt[0, 0], gt[0, 0].expand(3, -1, -1), tri_pred[0, 0], alpha[0, 0].expand(3, -1, -1)]
    for k in range(6):
        assert torch.allclose(panels[k], down(want[k]), atol=1e-6)
    g = make_grid_u8(panels, nrow=2)
    ph, pw = h // 2, w // 2
    assert g.shape == (3 * (ph + 2) + 2, 2 * (pw + 2) + 2, 3)
    for k in range(6):
        y, x = (k // 2) * (ph + 2) + 2, (k % 2) * (pw + 2) + 2
        tile = (panels[k] * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
        assert np.array_equal(g[y:y + ph, x:x + pw], tile)
    assert g[:2].max() == 0 and g[:, :2].max() == 0 and g[ph + 2:ph + 4].max() == 0          # padding stays black


def test_bench_self_launch_plumbing():
    """bench.py --gpus N started as a plain process launches its N ranks itself (one process per GPU), runs in-process
    when N == 1 or when a launcher already set RANK, and refuses a node with fewer GPUs instead of reporting n_gpus it
    did not use (`--gpus 8` silently ran one rank)."""
    from otvm_amd.dist import self_launch_command
    assert self_launch_command(1, {}, 1, "bench.py", ["--gpus", "1"]) is None
    assert self_launch_command(8, {"RANK": "3", "WORLD_SIZE": "8"}, 8, "bench.py", []) is None
    cmd = self_launch_command(4, {}, 8, "/x/bench.py", ["--gpus", "4", "--steps", "5"], python="py", port=29999)
    assert cmd == ["py", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                   "--master-port", "29999", "/x/bench.py", "--gpus", "4", "--steps", "5"]
    auto = self_launch_command(2, {}, 2, "bench.py", [])
    assert 1024 < int(auto[auto.index("--master-port") + 1]) < 65536
    with pytest.raises(SystemExit):
        self_launch_command(8, {}, 1, "bench.py", [])
    with pytest.raises(SystemExit):
        self_launch_command(0, {}, 1, "bench.py", [])


def test_bench_refuses_more_gpus_than_visible():
    """End to end on this (GPU-less) container: `python bench.py --gpus 2` exits non-zero with a message, it does not
    fall back to one rank."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a node with fewer than 2 GPUs")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU" in (r.stdout + r.stderr)


def test_viz_panels_match_reference_write_image():
    """SURVEY.md 8f-3 pixel parity: the tensor (and nrow) the reference's write_image (eval.py:96-115) hands to
    torchvision's save_image, captured by tests/golden/make_viz_golden.py from the imported reference, against
    viz.viz_panels on the same inputs; the uint8 grid is then the documented make_grid / save_image arithmetic."""
    from otvm_amd.viz import make_grid_u8, viz_panels
    g = np.load(os.path.join(ROOT, "tests", "golden", "viz.npz"))
    for i in (0, 1):
        out5 = tuple(torch.from_numpy(g["in%d_%s" % (i, k)]) for k in ("imgs", "tri_pred", "tri_gt", "alpha", "gt"))
        panels = viz_panels(out5)
        ref = torch.from_numpy(g["out%d_imgs" % i])
        assert int(g["out%d_nrow" % i]) == 2
        assert panels.shape == ref.shape and float((panels - ref).abs().max()) <= 1e-6
        grid = make_grid_u8(panels, nrow=int(g["out%d_nrow" % i]))
        grid_ref = make_grid_u8(ref, nrow=2)
        assert int(np.abs(grid.astype(np.int32) - grid_ref.astype(np.int32)).max()) <= 1     # 1e-6 around a rounding edge
        assert (grid != grid_ref).mean() < 1e-3


def test_tune_file_is_robust_and_tagged(tmp_path, monkeypatch):
    """OTVM_TUNE_FILE: a truncated or foreign file is ignored with a warning instead of making the package
    unimportable, saves are atomic (temp + rename) and carry the device / ABI they were timed for."""
    import json
    import warnings
    from otvm_amd import engine, lib
    path = str(tmp_path / "tune.json")
    monkeypatch.setattr(engine, "TUNE_FILE", path)
    monkeypatch.setattr(engine, "_TUNE_CACHE", {})
    open(path, "w").write('{"choices": {"[1, 2')                     # truncated by a concurrent writer
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        engine._load_tune_file()
    assert engine._TUNE_CACHE == {} and any("ignoring OTVM_TUNE_FILE" in str(x.message) for x in w)
    engine._TUNE_CACHE[(1, 2, 3)] = 35
    engine._save_tune_file()
    doc = json.load(open(path))
    assert doc["abi"] == lib.ABI_VERSION and "device" in doc and doc["choices"] == {"[1, 2, 3]": 35}
    assert [f for f in os.listdir(str(tmp_path)) if f.endswith(".tmp")] == []
    engine._TUNE_CACHE.clear()
    engine._load_tune_file()
    assert engine._TUNE_CACHE == {(1, 2, 3): 35}
    doc["abi"] = lib.ABI_VERSION - 1                                  # timed for another set of kernels
    json.dump(doc, open(path, "w"))
    engine._TUNE_CACHE.clear()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        engine._load_tune_file()
    assert engine._TUNE_CACHE == {} and any("ignoring OTVM_TUNE_FILE" in str(x.message) for x in w)


def test_rank_affinity_and_graph_policy(monkeypatch):
    """8-rank readiness: each rank gets a contiguous share of the cores; hipGraph replay switches itself
    on when several ranks share the host or the frame is small, unless OTVM_GRAPHS decides."""
    from otvm_amd import dist as D
    from otvm_amd import engine
    calls = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: calls.append(list(cpus)), raising=False)
    cpus = list(range(4, 36))                                        # 32 cores visible to the job
    assert D.pin_rank_affinity(0, 8, cpus) == [4, 5, 6, 7] and D.pin_rank_affinity(7, 8, cpus) == [32, 33, 34, 35]
    assert calls == [[4, 5, 6, 7], [32, 33, 34, 35]]
    assert D.pin_rank_affinity(0, 1, cpus) is None                   # single rank: untouched
    assert D.pin_rank_affinity(3, 64, cpus) is None                  # fewer cores than ranks: untouched
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert engine.graphs_wanted(None, 2176 * 3840) is False and engine.graphs_wanted(None, 1088 * 1920) is True
    assert engine.graphs_wanted(None, 480 * 832) is True
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert engine.graphs_wanted(None, 2176 * 3840) is True
    assert engine.graphs_wanted(False, 480 * 832) is False and engine.graphs_wanted(True, 1088 * 1920) is True
    monkeypatch.delenv("OTVM_DIST_BACKEND", raising=False)
    assert D.dist_backend() == "nccl" and D.reduce_device("cuda:3") == "cuda:3"
    monkeypatch.setenv("OTVM_DIST_BACKEND", "gloo")
    assert D.dist_backend() == "gloo" and D.reduce_device("cuda:3") == "cpu"


def test_batch_groups_and_sharded_runner_with_batches():
    """Lock-step batches (round 3): only sequences of one resolution share a batch, the longest first; run_sharded hands
    groups to matte_batch_fn and single leftovers to matte_fn, and the reduced frame count is unchanged."""
    from otvm_amd.dist import batch_groups, run_sharded
    keys = {0: (480, 832), 1: (480, 832), 2: (1080, 1920), 3: (480, 832), 4: (1080, 1920)}
    assert sorted(batch_groups([0, 1, 2, 3, 4], keys, [5, 9, 3, 7, 8], 2)) == sorted([[1, 3], [0], [4, 2]])
    assert batch_groups([0, 1, 3], keys, [5, 9, 3, 7, 8], 4) == [[1, 3, 0]]
    seqs = [dict(frames=list(range(n)), res=keys[i]) for i, n in enumerate([5, 9, 3, 7, 8])]
    calls = []

    def single(sq):
        calls.append(("single", len(sq["frames"])))
        return dict(alpha=torch.zeros(len(sq["frames"]), 1, 1))

    def batched(group):
        calls.append(("batch", [len(sq["frames"]) for sq in group]))
        return [dict(alpha=torch.zeros(len(sq["frames"]), 1, 1)) for sq in group]
    s = run_sharded(seqs, single, batch=2, matte_batch_fn=batched, key_fn=lambda sq: sq["res"])
    assert s["frames"] == 32 and sorted(s["outputs"]) == [0, 1, 2, 3, 4]
    assert sorted(calls, key=str) == sorted([("batch", [9, 7]), ("single", 5), ("batch", [8, 3])], key=str)


def test_bench_clip_is_baselines_length_whatever_the_step_count():
    """bench.py::clip_plan (round 6): the clip is BASELINE's -- 100 frames at 1920x1080 (configs[2]), 50 at 832x480 (configs[1]), 200
    at 4K (configs[4]) -- the last --steps frames are timed, the frames in front of warm-up are an untimed lead-in; and the memory
    schedule of eval.py:188-189 (memory every 5 of at most 5 slots) then has every timed frame read a full bank."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.clip_plan(1080, 1920, 20, 5) == (100, 75)          # the driver's command
    assert bench.clip_plan(1080, 1920, 80, 5) == (100, 15)          # the default command
    assert bench.clip_plan(1080, 1920, 97, 3) == (100, 0) and bench.clip_plan(1080, 1920, 200, 5) == (205, 0)
    assert bench.clip_plan(480, 832, 47, 3) == (50, 0) and bench.clip_plan(2160, 3840, 197, 3) == (200, 0)
    assert bench.clip_plan(1080, 1920, 8, 3, clip_frames=11) == (11, 0) and bench.clip_plan(720, 1280, 20, 5) == (25, 0)
    # slots read by frame t under the reference's policy (engine.bank_update == alpha/model.py:472-493), memory every 5, max 5
    from otvm_amd.engine import bank_update
    T, lead = bench.clip_plan(1080, 1920, 20, 5)
    bank, reads = [], []
    for t in range(T):
        reads.append(len(bank))
        kw = bench.frame_kwargs(t, T, 5, 5)
        if not kw["last_frame"]:
            bank, _ = bank_update(bank, object(), kw["first_frame"], kw["memorize"], kw["max_memory_num"])
    assert reads[0] == 0 and set(reads[lead + 5:]) == {5} and min(reads[16:]) == 5, reads[:20]


def test_bench_sums_the_conv_rows_of_a_counter_file(tmp_path):
    """bench.py::sum_conv_counter (the parser behind the live `roofline.traffic`): only the named counter, only the kernels the
    plan counts as convolution launches; one row per (dispatch, counter) as rocprofv3 writes them."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    p = tmp_path / "f_counter_collection.csv"
    p.write_text("Correlation_Id,Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n"
                 '1,1,"void (anonymous namespace)::conv_igemm_f16x3_kernel<256, 256>(Conv3Args)",FETCH_SIZE,1000\n'
                 '2,2,"void (anonymous namespace)::conv_patch_f16x3_kernel<8, 64>(PatchArgs)",FETCH_SIZE,500.5\n'
                 '3,3,"gn_apply_kernel(float const*)",FETCH_SIZE,77\n'
                 '4,4,"splitk_finish_kernel(float const*)",WRITE_SIZE,9\n'
                 '5,5,"void stm_bottleneck_f16x3_kernel<256>(BnkArgs)",FETCH_SIZE,10\n')
    assert bench.sum_conv_counter(str(p), "FETCH_SIZE") == (1510.5, 3)
    assert bench.sum_conv_counter(str(p), "WRITE_SIZE") == (9.0, 1)
    assert bench.sum_conv_counter(str(p), "SQ_WAVES") == (0.0, 0)
