"""GPU (-m gpu): the sharded multi-rank path EXECUTED on a GPU (BASELINE configs[3] in miniature).

configs[3] shards VideoMatting108's 48 validation sequences over the 8 GPUs of a node, one process per GPU, RCCL for the
final metric all-reduce (otvm_amd/dist.py, reference eval.py:42,80 runs one device per process).  The builder's GPU box
has ONE GPU and RCCL refuses two ranks on a device, so the same code path is rehearsed with the collective backend
switched to gloo (OTVM_DIST_BACKEND=gloo: the only collectives are the final metric reductions and the tune-cache
broadcast) and both ranks on cuda:0: `torch.distributed.run --nproc-per-node 2 -m otvm_amd.eval_cli` over a small
V108-layout tree.  Checked: the shard partition (longest-first greedy), every rank's PNGs are byte-for-byte the PNGs of a
single-rank run of the same command, and the reduced ground-truth metrics equal the single-rank ones.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _v108_tree(root, lengths, H=64, W=96):
    from PIL import Image
    from otvm_amd.synth_data import soft_alpha, synthetic_clip
    v = os.path.join(root, "VideoMatting108")
    os.makedirs(v)
    corr, names = {}, []
    for i, T in enumerate(lengths):
        fg, _ = synthetic_clip(H, W, T, seed=100 + i)
        bg, _ = synthetic_clip(H, W, T, seed=200 + i)
        clip = "vid%d/clip_0" % i
        names.append(clip)
        for t in range(T):
            a = np.rint(soft_alpha(H, W, t + i) * 255).astype(np.uint8)
            k = "%s/%05d.png" % (clip, t)
            corr[k] = "bgs%d/%05d.jpg" % (i, t)
            os.makedirs(os.path.dirname(os.path.join(v, "FG_done", k)), exist_ok=True)
            Image.fromarray(np.concatenate([fg[t][..., ::-1], a[..., None]], -1)).save(os.path.join(v, "FG_done", k))
            os.makedirs(os.path.join(v, "BG_done2", "bgs%d" % i), exist_ok=True)
            Image.fromarray(bg[t][..., ::-1].copy()).save(os.path.join(v, "BG_done2", "bgs%d" % i, "%05d.png" % t))
    json.dump(corr, open(os.path.join(v, "frame_corr.json"), "w"))
    open(os.path.join(v, "val_videos.txt"), "w").write("\n".join(names) + "\n")
    return names


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_on_one_gpu_through_eval_cli(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image
    from otvm_amd.dist import shard_sequences
    lengths = [4, 2, 3]
    root = os.path.join(str(tmp_path), "data")
    os.makedirs(root)
    names = _v108_tree(root, lengths)
    env = dict(os.environ)
    env["OTVM_TUNE_FILE"] = os.path.join(str(tmp_path), "tune.json")   # both runs launch the same kernel configurations
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    common = ["--data", root, "--synthetic-weights", "--skip", "3", "--trimap", "narrow", "--batch", "1"]
    out1, out2 = os.path.join(str(tmp_path), "out1"), os.path.join(str(tmp_path), "out2")
    j1, j2 = os.path.join(str(tmp_path), "s1.json"), os.path.join(str(tmp_path), "s2.json")
    r = subprocess.run([sys.executable, "-m", "otvm_amd.eval_cli"] + common + ["--out", out1, "--summary-json", j1],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    env2 = dict(env)
    env2["OTVM_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), "-m", "otvm_amd.eval_cli"] + common +
                       ["--out", out2, "--summary-json", j2], cwd=ROOT, env=env2, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    s1, s2 = json.load(open(j1)), json.load(open(j2))
    # the partition: longest-first greedy over frame counts (dist.shard_sequences), every sequence exactly once
    assert s2["shards"] == [shard_sequences(3, 0, 2, lengths), shard_sequences(3, 1, 2, lengths)] == [[0], [1, 2]]
    assert s1["shards"] == [[0, 1, 2]]
    assert s1["frames"] == s2["frames"] == sum(lengths)
    # every rank's PNGs == the single-rank run's
    for clip, T in zip(names, lengths):
        for t in range(T):
            rel = os.path.join("alpha", "test", "s4_OTVM", "pred", clip, "%05d.png" % t)
            a1, a2 = np.asarray(Image.open(os.path.join(out1, rel))), np.asarray(Image.open(os.path.join(out2, rel)))
            assert a1.shape == (64, 96) and np.array_equal(a1, a2), rel
    # reduced ground-truth metrics (integer-exact fp64 sums on the device, SUM all-reduce over the ranks) == single rank
    g1, g2 = s1["gt_metrics"], s2["gt_metrics"]
    assert g1["frames"] == g2["frames"] == sum(lengths)
    for k in ("sad", "mse", "mse_mean", "dtssd_mean", "dtssd_sum_err2", "dtssd_mask_sum"):
        assert abs(g1[k] - g2[k]) <= 1e-12 * max(1.0, abs(g1[k])), (k, g1[k], g2[k])
    assert s2["fps"] > 0
    assert len(set(s2["tune_digests"])) == 1 and s2["batch"] == 1
    # ---- the same two ranks with lock-step batches (the default of a multi-rank run; here forced to 2): rank 1 steps its two
    # clips through a (64, 96, 2) plan, whose convolution configurations are timed per batch size.  Rank 0 -- which steps ONE
    # clip -- must have built and timed that plan too before the ranks adopted its choices: the digests of the
    # configurations each rank launched are equal, and the PNGs stay within one 8-bit step of the single-rank run (another
    # tile = another fp32 summation order, nothing else).
    out3, j3 = os.path.join(str(tmp_path), "out3"), os.path.join(str(tmp_path), "s3.json")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), "-m", "otvm_amd.eval_cli"] + common[:-2] +
                       ["--batch", "2", "--out", out3, "--summary-json", j3], cwd=ROOT, env=env2, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    s3 = json.load(open(j3))
    assert s3["batch"] == 2 and len(s3["tune_digests"]) == 2 and len(set(s3["tune_digests"])) == 1, s3["tune_digests"]
    assert s3["frames"] == sum(lengths)
    for clip, T in zip(names, lengths):
        for t in range(T):
            rel = os.path.join("alpha", "test", "s4_OTVM", "pred", clip, "%05d.png" % t)
            a1, a3 = np.asarray(Image.open(os.path.join(out1, rel))), np.asarray(Image.open(os.path.join(out3, rel)))
            assert int(np.abs(a1.astype(np.int16) - a3.astype(np.int16)).max()) <= 1, rel


def test_bench_measures_conv_traffic_live(tmp_path):
    """bench.py's roofline leg measures `traffic` itself (round 5): two short child runs of the same command under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` after the timed region.  Where the profiler is usable the line says "live" and
    the bytes per conv launch sit between the algorithmic bytes and twice that; where it is not, the line must say so (a labelled
    fallback, never a crash, never a silent constant)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "OTVM_BENCH_LIVE_PMC"):
        env.pop(k, None)
    env["OTVM_TUNE_FILE"] = os.path.join(str(tmp_path), "tune.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                        "--height", "480", "--width", "832"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    # round 6: the clip is BASELINE's length whatever --steps says (50 frames at 832x480): an untimed lead-in in front, the LAST
    # `steps` frames timed -- all of them read a full bank
    assert line["lead_in_frames"] == 50 - 4 - 2 and "T=50" in line["config"]["workload"], line["config"]["workload"]
    assert line["config"]["T_read_timed_frames"]["histogram"] == {"5": 4}, line["config"]["T_read_timed_frames"]
    assert line["conv_calls_total"] > 0 and line["memory_read_calls_total"] == 49
    roof = line["roofline"]
    src = roof["traffic_source"]
    if src is not None and src.startswith("live"):
        ratio = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
        assert 0.9 < ratio < 2.0, (ratio, src)
    else:
        assert src is not None and "not available" in src, src


def test_rccl_with_one_rank_through_bench_and_eval_cli(tmp_path):
    """RCCL itself (backend "nccl", the default) EXECUTED: a one-rank torch.distributed launch is legal on one GPU.  bench.py
    and eval_cli then initialise the process group on RCCL with the rank's device, broadcast the tuned configurations
    (share_tune_cache), count the ranks with an all-reduce of ones (`ranks_seen`) and reduce the metric sums with the SUM / MAX
    all-reduces of dist.reduce_metrics -- the code path `bench.py --gpus 8` / `eval_cli --gpus 8` takes on a node
    (reference: one device per process, eval.py:42,80)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.pop("OTVM_DIST_BACKEND", None)                      # the default: nccl = RCCL
    env["OTVM_TUNE_FILE"] = os.path.join(str(tmp_path), "tune.json")
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port"]
    r = subprocess.run(launch + [str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                                 "--no-cpu-baseline", "--no-roofline", "--height", "480", "--width", "832"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["steps"] == 3
    assert res["dist_backend"] == "nccl" and res["ranks_seen"] == 1, (res["dist_backend"], res["ranks_seen"])
    assert len(res["per_rank"]) == 1
    root = os.path.join(str(tmp_path), "data")
    os.makedirs(root)
    lengths = [3, 2]
    _v108_tree(root, lengths)
    out, js = os.path.join(str(tmp_path), "out"), os.path.join(str(tmp_path), "s.json")
    r = subprocess.run(launch + [str(_free_port()), "-m", "otvm_amd.eval_cli", "--data", root, "--synthetic-weights", "--skip", "3",
                                 "--trimap", "narrow", "--out", out, "--summary-json", js],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    s = json.load(open(js))
    assert s["shards"] == [[0, 1]] and s["frames"] == sum(lengths) and len(s["tune_digests"]) == 1
    assert s["gt_metrics"]["frames"] == sum(lengths)


def test_rccl_two_ranks_two_gpus(tmp_path):
    """Arms itself on a box with at least two GPUs (the builder's and the driver's test boxes have one: SKIPPED there, visibly).
    `bench.py --gpus 2` and `eval_cli --gpus 2` on the default backend (nccl = RCCL over xGMI), one rank per GPU -- the launch
    the driver's scaling run uses (reference: one device per process, eval.py:42,80).  Checked: both ranks took part and sat
    on distinct devices, every rank launched the same kernel configurations (equal tune digests), every PNG is byte-for-byte
    the PNG of the single-rank run, the RCCL-reduced ground-truth metrics equal the single-rank sums."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (found %d): the RCCL two-rank path is rehearsed on gloo by "
                    "test_two_ranks_on_one_gpu_through_eval_cli" % (torch.cuda.device_count() if torch.cuda.is_available() else 0))
    from PIL import Image
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.pop("OTVM_DIST_BACKEND", None)                      # the default: nccl = RCCL
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OTVM_TUNE_FILE"] = os.path.join(str(tmp_path), "tune.json")
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline", "--no-roofline", "--height", "480", "--width", "832"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["dist_backend"] == "nccl" and res["scaling"] == "weak"
    assert len(res["per_rank"]) == 2 and len({row["device"] for row in res["per_rank"]}) == 2, res["per_rank"]
    assert res["value"] > 0 and abs(res["value"] - 2 * 3 / (res["ms_per_step"] * 3 / 1000.0)) <= 1e-6 * res["value"]
    lengths = [4, 2, 3]
    root = os.path.join(str(tmp_path), "data")
    os.makedirs(root)
    names = _v108_tree(root, lengths)
    common = ["--data", root, "--synthetic-weights", "--skip", "3", "--trimap", "narrow", "--batch", "1"]
    out1, out2 = os.path.join(str(tmp_path), "out1"), os.path.join(str(tmp_path), "out2")
    j1, j2 = os.path.join(str(tmp_path), "s1.json"), os.path.join(str(tmp_path), "s2.json")
    r = subprocess.run([sys.executable, "-m", "otvm_amd.eval_cli"] + common + ["--out", out1, "--summary-json", j1],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, "-m", "otvm_amd.eval_cli", "--gpus", "2"] + common + ["--out", out2, "--summary-json", j2],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    s1, s2 = json.load(open(j1)), json.load(open(j2))
    assert s2["shards"] == [[0], [1, 2]] and s1["shards"] == [[0, 1, 2]] and s1["frames"] == s2["frames"] == sum(lengths)
    assert len(s2["tune_digests"]) == 2 and len(set(s2["tune_digests"])) == 1, s2["tune_digests"]
    for clip, T in zip(names, lengths):
        for t in range(T):
            rel = os.path.join("alpha", "test", "s4_OTVM", "pred", clip, "%05d.png" % t)
            a1, a2 = np.asarray(Image.open(os.path.join(out1, rel))), np.asarray(Image.open(os.path.join(out2, rel)))
            assert np.array_equal(a1, a2), rel
    g1, g2 = s1["gt_metrics"], s2["gt_metrics"]
    assert g1["frames"] == g2["frames"] == sum(lengths)
    for k in ("sad", "mse", "mse_mean", "dtssd_mean", "dtssd_sum_err2", "dtssd_mask_sum"):
        assert abs(g1[k] - g2[k]) <= 1e-12 * max(1.0, abs(g1[k])), (k, g1[k], g2[k])
