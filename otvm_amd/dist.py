"""Sequence-sharded multi-GPU evaluation.

Video sequences are independent units (all recurrent state -- the memory bank -- is per sequence,
reference models/alpha/model.py:425-429), frames inside a sequence are strictly sequential.  So the only
parallelism across GPUs is one-sequence-per-GPU: rank r takes sequences r, r+world, ... ; every rank holds a
full weight copy; no tensor ever crosses GPUs.  The reference itself evaluates on a single device
(eval.py:42,80).  The one collective is the final reduction of the metric sums: a SUM all-reduce of
[sum_SAD, frames, seconds] and a MAX all-reduce of [max-abs error, wall seconds] (RCCL over xGMI when the
backend is "nccl"; tens of bytes, latency-bound).
"""
import time

import torch


def self_launch_command(n_gpus, env, device_count, script, argv, python=None, port=None):
    """``bench.py --gpus N`` (or eval_cli) started as a plain process: the command that launches its N ranks, one
    process per GPU (the reference runs one device per process, eval.py:42,80), or None when the caller should run
    in-process (N == 1, or already inside a torch.distributed launch: RANK set).  Raises when the node has fewer than
    N GPUs -- a run must never report n_gpus it did not use."""
    import socket
    import sys
    if n_gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" in env or n_gpus == 1:
        return None
    if device_count < n_gpus:
        raise SystemExit("--gpus %d requested but only %d GPU(s) are visible on this node" % (n_gpus, device_count))
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)


def dist_backend():
    """"nccl" (= RCCL over xGMI on ROCm) unless OTVM_DIST_BACKEND says otherwise.  "gloo" exists for rehearsing the
    multi-rank path where RCCL cannot run: several ranks sharing ONE GPU (RCCL refuses two ranks on a device) or CPU
    tests.  The only collectives of the path are the final metric reductions (tens of bytes), so the backend does not
    matter for throughput."""
    import os
    return os.environ.get("OTVM_DIST_BACKEND", "nccl")


def init_process_group(device=None, **kw):
    import torch.distributed as dist
    backend = dist_backend()
    if backend == "nccl" and device is not None:
        kw.setdefault("device_id", device)
    dist.init_process_group(backend, **kw)
    return backend


def reduce_device(device):
    """Where the metric accumulators live for the all-reduce: the rank's GPU under RCCL, the host under gloo."""
    return device if dist_backend() == "nccl" else "cpu"


def pin_rank_affinity(local_rank, local_world, cpus=None):
    """Give this rank (and every thread it starts afterwards: the IO pipeline's decode / encode pools, torch's intra-op pool)
    its own contiguous share of the node's cores.  Eight ranks of a node each issue ~320 launches per frame from one Python
    thread next to 8 IO threads; unpinned they migrate across sockets and contend for the same cores.  Returns the cpu list
    (None when the platform has no sched_setaffinity or there are fewer cores than ranks)."""
    import os
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    share = len(cpus) // local_world
    if share < 1:
        return None
    mine = cpus[local_rank * share:(local_rank + 1) * share]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    torch.set_num_threads(max(1, min(share, torch.get_num_threads())))
    return mine


def shard_sequences(n_sequences, rank, world, lengths=None):
    """Indices of the sequences rank ``rank`` processes.  With ``lengths`` (frames per sequence) the split is
    longest-first greedy (balanced frame counts); otherwise round-robin."""
    if lengths is None:
        return list(range(rank, n_sequences, world))
    order = sorted(range(n_sequences), key=lambda i: (-lengths[i], i))
    load = [0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += lengths[i]
        if r == rank:
            mine.append(i)
    return sorted(mine)


def reduce_metrics(sums, maxes, device="cpu"):
    """All-reduce metric accumulators across ranks (no-op without an initialised process group)."""
    import torch.distributed as dist
    s = torch.tensor(list(sums), dtype=torch.float64, device=device)
    m = torch.tensor(list(maxes), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return s.tolist(), m.tolist()


def batch_groups(indices, keys, lengths, batch):
    """Group sequence indices into lock-step batches of at most ``batch``: only sequences with the same key (resolution)
    share a batch; inside a key the longest go together (a batch runs as many steps as its longest clip, a clip that has
    ended idles along -- video.run_video_matte_batch)."""
    order = sorted(indices, key=lambda i: (str(keys[i]), -lengths[i], i))
    groups, cur = [], []
    for i in order:
        # (a sequence without a key -- no frames, resolution unknown -- never shares a batch)
        if cur and (keys[i] is None or keys[i] != keys[cur[0]] or len(cur) == batch):
            groups.append(cur)
            cur = []
        cur.append(i)
    if cur:
        groups.append(cur)
    return groups


def planned_batch_shapes(lengths, keys, world, batch):
    """Every (key, group size) that batch_groups produces on ANY rank of a ``world``-rank run: the plans rank 0 has to build
    (and time) before engine.share_tune_cache, so that all ranks launch identical configurations for every batched plan --
    the tuner's signature carries the batch size."""
    shapes = set()
    n = len(lengths)
    for r in range(world):
        mine = shard_sequences(n, r, world, lengths)
        if batch > 1:
            for grp in batch_groups(mine, {i: keys[i] for i in mine}, lengths, batch):
                shapes.add((keys[grp[0]], len(grp)))
        else:
            shapes.update((keys[i], 1) for i in mine)
    return shapes


def default_batch(padded_pixels):
    """Sequences per lock-step launch when the caller does not say (`eval_cli --batch 0`): a throughput run (configs[3]: six
    sequences per GPU) steps 2 clips at 1080p (+5 % aggregate) and 4 at <= 480p (+41 %: the small maps of one 480p frame
    cannot fill 256 CUs); 4K frames fill the chip alone."""
    if padded_pixels <= 640 * 1024:
        return 4
    if padded_pixels <= (1 << 22):
        return 2
    return 1


def run_sharded(sequences, matte_fn, rank=0, world=1, device="cpu", reference_fn=None, batch=1, matte_batch_fn=None, key_fn=None):
    """Matte ``sequences`` (list of dicts with at least 'frames') sharded over ranks.
    batch > 1 with matte_batch_fn(list of sequences) -> list of outputs: this rank's sequences are stepped in lock-step groups
    of up to ``batch`` clips of equal key_fn(seq) (their resolution).

    matte_fn(seq) -> dict(alpha=[T,H,W] tensor)       (the HIP path: video.run_video_matte on this rank's GPU)
    reference_fn(seq) -> [T,H,W] tensor or None         (optional ground truth / oracle alpha for SAD, max-abs)
    Returns the globally reduced summary dict (identical on every rank)."""
    lengths = [len(s["frames"]) for s in sequences]
    mine = shard_sequences(len(sequences), rank, world, lengths)
    sad = frames = 0.0
    maxabs = 0.0
    t0 = time.perf_counter()
    outputs = {}
    if batch > 1 and matte_batch_fn is not None:
        keys = {i: key_fn(sequences[i]) if key_fn is not None else None for i in mine}
        todo = []
        for grp in batch_groups(mine, keys, lengths, batch):
            outs = matte_batch_fn([sequences[i] for i in grp]) if len(grp) > 1 else [matte_fn(sequences[grp[0]])]
            todo += list(zip(grp, outs))
    else:
        todo = ((i, matte_fn(sequences[i])) for i in mine)
    for i, out in todo:
        outputs[i] = out
        frames += len(sequences[i]["frames"])
        if reference_fn is not None:
            ref = reference_fn(sequences[i])
            if ref is not None:
                d = (out["alpha"].float().cpu() - ref.float().cpu()).abs()
                sad += float(d.sum()) / 1000.0                      # utils/tmp/metric.py:177-182
                maxabs = max(maxabs, float(d.max()))
    if torch.cuda.is_available():                               # (device may be "cpu" when the reduction runs over gloo)
        torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    # ground-truth metrics accumulated on the device by video.ClipMetrics (sequences run with gt_alpha_u8)
    # plus the sums of the reference's per-frame values (utils/tmp/metric.py:184-189: MSE = mean over frames of
    # err^2 / (mask_sum + 1); :252-264: dtSSD = per-pair sqrt(err^2) with its own normaliser), so the reduced report
    # carries the numbers the reference's BatchMetric would print, not only pooled ratios
    keys = ("frames", "sad_sum", "mse_num", "mask_sum", "dt_err2_sum", "dt_mask_sum", "mse_frame_sum", "dtssd_pair_sum", "dtssd_norm_sum",
            "pairs")
    clip = [0.0] * len(keys)
    for out in outputs.values():
        m = out.get("metrics") if isinstance(out, dict) else None
        if m:
            m = dict(m)
            m["mse_frame_sum"] = float(sum(m.get("mse_per_frame", [])))
            dt, dn = m.get("dtssd_per_pair", []), m.get("dtssd_num_per_pair", [])
            m["dtssd_pair_sum"] = float(sum(dt))                             # the (error, num) pairs dtSSD returns
            m["dtssd_norm_sum"] = float(sum(e / n for e, n in zip(dt, dn)))
            m["pairs"] = float(len(dt))
            clip = [c + float(m[k]) for c, k in zip(clip, keys)]
    red, (maxabs_g, wall_g) = reduce_metrics([sad, frames, secs] + clip, [maxabs, secs], device)
    sad_g, frames_g, secs_sum = red[:3]
    summary = dict(sad=sad_g, frames=frames_g, gpu_seconds=secs_sum, wall_seconds=wall_g, max_abs=maxabs_g,
                   fps=frames_g / wall_g if wall_g > 0 else 0.0, sequences=mine, outputs=outputs)
    # who matted what (rank -> sequence indices), identical on every rank: the partition is part of the report
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        shards = [None] * dist.get_world_size()
        dist.all_gather_object(shards, mine)
        summary["shards"] = shards
    else:
        summary["shards"] = [mine]
    g = dict(zip(keys, red[3:]))
    if g["frames"] > 0:
        # per-frame means as utils/tmp/metric.py reports them (SAD /1000 per frame; MSE over evaluated pixels)
        summary["gt_metrics"] = dict(frames=g["frames"], sad=g["sad_sum"] / g["frames"],
                                     mse=g["mse_num"] / max(1.0, g["mask_sum"]),                # pooled over all pixels
                                     mse_mean=g["mse_frame_sum"] / g["frames"],                  # reference: per-frame mean
                                     dtssd_mean=g["dtssd_pair_sum"] / max(1.0, g["pairs"]),       # mean of the per-pair errors
                                     dtssd_norm_mean=g["dtssd_norm_sum"] / max(1.0, g["pairs"]),  # ... of error / num
                                     dtssd_sum_err2=g["dt_err2_sum"], dtssd_mask_sum=g["dt_mask_sum"])
    return summary
