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


def run_sharded(sequences, matte_fn, rank=0, world=1, device="cpu", reference_fn=None):
    """Matte ``sequences`` (list of dicts with at least 'frames') sharded over ranks.

    matte_fn(seq) -> dict(alpha=[T,H,W] tensor)       (the HIP path: video.run_video_matte on this rank's GPU)
    reference_fn(seq) -> [T,H,W] tensor or None         (optional ground truth / oracle alpha for SAD, max-abs)
    Returns the globally reduced summary dict (identical on every rank)."""
    lengths = [len(s["frames"]) for s in sequences]
    mine = shard_sequences(len(sequences), rank, world, lengths)
    sad = frames = 0.0
    maxabs = 0.0
    t0 = time.perf_counter()
    outputs = {}
    for i in mine:
        out = matte_fn(sequences[i])
        outputs[i] = out
        frames += len(sequences[i]["frames"])
        if reference_fn is not None:
            ref = reference_fn(sequences[i])
            if ref is not None:
                d = (out["alpha"].float().cpu() - ref.float().cpu()).abs()
                sad += float(d.sum()) / 1000.0                      # utils/tmp/metric.py:177-182
                maxabs = max(maxabs, float(d.max()))
    if torch.cuda.is_available() and str(device).startswith("cuda"):
        torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    (sad_g, frames_g, secs_sum), (maxabs_g, wall_g) = reduce_metrics([sad, frames, secs], [maxabs, secs], device)
    return dict(sad=sad_g, frames=frames_g, gpu_seconds=secs_sum, wall_seconds=wall_g, max_abs=maxabs_g,
                fps=frames_g / wall_g if wall_g > 0 else 0.0, sequences=mine, outputs=outputs)
