"""eval.py-shaped command line (reference eval.py:21-94, scripts/eval_s4_demo.sh) on the HIP path.

    python -m otvm_amd.eval_cli --demo --data ./demo --weights weights/s4_OTVM.pth --out ./demo_results [--viz]
    python -m otvm_amd.eval_cli --data <root holding VideoMatting108/> --weights weights/s4_OTVM.pth   # val split
    python -m otvm_amd.eval_cli --demo --data ./demo --synthetic-weights            # plumbing check, no checkpoint

Dataset layouts: otvm_amd/datasets.py (Demo_Test / VideoMatting108_Test, reference dataset.py:959-1070).  Without
--demo the VideoMatting108 validation split is evaluated as eval.py:86-89 does: the first-frame trimap is derived
from the ground-truth alpha with the --trimap dilation, and SAD / MSE / dtSSD against the ground truth are
accumulated on the device and reduced over ranks.  Alpha PNGs are written as trunc(alpha*255) (eval.py:209-217);
--viz adds the six-panel composite frames (eval.py:96-115) and, when ffmpeg exists, the mp4 (eval.py:229-242).
Frame IO runs through otvm_amd/io_pipeline.py: the demo flow decodes ahead in a thread pool and uploads on a copy
stream, both flows download the 8-bit alphas asynchronously and PNG-encode them in a pool (the reference's loop blocks
on .cpu() + cv2.imwrite per frame, eval.py:209-217).
Reproducibility: the first call per resolution times the convolution configurations on the device (a per-process choice
among kernels that differ in fp32 summation order): two runs agree bit for bit only with fixed configurations -- set
OTVM_TUNE_FILE=path (choices are written once and re-used; multi-rank runs share rank 0's choices in any case) or
OTVM_AUTOTUNE=0 (built-in heuristic).  `--batch B` steps B sequences of equal resolution per launch.
Multi-GPU: `--gpus N` starts N ranks (one process per GPU) itself, or launch with torchrun; sequences are sharded
one-per-GPU and the metric sums meet in one all-reduce.
"""
import argparse
import os

import torch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True)
    ap.add_argument("--out", default="./demo_results")
    ap.add_argument("--demo", action="store_true", help="demo layout; default: VideoMatting108 validation split")
    ap.add_argument("--subset", action="store_true", help="VideoMatting108: val_videos_subset.txt")
    ap.add_argument("--viz", action="store_true", help="also write the six-panel composites (eval.py --viz)")
    ap.add_argument("--max-frames", type=int, default=None, help="only the first N frames of every sequence")
    ap.add_argument("--weights", default=None)
    ap.add_argument("--synthetic-weights", action="store_true")
    ap.add_argument("--trimap", default="medium", choices=["narrow", "medium", "wide"])
    ap.add_argument("--skip", type=int, default=10)          # cfg.TEST.MEMORY_SKIP_FRAME (config.py:23)
    ap.add_argument("--max-num", type=int, default=5)        # cfg.TEST.MEMORY_MAX_NUM   (config.py:22)
    ap.add_argument("--precision", default=None, choices=["f32", "f16x3", "f16"],
                    help="f16x3 (default, fp32-class), f32 (exact-fp32 MFMA), f16 (one fp16 MFMA pass: reduced precision, labelled mode)")
    ap.add_argument("--gpus", type=int, default=None, help="start this many ranks, one per GPU (default: the launcher's)")
    ap.add_argument("--sync-io", action="store_true", help="write PNGs synchronously in the frame loop (as eval.py does)")
    ap.add_argument("--batch", type=int, default=None,
                    help="sequences of equal resolution stepped in lock-step per GPU (one launch per layer over the batch; "
                         "frames are decoded up front in this mode).  Default: 1 for a single rank; a multi-rank run "
                         "(--gpus N / torchrun: a throughput run, configs[3]) picks 2 at 1080p and 4 at <= 480p "
                         "(dist.default_batch; measured +5 % / +41 % aggregate)")
    ap.add_argument("--summary-json", default=None, help="rank 0 writes the reduced summary (frames, fps, metrics, shards) here")
    args = ap.parse_args(argv)
    from PIL import Image
    from . import helpers
    from .dist import run_sharded, self_launch_command
    from .io_pipeline import AlphaWriter, FramePrefetcher, _Listish
    from .video import run_video_matte

    if args.gpus is not None and args.gpus > 1 and "RANK" not in os.environ:
        import subprocess
        import sys
        cmd = self_launch_command(args.gpus, os.environ, torch.cuda.device_count(), "-m",
                                  ["otvm_amd.eval_cli"] + list(sys.argv[1:] if argv is None else argv))
        raise SystemExit(subprocess.call(cmd))

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("OTVM_DIST_BACKEND", "nccl") != "nccl":
        local_rank %= max(1, torch.cuda.device_count())     # gloo rehearsal: more ranks than GPUs share the devices round-robin
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    # under a launcher (also with ONE rank: RCCL initialised and used).  A stray RANK variable alone is not a launcher: without
    # the rendezvous variables init_process_group would hang or fail where a plain process ran before
    distributed = world > 1 or all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))
    if distributed:
        # one process per GPU, RCCL for the final metric reduction; OTVM_DIST_BACKEND=gloo rehearses the multi-rank path with
        # several ranks on ONE GPU (RCCL refuses that).  Each rank -- its launch thread and the IO pools it starts -- gets
        # its own share of the node's cores.
        from .dist import init_process_group, pin_rank_affinity
        init_process_group(dev)
        pin_rank_affinity(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    dk = {"narrow": 5, "medium": 12, "wide": 20}[args.trimap]              # eval.py:67-72
    cfg = helpers.default_cfg()
    model = helpers.get_model_alpha(cfg, helpers.get_model_trimap(cfg, "Test", dk), "Test", dk)
    if args.synthetic_weights:
        from .synth_weights import synthetic_state_dict
        model.load_state_dict(synthetic_state_dict(0), strict=True)
    else:
        model.load_state_dict(torch.load(args.weights, map_location="cpu"), strict=True)     # eval.py:77-79
    model.precision = args.precision
    # eval.py:80 wraps the model in nn.DataParallel with ONE visible device (CUDA_VISIBLE_DEVICES, eval.py:42).  Under a
    # launcher every GPU is visible to every rank, so the wrapper is pinned to this rank's device: with the default
    # device_ids it would scatter the frames over all GPUs and its per-call replicas would lose the memory bank.
    model = torch.nn.DataParallel(model.to(dev), device_ids=[dev.index], output_device=dev.index).eval()
    from . import datasets, viz
    ds = datasets.Demo_Test(args.data) if args.demo else datasets.VideoMatting108_Test(args.data, mode="val",
                                                                                       use_subset=args.subset)
    items = list(ds)
    seqs = [dict(name=it[6], frames=it[2][:args.max_frames], item=it) for it in items]   # paths: sharding by length
    root_out = os.path.join(args.out, "alpha", "test", helpers.get_model_name(cfg))

    def resolution(sq):
        if not sq["frames"]:
            return None                                   # an empty sequence has no resolution (and is never batched with another)
        with Image.open(os.path.join(sq["item"][1], sq["frames"][0])) as im:
            return (im.height, im.width)
    keys = [resolution(sq) for sq in seqs]
    if args.batch is None:
        # a multi-rank run is a throughput run: lock-step batches by default, sized by the largest resolution in the set
        from .dist import default_batch
        from .engine import pad_amounts
        args.batch = 1
        # (--viz and --sync-io are single-clip features -- matte_batch writes no viz frames and decodes every clip of a group up
        #  front -- so they keep batch 1 unless --batch is given explicitly)
        if world > 1 and any(k is not None for k in keys) and not (args.viz or args.sync_io):
            def padded(k):
                lw, uw, lh, uh = pad_amounts(k[0], k[1], 32)
                return (k[0] + lh + uh) * (k[1] + lw + uw)
            args.batch = default_batch(max(padded(k) for k in keys if k is not None))
    if args.batch > 1 and (args.viz or args.sync_io):
        raise SystemExit("eval_cli: --batch %d steps clips in lock-step through matte_batch, which implements neither --viz nor "
                         "--sync-io; drop one of them" % args.batch)
    if rank == 0:
        print("eval_cli: %d rank(s), lock-step batch %d per rank" % (world, args.batch))
    if distributed:
        # every rank must launch the same kernel configurations, or a clip's alpha (fp32 summation order) would depend on
        # the rank that got it: rank 0 builds -- and times -- the plans of all resolutions in the data set, the others adopt
        # its choices (engine.share_tune_cache) before they build theirs
        # (the tuner's signature carries the batch size: every (resolution, group size) any rank will step is built here)
        from .dist import planned_batch_shapes
        from .engine import share_tune_cache
        if rank == 0:
            eng = model.module._get_engine()
            shapes = planned_batch_shapes([len(sq["frames"]) for sq in seqs], keys, world, max(1, args.batch))
            for (k, b_) in sorted(s_ for s_ in shapes if s_[0] is not None):
                eng.plan(k[0], k[1], b_)
            torch.cuda.synchronize(dev)
        share_tune_cache(0)

    def matte(seq):
        demo = seq["item"][0] == "demo"
        data = datasets.load_sequence(seq["item"], max_frames=args.max_frames, decode_frames=not demo or args.sync_io)
        outdir = os.path.join(root_out, "pred", seq["name"])
        os.makedirs(outdir, exist_ok=True)
        vizdir = os.path.join(args.out, "viz", "test", helpers.get_model_name(cfg), "viz", seq["name"])
        if args.viz:
            os.makedirs(vizdir, exist_ok=True)
        writer = None if args.sync_io else AlphaWriter(dev, outdir, [n + ".png" for n in data["names"]])

        def save(i, alpha, u8, out):
            if writer is None:
                Image.fromarray(u8.cpu().numpy()).save(os.path.join(outdir, data["names"][i] + ".png"))
            else:
                writer.put(i, u8)
            if args.viz:
                viz.write_viz_frame(os.path.join(vizdir, "f%d.jpg" % i), out)
        if demo:
            if args.sync_io:
                frames, rgb, pre = data["frames"], False, None
            else:                                  # decode ahead + pinned upload on a copy stream (RGB order)
                pre = FramePrefetcher(data["frame_paths"], dev)
                frames, rgb = _Listish(pre), True
            res = run_video_matte(model, frames, trimap=data["trimap"], skip=args.skip, max_num=args.max_num,
                                  on_frame=save, device=dev, frames_are_rgb=rgb, keep_on_device=True)
            if pre is not None:
                pre.close()
        else:
            res = run_video_matte(model, data["frames"], alphas=data["alphas"], backgrounds=data["backgrounds"],
                                  skip=args.skip, max_num=args.max_num, on_frame=save, device=dev,
                                  gt_alpha_u8=data["gt_alpha_u8"], gt_mask="unknown", keep_on_device=True)
        if writer is not None:
            writer.close()
        if args.viz:
            viz.make_viz_video(os.path.join(vizdir, "f%d.jpg"),                          # eval.py:229-242
                               os.path.join(args.out, "viz", "test", helpers.get_model_name(cfg), "viz",
                                            seq["name"].replace("/", "_") + ".mp4"))
        return res
    def matte_batch(group):
        """--batch: the clips of ``group`` (one resolution) in lock-step; PNGs through one asynchronous writer per clip."""
        from .video import run_video_matte_batch
        datas = [datasets.load_sequence(sq["item"], max_frames=args.max_frames, decode_frames=True) for sq in group]
        writers = []
        for sq, data in zip(group, datas):
            outdir = os.path.join(root_out, "pred", sq["name"])
            os.makedirs(outdir, exist_ok=True)
            writers.append(AlphaWriter(dev, outdir, [n + ".png" for n in data["names"]]))
        save = lambda b, i, alpha, u8, out: writers[b].put(i, u8)
        if group[0]["item"][0] == "demo":
            res = run_video_matte_batch(model, [d["frames"] for d in datas], trimaps=[d["trimap"] for d in datas], skip=args.skip,
                                        max_num=args.max_num, on_frame=save, device=dev, keep_on_device=True)
        else:
            res = run_video_matte_batch(model, [d["frames"] for d in datas], alphas=[d["alphas"] for d in datas],
                                        backgrounds=[d["backgrounds"] for d in datas], skip=args.skip, max_num=args.max_num,
                                        on_frame=save, device=dev, keep_on_device=True,
                                        gt_alpha_u8=[d["gt_alpha_u8"] for d in datas], gt_mask="unknown")
        for w in writers:
            w.close()
        return res

    from .dist import reduce_device
    summary = run_sharded(seqs, matte, rank=rank, world=world, device=reduce_device(dev) if distributed else dev,
                          batch=max(1, args.batch), matte_batch_fn=matte_batch, key_fn=resolution)
    if distributed:
        import torch.distributed as dist
        # which kernel configurations this rank launched (fp32 summation orders): identical on all ranks by construction
        # (share_tune_cache), reported so that a run can prove it
        from .engine import kernel_config_digest
        digests = [None] * world
        dist.all_gather_object(digests, kernel_config_digest())
        if len(set(digests)) != 1 and rank == 0:
            import warnings
            warnings.warn("eval_cli: the ranks launched different kernel configurations (tuned choices or OTVM_* switches differ "
                          "between ranks): a clip's last bits depend on the rank that got it.  Digests: %s" % digests)
        summary["tune_digests"] = digests
        summary["batch"] = args.batch
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and args.summary_json:
        import json
        json.dump({k: v for k, v in summary.items() if k != "outputs"}, open(args.summary_json, "w"), indent=0)
    if rank == 0:
        print("done | %d frames | %.2f frames/s over %d GPU(s)" % (summary["frames"], summary["fps"], world))
        if "gt_metrics" in summary:
            g = summary["gt_metrics"]
            print("vs ground truth (unknown band) | SAD/frame %.4f | MSE/frame %.6f (pooled %.6f) | dtSSD/pair %.6f | frames %d"
                  % (g["sad"], g["mse_mean"], g["mse"], g["dtssd_mean"], g["frames"]))
    return summary


if __name__ == "__main__":
    main()
