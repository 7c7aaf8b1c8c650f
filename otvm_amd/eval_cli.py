"""eval.py-shaped command line (reference eval.py:21-94, scripts/eval_s4_demo.sh) on the HIP path.

    python -m otvm_amd.eval_cli --demo --data ./demo --weights weights/s4_OTVM.pth --out ./demo_results
    python -m otvm_amd.eval_cli --demo --data ./demo --synthetic-weights            # plumbing check, no checkpoint

Demo layout (reference dataset.py:1019-1070): <data>/<seq>/frames/*.jpg and <data>/<seq>/trimap/<first>.png.
Images are read with PIL (RGB) and handed over as RGB; alpha PNGs are written as trunc(alpha*255)
(eval.py:209-217).  With torch.distributed initialised (torchrun) sequences are sharded one-per-GPU.
"""
import argparse
import os

import numpy as np
import torch


def list_demo(data_root):
    seqs = []
    for v in sorted(os.listdir(data_root)):
        fdir = os.path.join(data_root, v, "frames")
        if not os.path.isdir(fdir):
            continue
        names = sorted(os.listdir(fdir))
        tri = None
        for n in names:
            p = os.path.join(data_root, v, "trimap", os.path.splitext(n)[0] + ".png")
            if os.path.isfile(p):
                tri = p
                break
        seqs.append(dict(name=v, frames=[os.path.join(fdir, n) for n in names], trimap=tri))
    return seqs


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True)
    ap.add_argument("--out", default="./demo_results")
    ap.add_argument("--demo", action="store_true", help="demo layout (the only dataset layout implemented)")
    ap.add_argument("--weights", default=None)
    ap.add_argument("--synthetic-weights", action="store_true")
    ap.add_argument("--trimap", default="medium", choices=["narrow", "medium", "wide"])
    ap.add_argument("--skip", type=int, default=10)          # cfg.TEST.MEMORY_SKIP_FRAME (config.py:23)
    ap.add_argument("--max-num", type=int, default=5)        # cfg.TEST.MEMORY_MAX_NUM   (config.py:22)
    ap.add_argument("--precision", default=None, choices=["f32", "f16x3"])
    args = ap.parse_args(argv)
    from PIL import Image
    from . import helpers
    from .dist import run_sharded
    from .video import run_video_matte, trimap_file_to_onehot

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    dk = {"narrow": 5, "medium": 12, "wide": 20}[args.trimap]              # eval.py:67-72
    cfg = helpers.default_cfg()
    model = helpers.get_model_alpha(cfg, helpers.get_model_trimap(cfg, "Test", dk), "Test", dk)
    if args.synthetic_weights:
        from .synth_weights import synthetic_state_dict
        model.load_state_dict(synthetic_state_dict(0), strict=True)
    else:
        model.load_state_dict(torch.load(args.weights, map_location="cpu"), strict=True)     # eval.py:77-79
    model.precision = args.precision
    model = torch.nn.DataParallel(model.to(dev)).eval()                                        # eval.py:80
    seqs = list_demo(args.data)

    def matte(seq):
        frames = [np.asarray(Image.open(p).convert("RGB")) for p in seq["frames"]]
        tri = trimap_file_to_onehot(np.asarray(Image.open(seq["trimap"])))
        outdir = os.path.join(args.out, "alpha", "test", helpers.get_model_name(cfg), "pred", seq["name"])
        os.makedirs(outdir, exist_ok=True)

        def save(i, alpha, u8, out):
            name = os.path.splitext(os.path.basename(seq["frames"][i]))[0] + ".png"
            Image.fromarray(u8.cpu().numpy()).save(os.path.join(outdir, name))
        return run_video_matte(model, frames, trimap=tri, skip=args.skip, max_num=args.max_num, frames_are_rgb=True,
                               on_frame=save, device=dev)
    summary = run_sharded(seqs, matte, rank=rank, world=world, device=dev)
    if rank == 0:
        print("done | %d frames | %.2f frames/s over %d GPU(s)" % (summary["frames"], summary["fps"], world))


if __name__ == "__main__":
    main()
