"""Replays bench.py's sequence of calls (timed clip -> instrumented replay -> CPU cross-check) with per-stage differences of
the cross-check frame against the oracle (diagnosis aid).

    python tools/bench_flow_diag.py [--steps 20 --warmup 5] [--replay 1]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--replay", type=int, default=1)
    ap.add_argument("--sync-between", type=int, default=0)
    args = ap.parse_args()
    from oracle.otvm_oracle import OtvmOracle
    from otvm_amd.synth_data import disc_trimap
    from otvm_amd.synth_weights import synthetic_state_dict
    from tests.test_gpu_frame import fmt, stage_report
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    sd = synthetic_state_dict(0)
    H, W, T = args.height, args.width, args.steps + args.warmup
    frames = bench.device_clip(H, W, T, seed=1, dev=dev)
    tri = torch.from_numpy(disc_trimap(H, W))[None, None].to(dev)
    a = torch.ones(1, 1, 1, H, W, device=dev)

    def run_frames(t0, t1, ready=True):
        for t in range(t0, t1):
            out = model(a, frames[t], frames[t], tri=None, tri_gt=tri, large_input=False, _inputs_ready=ready,
                        **bench.frame_kwargs(t, T, 5, 5))
        return out
    run_frames(0, T)
    torch.cuda.synchronize()
    eng = model._engine
    if args.replay:
        eng.prof = []
        run_frames(T - min(args.steps, 10), T)
        torch.cuda.synchronize()
        eng.prof = None
    if args.sync_between:
        torch.cuda.synchronize()
    t_s = 2
    run_frames(0, t_s)
    eng.flush()
    torch.cuda.synchronize()
    pl = eng.last_plan
    hw, h16, w16 = pl.hw, pl.Hp // 16, pl.Wp // 16
    orc = OtvmOracle(sd, dilate_kernel=12)
    orc.bank = [(s["k"].t.reshape(hw, 128).t().reshape(128, h16, w16).cpu().contiguous(),
                 s["v"].t.reshape(hw, 512).t().reshape(512, h16, w16).cpu().contiguous(), s["frame"]) for s in eng.bank]
    print("bank frames", [s["frame"] for s in eng.bank])
    kw = bench.frame_kwargs(t_s, T, 5, 5)
    cap = {}
    ref = orc.frame(a.cpu(), frames[t_s].cpu(), frames[t_s].cpu().clone(), tri_gt=tri.cpu(), frame_id=t_s, capture=cap, **kw)
    hip = model(a, frames[t_s], frames[t_s], tri=None, tri_gt=tri, **kw)
    torch.cuda.synchronize()
    print("alpha max-abs %.3e | %s" % (float((hip[3].cpu() - ref[3]).abs().max()), fmt(stage_report(pl, cap, False))))
    hip2 = model(a, frames[t_s], frames[t_s], tri=None, tri_gt=tri, **kw)      # same frame again (bank now has one more slot)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
