"""Every configuration the plan-time autotuner may pick, on the REAL layer shapes of a plan: for each convolution of the
frame (inputs = the activations a frame left in the plan's buffers) run every candidate of otvm_conv2d_candidates and
compare its output with the built-in heuristic's -- all of them compute the same convolution, only the fp32 summation
order may differ.  (tests/test_gpu_kernels.py::test_conv_every_tunable_configuration does this on a handful of shapes.)

    python tools/tune_verify.py [--height 1080 --width 1920] [--tol 2e-4]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--tol", type=float, default=2e-4)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--diag", action="store_true", help="report non-finite inputs / tables of every layer before it is replayed")
    args = ap.parse_args()
    from otvm_amd import lib as L
    from otvm_amd.synth_data import disc_trimap
    os.environ.setdefault("OTVM_AUTOTUNE", "0")
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(dev)
    H, W, T = args.height, args.width, 4
    frames = bench.device_clip(H, W, T, seed=1, dev=dev)
    tri = torch.from_numpy(disc_trimap(H, W))[None, None].to(dev)
    a = torch.ones(1, 1, 1, H, W, device=dev)
    state = {"t": 0}

    def run_frame():
        """One more frame of the clip: every buffer, statistics block and GroupNorm table of the plan is what a frame leaves
        behind.  Called before every layer's sweep: a replayed layer writes its output (and, fused, its output's statistics)
        outside the frame's order -- round 4's block tails (conv3 writes the normalised block output itself) would otherwise
        chain the replays: un-normalised activations grow from block to block and leave fp16's range by layer3.1."""
        t = state["t"]
        state["t"] += 1
        model(a, frames[t % T], frames[t % T], tri=None, tri_gt=tri, large_input=False, **bench.frame_kwargs(t, 1 << 30, 5, 5))
        model._engine.flush()
        torch.cuda.synchronize()

    for _ in range(3):
        run_frame()
    eng = model._engine
    pl = eng.last_plan
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    codes = (C.c_int * 128)()
    bad = n_cfg = 0
    worst, zero_refs = 0.0, 0
    stats_scratch = torch.zeros(64, dtype=torch.float64, device=dev)
    for p, name in pl._convs:
        Ho, Wo, ld = int(p.Ho), int(p.Wo), int(p.out_ld)
        # the plan buffer may be a channel slice of a wider one: rows of out_ld floats, the first Cout columns are compared
        ptr = int(p.out)
        n = int(lib.otvm_conv2d_candidates(C.byref(p), codes, 128))
        cands = [0] + [int(codes[i]) for i in range(n)]
        keep_tune, keep_stats, keep_tab = int(p.tune), p.gn_stats, p.gn_scale_out
        if p.gn_stats:
            # replayed outside its frame the statistics go to a scratch block nobody clears -- and the table of the OUTPUT's
            # GroupNorm (ABI 16: written by the launch's last workgroup FROM those statistics, read by the next layer as its
            # input normalisation) must not be rewritten from them: garbage sums -> NaN tables -> NaN / all-zero layers downstream
            p.gn_stats = stats_scratch.data_ptr()
            p.gn_scale_out = None
        in_place = int(p.out) == int(p.inp) or (p.residual and int(p.residual) == int(p.out))
        if in_place or not ptr:                                  # (a head-carrying conv may not write its hidden state at all)
            p.tune, p.gn_stats, p.gn_scale_out = keep_tune, keep_stats, keep_tab
            continue
        p.tune, p.gn_stats, p.gn_scale_out = keep_tune, keep_stats, keep_tab
        run_frame()
        if p.gn_stats:
            p.gn_stats = stats_scratch.data_ptr()
            p.gn_scale_out = None
        if args.diag:
            xin = _read(int(p.inp), int(p.H) * int(p.W), int(p.in_ld), int(p.Cin), dev)
            msg = []
            if not torch.isfinite(xin).all():
                msg.append("input has %d non-finite values" % int((~torch.isfinite(xin)).sum()))
            for fld, n_ in (("in_scale", int(p.Cin)), ("in_shift", int(p.Cin)), ("w_scale", int(p.Cout)), ("bias", int(p.Cout)),
                            ("res_scale", int(p.Cout))):
                a_ = getattr(p, fld)
                if a_:
                    tab = _read(int(a_), 1, n_, n_, dev)
                    if not torch.isfinite(tab).all():
                        msg.append("%s table non-finite" % fld)
            if p.residual:
                r_ = _read(int(p.residual), Ho * Wo, int(p.res_ld), int(p.Cout), dev)
                if not torch.isfinite(r_).all():
                    msg.append("residual non-finite")
            if msg:
                print("DIAG %-45s %s" % (name, "; ".join(msg)))
        ref = None
        for c in cands:
            for rep in range(args.repeat):
                p.tune = c
                rc = lib.otvm_conv2d(C.byref(p), st)
                if rc != 0:
                    print("FAIL launch", name, c, L.load().otvm_last_error().decode())
                    bad += 1
                    break
                got = _read(ptr, Ho * Wo, ld, int(p.Cout), dev)
                n_cfg += 1
                if ref is None:
                    ref = got
                    scale = max(1.0, float(ref.abs().max()))
                    zero_refs += int(float(ref.abs().max()) == 0.0)
                    continue
                d = float((got - ref).abs().max())
                worst = max(worst, d / scale)
                if args.diag and not torch.isfinite(got).all() and not getattr(main, "_diag_done_" + name.replace(".", "_"), False):
                    setattr(main, "_diag_done_" + name.replace(".", "_"), True)
                    bad_ = ~torch.isfinite(got)
                    xin2 = _read(int(p.inp), int(p.H) * int(p.W), int(p.in_ld), int(p.Cin), dev)
                    ch = bad_.sum(0).nonzero().flatten().tolist()
                    px = bad_.sum(1).nonzero().flatten().tolist()
                    print("DIAG2 %s tune %d rep %d: %d non-finite outputs (ref finite: %s), channels %s..%s (%d), pixels %s..%s (%d); input now has %d "
                          "non-finite, max |input| %.3e; out ptr %#x in ptr %#x in bytes %d out bytes %d; stats %#x in_scale %s w_scale %#x bias %s residual %s"
                          % (name, c, rep, int(bad_.sum()), bool(torch.isfinite(ref).all()), ch[:1], ch[-1:], len(ch), px[:1], px[-1:], len(px),
                             int((~torch.isfinite(xin2)).sum()), float(xin2.abs().max()), int(p.out), int(p.inp), int(p.H) * int(p.W) * int(p.in_ld) * 4,
                             Ho * Wo * ld * 4, int(p.gn_stats or 0), p.in_scale, int(p.w_scale or 0), p.bias, p.residual))
                if not torch.isfinite(got).all() or d > args.tol * scale:
                    print("MISMATCH %-45s tune %3d rep %d: max-abs %.3e (scale %.2e) shape Cin %d Cout %d k%d s%d d%d %dx%d"
                          % (name, c, rep, d, scale, p.Cin, p.Cout, p.kh, p.stride, p.dil, p.H, p.W))
                    bad += 1
        p.tune, p.gn_stats, p.gn_scale_out = keep_tune, keep_stats, keep_tab
    torch.cuda.synchronize()
    print("tune_verify %dx%d: %d layers, %d configuration runs, %d mismatches; worst difference between two configurations "
          "%.2e of the layer's max |output| (0 would mean the check compares nothing), %d layers with an all-zero output"
          % (W, H, len(pl._convs), n_cfg, bad, worst, zero_refs))
    return 1 if bad else 0


def _read(ptr, P, ld, cout, dev):
    """fp32 rows [P][ld] at device address ptr -> a copy of the first cout columns."""
    nbytes = P * ld * 4
    raw = torch.empty(P * ld, dtype=torch.float32, device=dev)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipMemcpy(ctypes.c_void_p(raw.data_ptr()), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 3)   # device to device
    assert rc == 0, rc
    return raw.view(P, ld)[:, :cout].clone()


if __name__ == "__main__":
    sys.exit(main())
