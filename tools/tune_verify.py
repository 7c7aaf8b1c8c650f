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
    for t in range(3):
        model(a, frames[t], frames[t], tri=None, tri_gt=tri, large_input=False, **bench.frame_kwargs(t, 8, 5, 5))
    torch.cuda.synchronize()
    eng = model._engine
    eng.flush()
    pl = eng.last_plan
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    codes = (C.c_int * 64)()
    bad = n_cfg = 0
    worst, zero_refs = 0.0, 0
    stats_scratch = torch.zeros(64, dtype=torch.float64, device=dev)
    for p, name in pl._convs:
        Ho, Wo, ld = int(p.Ho), int(p.Wo), int(p.out_ld)
        # the plan buffer may be a channel slice of a wider one: rows of out_ld floats, the first Cout columns are compared
        ptr = int(p.out)
        n = int(lib.otvm_conv2d_candidates(C.byref(p), codes, 64))
        cands = [0] + [int(codes[i]) for i in range(n)]
        keep_tune, keep_stats = int(p.tune), p.gn_stats
        if p.gn_stats:
            p.gn_stats = stats_scratch.data_ptr()
        in_place = int(p.out) == int(p.inp) or (p.residual and int(p.residual) == int(p.out))
        if in_place:
            p.tune, p.gn_stats = keep_tune, keep_stats
            continue
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
                if not torch.isfinite(got).all() or d > args.tol * scale:
                    print("MISMATCH %-45s tune %3d rep %d: max-abs %.3e (scale %.2e) shape Cin %d Cout %d k%d s%d d%d %dx%d"
                          % (name, c, rep, d, scale, p.Cin, p.Cout, p.kh, p.stride, p.dil, p.H, p.W))
                    bad += 1
        p.tune, p.gn_stats = keep_tune, keep_stats
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
