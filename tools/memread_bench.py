"""Micro-benchmark of otvm_memory_read_f16x3 (tuning aid; GPU only).

    python tools/memread_bench.py [--case T,h,w ...] [--iters 20]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otvm_amd import lib as L                      # noqa: E402

DEFAULT = [(5, 68, 120), (5, 30, 52), (2, 136, 240), (1, 68, 120), (20, 68, 120)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--case", action="append")
    ap.add_argument("--warm", type=int, default=3, help="untimed warm-up reads (0 under a profiler that counts every launch)")
    args = ap.parse_args()
    cases = [tuple(int(v) for v in c.split(",")) for c in args.case] if args.case else DEFAULT
    lib = L.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for (T, h, w) in cases:
        hw = h * w
        q = torch.randn(hw, 128, device=dev)
        slots = []
        for t in range(T):
            k = torch.randn(hw, 128, device=dev)
            v = torch.randn(hw, 512, device=dev)
            sl = torch.zeros(int(lib.otvm_bank_slot_bytes_f16x3(hw)), dtype=torch.uint8, device=dev)
            L.check(lib.otvm_bank_pack_f16x3(k.data_ptr(), v.data_ptr(), hw, sl.data_ptr(), st))
            slots.append(sl)
        sp = (C.c_void_p * T)(*[s.data_ptr() for s in slots])
        out = torch.empty(hw, 512, device=dev)
        ws = torch.empty(int(lib.otvm_memory_read_ws_bytes(hw, T)), dtype=torch.uint8, device=dev)

        def run():
            L.check(lib.otvm_memory_read_f16x3(q.data_ptr(), 128, sp, T, hw, out.data_ptr(), 512, ws.data_ptr(), st))
        for _ in range(args.warm):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        fl = 1280.0 * T * hw * hw
        print("T %3d  %3dx%-3d (hw %6d): %8.3f ms  %7.1f TFLOP/s  MFMA busy %.1f %% of the 2.5 PFLOP/s f16 rate"
              % (T, h, w, hw, ms, fl / ms / 1e9, 3 * fl / ms / 1e9 / 2500 * 100))


if __name__ == "__main__":
    main()
