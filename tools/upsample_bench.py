"""otvm_upsample_bilinear (x2) alone on the device at the frame's three sizes (GPU only): ms and TB/s of algorithmic bytes.

    OTVM_UPSAMPLE2X_ROWS=1 python tools/upsample_bench.py     # one input row per workgroup (the round-4 form)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otvm_amd import lib as L                      # noqa: E402


def main():
    lib = L.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for (Hi, Wi, C, with_add) in ((136, 240, 256, 1), (272, 480, 64, 0), (544, 960, 64, 0), (60, 104, 256, 1), (240, 416, 64, 0)):
        x = torch.randn(Hi * Wi * C, device=dev)
        add = torch.randn(4 * Hi * Wi * C, device=dev) if with_add else None
        out = torch.empty(4 * Hi * Wi * C, device=dev)

        def run():
            L.check(lib.otvm_upsample_bilinear(x.data_ptr(), Hi, Wi, C, C, None, None, 0, add.data_ptr() if with_add else None, C,
                                               out.data_ptr(), 2 * Hi, 2 * Wi, C, st), "upsample")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        by = Hi * Wi * C * 4 * (1 + 4 + (4 if with_add else 0))
        print("upsample2x %4dx%-4d C %3d%s: %7.4f ms  %5.2f TB/s (algorithmic %d MB)" % (Hi, Wi, C, " + add" if with_add else "      ", ms, by / ms / 1e9, by >> 20))


if __name__ == "__main__":
    main()
