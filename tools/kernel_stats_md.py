"""rocprofv3 --kernel-trace --stats (csv) -> the markdown summary kept under profiles/.

    python tools/kernel_stats_md.py <kernel_stats.csv> <frames> "<command line that was profiled>" > profiles/rNN_kernel_stats.md
"""
import csv
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return n[:92]


def group(n):
    if "conv_igemm_f16x3" in n or "conv_igemm_f32" in n:
        return "conv (implicit GEMM)"
    if "conv_patch" in n:
        return "conv (patch)"
    if "stm_bottleneck" in n:
        return "conv (fused bottleneck)"
    if "gn_apply" in n or "gn_stats" in n:
        return "gn_apply"
    if "memory_read" in n or "bank_pack" in n:
        return "memory_read"
    if "edt_" in n or "classify" in n:
        return "edt/trimap encode"
    return "other"


def main():
    path, frames, cmd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# rocprofv3 --kernel-trace --stats -- %s" % cmd)
    print("# MI355X (gfx950), %d frames in the trace (warm-up included); raw CSV next to this file" % frames)
    print("# note: the query encoder of frame t+1 and the early part of the memory read run on side streams (DESIGN.md 4): the sum of kernel times exceeds wall time\n")
    print("| kernel | calls | total ms | avg us | % of kernel time |")
    print("|---|---:|---:|---:|---:|")
    g = {}
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        t = float(r["TotalDurationNs"])
        g[group(r["Name"])] = g.get(group(r["Name"]), 0.0) + t
        if t / tot >= 0.0003:
            print("| %s | %s | %.2f | %.1f | %.2f |" % (short(re.sub(r"\s+", " ", r["Name"])), r["Calls"], t / 1e6,
                                                     float(r["AverageNs"]) / 1e3, 100 * t / tot))
    print("\n## per frame (sum of kernel time / %d frames)\n" % frames)
    for k, v in sorted(g.items(), key=lambda kv: -kv[1]):
        print("- %s: %.2f ms" % (k, v / 1e6 / frames))
    print("- total: %.2f ms of kernel time per frame" % (tot / 1e6 / frames))


if __name__ == "__main__":
    main()
