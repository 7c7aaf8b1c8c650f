"""Per-(kernel, grid) HBM-side traffic of a tools/conv_bench.py run from the FETCH_SIZE and WRITE_SIZE PMC passes (separate
rocprofv3 runs with --kernel-trace; KiB -> bytes; FETCH_SIZE x2 on gfx950 -- MI355X_MICROARCH.md).  Launches of one kernel on
different layer shapes differ in their grid size, so every row is one layer.

    python tools/pmc_layer_traffic.py <fetch dir> <write dir> [substring ...]
"""
import collections
import csv
import glob
import sys


def load(d, counter):
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1"))
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], grid, wg, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.OrderedDict()
    seen = set()
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] != counter or r["Dispatch_Id"] not in dur:
            continue
        name, grid, wg, ns = dur[r["Dispatch_Id"]]
        k = (name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60], grid, wg)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[2] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            a[0] += 1
            a[1] += ns
    return agg


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    subs = sys.argv[3:]
    F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    print("| kernel | workgroups | launches | avg us | read MB / launch | written MB / launch | total GB/s |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k in F:
        if subs and not any(s in k[0] for s in subs):
            continue
        n, ns, kib = F[k]
        nw, nsw, kibw = W.get(k, [0, 0.0, 0.0])
        if n == 0 or ns == 0:
            continue
        rd = 2.0 * kib * 1024.0 / n
        wr = kibw * 1024.0 / max(1, nw)
        try:
            wgs = int(k[1]) // max(1, int(k[2]))
        except ValueError:
            wgs = -1
        print("| %s | %d | %d | %.1f | %.1f | %.1f | %.0f |" % (k[0], wgs, n, ns / n / 1e3, rd / 1e6, wr / 1e6, (rd + wr) / (ns / n)))


if __name__ == "__main__":
    main()
