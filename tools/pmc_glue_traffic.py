"""HBM-side traffic and achieved GB/s per kernel family from the FETCH_SIZE and WRITE_SIZE PMC passes (separate rocprofv3
runs with --kernel-trace; KiB -> bytes; FETCH_SIZE x2 on gfx950 -- MI355X_MICROARCH.md).  Each pass brings its own kernel
durations: read GB/s = fetched bytes / kernel time of the fetch pass, write GB/s likewise, the table shows their sum.

    python tools/pmc_glue_traffic.py <fetch dir> <write dir> <frames> [substring ...]
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
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])          # launches, ns, KiB
    seen = set()
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] != counter or r["Dispatch_Id"] not in dur:
            continue
        name, ns = dur[r["Dispatch_Id"]]
        k = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:48]
        a = agg[k]
        a[2] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            a[0] += 1
            a[1] += ns
    return agg


def main():
    fd, wd, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
    subs = sys.argv[4:]
    F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    print("| kernel | launches / frame | avg us | read MB / launch | written MB / launch | read GB/s | write GB/s | total GB/s | of 8 TB/s |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for k in sorted(F, key=lambda k: -F[k][1]):
        if subs and not any(s in k for s in subs):
            continue
        n, ns, kib = F[k]
        nw, nsw, kibw = W.get(k, [0, 0.0, 0.0])
        if n == 0 or ns == 0:
            continue
        rd = 2.0 * kib * 1024.0
        wr = kibw * 1024.0
        rg = rd / ns
        wg = wr / nsw if nsw else 0.0
        print("| %s | %.1f | %.1f | %.2f | %.2f | %.0f | %.0f | %.0f | %.2f |" %
              (k, n / frames, ns / n / 1e3, rd / n / 1e6, wr / max(1, nw) / 1e6, rg, wg, rg + wg, (rg + wg) / 8000.0))


if __name__ == "__main__":
    main()
