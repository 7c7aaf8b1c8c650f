"""One frame of a rocprofv3 kernel trace of bench.py as a per-stream listing (start offset, duration, stream, kernel), to
see which launches of the side streams actually overlap which launches of the main stream.

    python tools/frame_trace_dump.py <kernel_trace.csv> [frame index = 13] [min listed us = 40]
"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    fi = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    streams = [s for s, _ in collections.Counter(r["Stream_Id"] for r in rows).most_common()]
    tag = {s: "M S1 S2 S3 S4".split()[i] if i < 5 else "S?" for i, s in enumerate(streams)}
    idx = [i for i, r in enumerate(rows) if "crop_outputs" in r["Kernel_Name"]]
    t0, t1 = int(rows[idx[fi]]["End_Timestamp"]), int(rows[idx[fi + 1]]["End_Timestamp"])
    fr = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
    print("frame %d: %.3f ms, %d launches" % (fi, (t1 - t0) / 1e6, len(fr)))
    for s in streams:
        q = [r for r in fr if r["Stream_Id"] == s]
        if q:
            print("  stream %s: %d launches, %.3f ms kernel time, active %.3f .. %.3f ms" %
                  (tag[s], len(q), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in q) / 1e6,
                   (int(q[0]["Start_Timestamp"]) - t0) / 1e6, (max(int(r["End_Timestamp"]) for r in q) - t0) / 1e6))
    small = collections.defaultdict(lambda: [0, 0.0])
    for r in fr:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        us = (en - st) / 1e3
        if us < min_us:
            small[tag[r["Stream_Id"]]][0] += 1
            small[tag[r["Stream_Id"]]][1] += us
            continue
        if any(v[0] for v in small.values()):
            print("      ... " + ", ".join("%s: %d small launches %.0f us" % (k, v[0], v[1]) for k, v in sorted(small.items()) if v[0]))
            small.clear()
        print("%8.3f %7.1f us  %-2s  grid %-8s %s" % ((st - t0) / 1e6, us, tag[r["Stream_Id"]], r.get("Grid_Size_X", r.get("Grid_Size", "")),
                                                 short(r["Kernel_Name"])))
    # concurrency: wall time with 1, 2, 3 streams active
    ev = []
    for r in fr:
        ev.append((int(r["Start_Timestamp"]), 1))
        ev.append((int(r["End_Timestamp"]), -1))
    ev.sort()
    hist, cur, last = collections.Counter(), 0, t0
    for t, d in ev:
        hist[min(cur, 3)] += t - last
        cur, last = cur + d, t
    print("wall time by number of kernels in flight: " + ", ".join("%d: %.3f ms" % (k, v / 1e6) for k, v in sorted(hist.items())))


if __name__ == "__main__":
    main()
