"""Per-frame timeline from a rocprofv3 kernel trace of bench.py: where the phases of a frame start on the device, how busy
the device is, and how much of the side stream's work (the query encoder) overlaps the previous frame.

    python tools/frame_timeline.py <kernel_trace.csv>
"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    main_stream = collections.Counter(r["Stream_Id"] for r in rows).most_common(1)[0][0]
    # a frame starts with its (main-stream) preprocess launch that writes everything but the query-encoder input
    idx = [i for i, r in enumerate(rows) if "crop_outputs" in r["Kernel_Name"]]
    print("| frame | length ms | device busy ms | sum of kernel time ms | side-stream kernel time ms | memory read start ms | fba_head7 start ms |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for fi in range(12, min(len(idx) - 1, 18)):
        t0, t1 = int(rows[idx[fi]]["End_Timestamp"]), int(rows[idx[fi + 1]]["End_Timestamp"])
        fr = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
        ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in fr)
        busy, (cs, ce) = 0, ev[0]
        for s, e in ev[1:]:
            if s > ce:
                busy += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        busy += ce - cs
        side = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fr if r["Stream_Id"] != main_stream)

        def first(sub):
            for r in fr:
                if sub in r["Kernel_Name"]:
                    return (int(r["Start_Timestamp"]) - t0) / 1e6
            return float("nan")
        print("| %d | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f |" % (fi, (t1 - t0) / 1e6, busy / 1e6,
                                                                 sum(e - s for s, e in ev) / 1e6, side / 1e6,
                                                                 first("memory_read_f16x3_kernel"), first("fba_head")))


if __name__ == "__main__":
    main()
