"""Effective shader clock per kernel from one rocprofv3 pass:  GRBM_GUI_ACTIVE / kernel duration.

    rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d OUT -- <cmd>
    python tools/clock_probe.py OUT
"""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
            continue
        name, ns = dur[r["Dispatch_Id"]]
        a = agg[name[:90]]
        a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += ns
    for k, (n, cyc, ns) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        if ns > 0:
            print("%-90s n=%4d  avg %.1f us  clock %.2f GHz" % (k, n, ns / n / 1e3, cyc / ns))


if __name__ == "__main__":
    main()
