"""Per-kernel sums of rocprofv3 PMC counters (one or more counter_collection.csv files) as a markdown table.

    python tools/pmc_table.py <dir-or-csv> [<dir-or-csv> ...] [--top 16]
"""
import collections
import csv
import glob
import os
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    top = 16
    if "--top" in sys.argv:
        top = int(sys.argv[sys.argv.index("--top") + 1])
        args = [a for a in args if a != str(top)]
    vals = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(int))
    counters = []
    for a in args:
        paths = [a] if a.endswith(".csv") else glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
        for p in paths:
            for r in csv.DictReader(open(p)):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:64]
                c = r["Counter_Name"]
                if c not in counters:
                    counters.append(c)
                vals[k][c] += float(r["Counter_Value"])
                calls[k][c] += 1
    key = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in counters else counters[0]
    order = sorted(vals, key=lambda k: -vals[k].get(key, 0.0))[:top]
    print("| kernel | launches | " + " | ".join(counters) + " |")
    print("|---|---:|" + "---:|" * len(counters))
    for k in order:
        n = max(calls[k].values())
        print("| %s | %d | " % (k, n) + " | ".join("%.3g" % (vals[k].get(c, 0.0) / max(1, calls[k].get(c, 1))) for c in counters) + " |")


if __name__ == "__main__":
    main()
