"""MFMA-pipe utilisation per kernel from one rocprofv3 PMC pass.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d OUT -- <cmd>
    python tools/pmc_mfma.py OUT [substring ...]

SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (32 cycles per v_mfma_f32_32x32x16_f16);
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  busy % = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024); the clock is
GUI_ACTIVE / 8 / duration.  Checked against a known instruction count (memory read at 1080p, T = 5:
3 * 1280 * T * hw^2 / 32768 MFMAs).
"""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    subs = sys.argv[2:]
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    seen = set()
    for r in csv.DictReader(open(cc)):
        if r["Dispatch_Id"] not in dur:
            continue
        name, ns = dur[r["Dispatch_Id"]]
        key = name.replace("(anonymous namespace)::", "")[:80]
        a = agg[key]
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            a[1] += float(r["Counter_Value"])
        elif r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            a[2] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            a[0] += 1; a[3] += ns
    print("| kernel | launches | avg us | clock GHz | MFMA busy % of elapsed cycles | MFMA busy % at the 2.4 GHz peak clock |")
    print("|---|---:|---:|---:|---:|---:|")
    for k, (n, gui, busy, ns) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
        if busy <= 0 or (subs and not any(s in k for s in subs)):
            continue
        cyc = gui / 8.0
        print("| %s | %d | %.1f | %.2f | %.1f | %.1f |" % (k, n, ns / n / 1e3, cyc / ns, 100 * busy / (cyc * 1024),
                                                        100 * busy / (ns * 2.4 * 1024)))


if __name__ == "__main__":
    main()
