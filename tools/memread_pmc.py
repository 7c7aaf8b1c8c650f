"""Memory read on a growing bank: HBM traffic (FETCH_SIZE / WRITE_SIZE) and MFMA-busy per read, from three rocprofv3 PMC
passes of `tools/memread_bench.py --case T,h,w` (one invocation per bank size; counters in separate runs with
--kernel-trace only, as MI355X_MICROARCH.md prescribes; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE x2 on gfx950).

    python tools/memread_pmc.py OUTDIR T h w reads      (OUTDIR holds mfma/ fetch/ write/ from tools/memread_pmc.sh)

Prints one markdown row: slots, bank bytes (the algorithmic read: every slot once), kernel time per read, TFLOP/s,
MFMA busy, fetched bytes per read, the ratio to the bank size and the resulting GB/s.
"""
import csv
import glob
import sys


def load(d, counter_names):
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        if "memory_read_f16x3_kernel" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = {c: 0.0 for c in counter_names}
    for r in csv.DictReader(open(cc)):
        if r["Dispatch_Id"] in dur and r["Counter_Name"] in tot:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
    return sum(dur.values()), len(dur), tot


def main():
    out, T, h, w, reads = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    hw = h * w
    hp = (hw + 63) // 64 * 64
    bank = T * hp * (128 + 512) * 2 * 2                     # fp16 hi + lo of keys and values
    ns_m, n_m, cm = load(out + "/mfma", ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
    ns_f, n_f, cf = load(out + "/fetch", ["FETCH_SIZE"])
    ns_w, n_w, cw = load(out + "/write", ["WRITE_SIZE"])
    ms = ns_m / reads / 1e6
    fl = 1280.0 * T * hw * hw
    busy = 100.0 * cm["SQ_VALU_MFMA_BUSY_CYCLES"] / (cm["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    busy_peak = 100.0 * cm["SQ_VALU_MFMA_BUSY_CYCLES"] / (ns_m * 2.4 * 1024.0)
    clock = cm["GRBM_GUI_ACTIVE"] / 8.0 / ns_m
    rd = 2.0 * cf["FETCH_SIZE"] * 1024.0 / reads
    wr = cw["WRITE_SIZE"] * 1024.0 / reads
    ms_f = ns_f / reads / 1e6
    print("| %d | %.2f | %d | %.2f | %.0f | %.2f | %.1f | %.1f | %.2f | %.2f | %.2f | %.0f |" %
          (T, bank / 1e9, n_m // reads, ms, fl / ms / 1e9, clock, busy, busy_peak, rd / 1e9, rd / bank, wr / 1e9, (rd + wr) / (ms_f * 1e-3) / 1e9))


if __name__ == "__main__":
    main()
