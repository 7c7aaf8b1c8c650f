"""HBM traffic of the convolution kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in
separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <frames> > profiles/rNN_conv_traffic.json

Units/corrections (guide, section HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced read stream, so reads are doubled.  Both come from the L2's fabric-side request counters
(Infinity-Cache hits included), i.e. an upper bound on true HBM traffic.
"""
import collections
import csv
import json
import sys


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        key = "conv" if any(t in k for t in ("conv_igemm", "conv_patch", "conv_stem", "conv_head", "conv_wave", "stm_bottleneck", "splitk_finish")) else "other"
        d[key][0] += 1
        d[key][1] += float(r["Counter_Value"])
    return d


def main():
    f, w, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
    F, W = agg(f, "FETCH_SIZE"), agg(w, "WRITE_SIZE")
    n = F["conv"][0]
    rd = 2.0 * F["conv"][1] * 1024.0            # gfx950 correction: x2
    wr = W["conv"][1] * 1024.0
    out = {
        "kernel": "conv_igemm_f16x3 + conv_patch_f16x3 + conv_stem_f16x3 + conv_head16_f16x3 + stm_bottleneck(128)_f16x3 + splitk_finish kernels (everything the plan counts as a convolution launch)",
        "launches": n, "frames": frames, "launches_per_frame": n / frames,
        "read_bytes_per_launch": rd / n, "write_bytes_per_launch": wr / max(1, W["conv"][0]),
        "traffic_bytes_per_launch": rd / n + wr / max(1, W["conv"][0]),
        "traffic_bytes_per_frame": (rd + wr) / frames,
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), KiB -> bytes, FETCH_SIZE x2 (gfx950)",
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
