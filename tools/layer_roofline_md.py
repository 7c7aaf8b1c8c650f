"""Per-layer roofline table from bench.py --layer-report: for every convolution of the frame its HIP-event time, the
algorithmic FLOPs and bytes (input + weights + output (+ residual) once, fp32), and which roof is closer -- the matrix
cores (nominal 833 TFLOP/s f16x3; ~550 under the power limit, tools/probes/mfma_probe.hip) or HBM (8 TB/s nominal; 5.5-6
achievable, gn_apply streams at 5.8).

    python tools/layer_roofline_md.py layers_1080p.json [title] > profiles/rNN_layer_roofline_1080p.md
"""
import json
import sys

PEAK_T, PWR_T, PEAK_B, ACH_B = 2500.0 / 3.0, 550.0, 8.0, 5.8


def main():
    rows = json.load(open(sys.argv[1]))
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    tot = sum(r["ms_per_frame"] for r in rows)
    print("# per-layer roofline, %s" % title)
    print("# time = HIP events around each otvm_conv2d launch (instrumented replay, launches of a layer summed); t_mfma = FLOPs / "
          "%.0f TFLOP/s (measured power-limited roof), t_hbm = bytes / %.1f TB/s (what a streaming kernel achieves here); "
          "bound = the larger of the two; eff = bound time / measured time; last column = measured time over the SUM of the two" % (PWR_T, ACH_B))
    print()
    print("| layer | launches | ms | GFLOP | MB | TFLOP/s | of 833 | TB/s | of 8 | bound | eff | ms / (t_mfma + t_hbm) |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---|---:|---:|")
    agg = {"mfma": [0.0, 0.0], "hbm": [0.0, 0.0]}
    add_sum = 0.0
    for r in rows:
        ms, gf, gb = r["ms_per_frame"], r["gflop_per_frame"], r.get("gbyte_per_frame", 0.0)
        t_m, t_b = gf / PWR_T, gb / ACH_B                     # ms (GFLOP / (TFLOP/s) = ms)
        bound = "mfma" if t_m >= t_b else "hbm"
        best = max(t_m, t_b)
        agg[bound][0] += ms; agg[bound][1] += best
        add_sum += t_m + t_b
        print("| %s | %.1f | %.3f | %.1f | %.0f | %.0f | %.2f | %.2f | %.2f | %s | %.2f | %.2f |" %
              (r["layer"].replace("conv ", ""), r["launches_per_frame"], ms, gf, gb * 1e3, r["tflops"], r["tflops"] / PEAK_T,
               r.get("tbyte_per_s", 0.0), r.get("tbyte_per_s", 0.0) / PEAK_B, bound, best / ms if ms else 0.0,
               ms / (t_m + t_b) if t_m + t_b > 0 else 0.0))
    print()
    print("total %.2f ms per frame; layers whose nearer roof is the matrix cores: %.2f ms measured vs %.2f ms at that roof; "
          "layers whose nearer roof is HBM: %.2f ms measured vs %.2f ms at that roof" %
          (tot, agg["mfma"][0], agg["mfma"][1], agg["hbm"][0], agg["hbm"][1]))
    print()
    print("additive model (under the power cap matrix-core time and memory time add instead of overlapping, DESIGN.md 3): "
          "sum over the layers of t_mfma + t_hbm = %.2f ms; measured %.2f ms = %.2f x" % (add_sum, tot, tot / add_sum))


if __name__ == "__main__":
    main()
