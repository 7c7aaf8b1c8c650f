"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a per-kernel stats table (markdown/CSV).

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def main(path, fmt="md"):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        d = agg.setdefault(short(n), [0, 0, 1e30, 0])
        dur = e - s
        d[0] += 1; d[1] += dur; d[2] = min(d[2], dur); d[3] = max(d[3], dur)
    total = sum(v[1] for v in agg.values())
    out = []
    out.append("| kernel | calls | total ms | avg us | min us | max us | % |")
    out.append("|---|---:|---:|---:|---:|---:|---:|")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.2f |" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3, v[2] / 1e3, v[3] / 1e3,
                                                                    100.0 * v[1] / total))
    out.append("")
    out.append("total kernel time: %.3f ms over %d dispatches" % (total / 1e6, len(rows)))
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
