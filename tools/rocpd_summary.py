#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite result (kernel-trace) into a per-kernel stats table,
the same columns as `--stats` csv output.  Usage: tools/rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def summarise(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [])
        a.append(e - s)
    total = sum(sum(v) for v in agg.values()) or 1
    out = ["| kernel | calls | total_ms | avg_us | min_us | max_us | % |", "|---|---|---|---|---|---|---|"]
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        short = name.split("(")[0][:70]
        out.append("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (
            short, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3,
            100.0 * sum(v) / total))
    return "\n".join(out)


if __name__ == "__main__":
    txt = summarise(sys.argv[1])
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)
