#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 rocpd database restricted to the steady-state window: from the
start of the N-th last launch of an anchor kernel to the end of the trace (or, with skip_last, to the start of the
skip_last-th last anchor launch: bench.py runs a few more extractor launches alone after the timed steps).
With end_anchor the window closes at the end of the last launch of that kernel before the window's nominal end (the
front end's last kernel of the timed steps: what follows is the drain of the low-priority bundle-adjustment stream).
Usage: tools/rocpd_window.py results.db anchor_kernel_substring n_steps [launches_per_step [skip_last [end_anchor]]]"""
import sqlite3
import sys


def main():
    path, anchor, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    per = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    skip = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    anc = [r for r in rows if anchor in r[0]]
    t0 = anc[-(n + skip) * per][1]
    t1 = anc[-skip * per][1] if skip else rows[-1][2] + 1
    if len(sys.argv) > 6:
        ends = [r[2] for r in rows if sys.argv[6] in r[0] and r[1] < t1]
        t1 = ends[-1] + 1
        skip = 1  # the span below runs to t1
    agg = {}
    for name, s, e in rows:
        if t0 <= s < t1:
            agg.setdefault(name.split("(")[0], []).append(e - s)
    span = ((t1 if skip else rows[-1][2]) - t0) / 1e6
    print("window: %.3f ms, %d steps -> %.3f ms/step" % (span, n, span / n))
    print("| kernel | launches/step | ms/step | avg_us |")
    print("|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("| %s | %.1f | %.3f | %.2f |" % (k[:60], len(v) / n, sum(v) / 1e6 / n, sum(v) / len(v) / 1e3))


if __name__ == "__main__":
    main()
