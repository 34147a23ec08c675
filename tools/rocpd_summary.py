#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite result (kernel-trace) into a per-kernel stats table, the same columns as
`--stats` csv output, one row per (kernel, grid size) -- a kernel launched once per batch and once per set-up image
must not be averaged into one line, the roofline is about the batch launch.

Usage: tools/rocpd_summary.py results.db [out.md] [--merge-grids] [--min-us X]
           [--bytes SUBSTR=BYTES_PER_LAUNCH[@MIN_GRID]] ... [--flops SUBSTR=FLOPS_PER_LAUNCH[@MIN_GRID]] ...
--bytes / --flops append one roofline line per entry: the launches of the kernels whose name contains SUBSTR (with
a grid of at least MIN_GRID work-items), their average duration, algorithmic bytes (FLOPs) / that average, and the
fraction of the MI355X peak (HBM 8 TB/s; FP64 matrix 78.6 TFLOP/s)."""
import sqlite3
import sys

HBM_PEAK_GBS, FP64_MFMA_PEAK_TF = 8000.0, 78.6


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gcols = [c for c in ("grid_x", "grid_y", "grid_z") if c in cols]
    if len(gcols) == 3:
        q = "select %s, start, end, grid_x * grid_y * grid_z from kernels" % name_col
    elif "grid_size" in cols:
        q = "select %s, start, end, grid_size from kernels" % name_col
    else:
        q = "select %s, start, end, 0 from kernels" % name_col
    return cur.execute(q).fetchall()


def summarise(path, merge=False, min_us=0.0, rooflines=()):
    rows = load(path)
    agg = {}
    for name, s, e, g in rows:
        agg.setdefault((name, 0 if merge else int(g or 0)), []).append(e - s)
    total = sum(sum(v) for v in agg.values()) or 1
    out = ["| kernel | grid (work-items) | calls | total_ms | avg_us | min_us | max_us | % |", "|---|---|---|---|---|---|---|---|"]
    for (name, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / len(v) / 1e3 < min_us:
            continue
        short = name.split("(")[0][:70]
        out.append("| %s | %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (
            short, "all" if merge else g, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3,
            100.0 * sum(v) / total))
    for kind, sub, amount, min_grid in rooflines:
        d = [e - s for name, s, e, g in rows if sub in name and int(g or 0) >= min_grid]
        if not d:
            out.append("\nroofline %s: no launch with a grid >= %d" % (sub, min_grid))
            continue
        avg_s = sum(d) / len(d) / 1e9
        if kind == "bytes":
            gbs = amount / avg_s / 1e9
            out.append("\nroofline `%s` (grid >= %d): %d launches, avg %.1f us, %.3e algorithmic bytes per launch -> %.0f GB/s = "
                       "%.4f of the %.0f GB/s HBM peak" % (sub, min_grid, len(d), avg_s * 1e6, amount, gbs, gbs / HBM_PEAK_GBS,
                                                           HBM_PEAK_GBS))
        else:
            tf = amount / avg_s / 1e12
            out.append("\nroofline `%s` (grid >= %d): %d launches, avg %.1f us, %.3e FLOP per launch -> %.2f TFLOP/s = %.4f of "
                       "the %.1f TFLOP/s FP64 matrix peak" % (sub, min_grid, len(d), avg_s * 1e6, amount, tf,
                                                              tf / FP64_MFMA_PEAK_TF, FP64_MFMA_PEAK_TF))
    return "\n".join(out)


def main(argv):
    pos, merge, min_us, roof = [], False, 0.0, []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a == "--merge-grids":
            merge = True
        elif a == "--min-us":
            i += 1
            min_us = float(argv[i])
        elif a in ("--bytes", "--flops"):
            i += 1
            sub, rest = argv[i].split("=", 1)
            amount, _, mg = rest.partition("@")
            roof.append((a[2:], sub, float(amount), int(mg or 0)))
        else:
            pos.append(a)
        i += 1
    txt = summarise(pos[0], merge, min_us, roof)
    if len(pos) > 1:
        open(pos[1], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1:])
