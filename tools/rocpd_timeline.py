"""Timeline of the kernels of one call from a rocprofv3 rocpd result: start / end in us relative to the first kernel of the
window, stream (queue) if the view has it.  python tools/rocpd_timeline.py <db> [first_kernel_substring] [occurrence] [count]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
extra = [c for c in ("queue_id", "stream_id", "queue") if c in cols]
q = "select %s, start, end%s from kernels order by start" % (name, "".join(", " + c for c in extra))
rows = cur.execute(q).fetchall()
key = sys.argv[2] if len(sys.argv) > 2 else "k_resize"
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cnt = int(sys.argv[4]) if len(sys.argv) > 4 else 70
idx = [i for i, r in enumerate(rows) if key in r[0]]
# the occ-th run of consecutive key kernels
i0 = idx[min(occ * 7, len(idx) - 1)]
t0 = rows[i0][1]
for r in rows[i0:i0 + cnt]:
    print("%9.1f %9.1f %7.1f  %s %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, " ".join(str(x) for x in r[3:]), r[0][:60]))
