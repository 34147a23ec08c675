#!/usr/bin/env python3
"""Per-kernel average of PMC counters from a rocprofv3 rocpd sqlite result."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
rows = cur.execute("select * from pmc_events").fetchall()
ix = {c: i for i, c in enumerate(cols)}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
kcol = "name" if "name" in ix else [c for c in cols if "kernel" in c or "name" in c][0]
ccol = "counter_name" if "counter_name" in ix else [c for c in cols if "counter" in c and "name" in c][0]
vcol = "value" if "value" in ix else [c for c in cols if "value" in c][0]
for r in rows:
    agg[r[ix[kcol]].split("(")[0]][r[ix[ccol]]].append(r[ix[vcol]])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s avg %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
