import sys, time, collections
sys.path.insert(0, sys.argv[1])
import numpy as np
from vieo_slam_amd import replay
seq = replay.Sequence(1, 60)
S = replay.HipStages()
acc = collections.defaultdict(float); cnt = collections.Counter()
for name in dir(S):
    f = getattr(S, name)
    if callable(f) and not name.startswith("_") and hasattr(f, "__self__"):
        def wrap(f=f, name=name):
            def g(*a, **k):
                t = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t; cnt[name] += 1; return r
            return g
        setattr(S, name, wrap())
R = replay.Replay(seq, S)
t = time.perf_counter()
R.run(60)
tot = time.perf_counter() - t
print("total %.1f ms/frame" % (tot / 60 * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("%-24s %7.2f ms/frame  (%d calls)" % (k, v / 60 * 1e3, cnt[k]))
print("python glue %.2f ms/frame" % ((tot - sum(acc.values())) / 60 * 1e3))
