"""Chained sequential replay (replay.ChainedReplay): ms per frame split into host preparation / launches / GPU wait, and a
cProfile of the driver; run under rocprofv3 --kernel-trace for the per-kernel times (profiles/r2g_chained_frame.md)."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from vieo_slam_amd import replay
n = 60
seq = replay.Sequence(1, n)
for k in range(n): seq.images(k)
Rc = replay.ChainedReplay(seq, replay.HipStages())
t0=time.perf_counter(); Rc.run(n); t=time.perf_counter()-t0
c = np.array(Rc.stats["ms_chain"])[5:]
print("total per frame %.2f ms; ms_frames mean %.2f; prep %.2f launch %.2f wait %.2f; lba mean %.2f" % (1e3*t/(n-1), np.mean(Rc.stats["ms_frames"][5:]), c[:,0].mean(), c[:,1].mean(), c[:,2].mean(), np.mean(Rc.stats["ms_lba"])))
import cProfile, pstats
Rc2 = replay.ChainedReplay(seq, replay.HipStages()); Rc2.initialise()
for k in range(1, 12): Rc2.step(k)
pr = cProfile.Profile(); pr.enable()
for k in range(12, 40): Rc2.step(k)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
