"""cProfile of the host side of the chained sequential replay (where the Python driver spends its time per frame)."""
import sys, time, numpy as np, cProfile, pstats
sys.path.insert(0, "/root/repo")
from vieo_slam_amd import replay
n = 45
seq = replay.Sequence(1, n)
for k in range(n): seq.images(k)
R = replay.ChainedReplay(seq, replay.HipStages()); R.initialise()
for k in range(1, 12): R.step(k)
pr = cProfile.Profile(); pr.enable()
for k in range(12, 42): R.step(k)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
