"""A rig sequence through the one-call tracker (replay_modes.RigTrackerReplay), for profiling: python tools/run_rig_sequence.py
[rig] [n_cams] [nfeat] [frames] [seed]; prints the per-frame call / GPU times."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vieo_slam_amd import replay_modes as rm

rig = sys.argv[1] if len(sys.argv) > 1 else "kb8"
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 5
seq = rm.RigSequence(seed, n, rig, nc)
for k in range(n):
    seq.images(k)
R = rm.RigTrackerReplay(seq, rm.HipRigStages(nf, nc), nf, lba_lag=6)
R.run(n)
ms = np.array(R.stats["ms_chain"])
print("frames", n, "call ms mean %.3f gpu %.3f" % (ms[:, 0].mean(), ms[:, 1].mean()))
for k in range(0, len(ms), 3):
    print(k + 1, "%.3f %.3f" % tuple(ms[k]), R.stats["n_matches"][k], R.stats["n_inliers"][k], "widened", R.stats["widened"])
print(R.trk.stats())
R.close()
