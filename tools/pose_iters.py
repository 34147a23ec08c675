import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from vieo_slam_amd import replay
from vieo_slam_amd.tracker import TrackerReplay
n = 40
seq = replay.Sequence(1, n)
R = TrackerReplay(seq, replay.HipStages())
its = []
orig = R.trk.track
def tr(*a, **k):
    o, v = orig(*a, **k)
    its.append((int(o["first"]["base"]["lm_iterations"]), int(o["second"]["base"]["lm_iterations"]), int(o["first"]["base"]["n_inliers"]), int(o["second"]["base"]["n_inliers"])))
    return o, v
R.trk.track = tr
R.run(n)
a = np.array(its)
print("lm iterations first/second mean", a[:, 0].mean(), a[:, 1].mean(), "min/max", a[:, :2].min(0), a[:, :2].max(0), "inliers", a[:, 2].mean(), a[:, 3].mean())
