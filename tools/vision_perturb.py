"""How a rounding-size perturbation travels through the vision-only replay (configs[0]) on the ORACLE alone: the replay is
run twice on the CPU restatement, the second time with the motion model's velocity moved by `eps` metres once, before
frame `at`.  Prints the position difference of every later frame.  (examples/replay_modes --vision against the Python driver
differ by the rounding of their 4x4 products -- 5e-16 m at frame 1 -- and the difference doubles per frame until it reaches
the optimiser's own noise floor: is that the estimator or the port?)
usage: python tools/vision_perturb.py [frames=40] [eps=1e-13] [at=2]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_lib  # noqa: E402
from tests.replay_oracle import OracleVisionStages  # noqa: E402
from vieo_slam_amd import replay, replay_modes as rm  # noqa: E402


class Perturbed(rm.VisionReplay):
    def __init__(self, *a, eps=0.0, at=2, **kw):
        super().__init__(*a, **kw)
        self.eps, self.at = eps, at

    def step(self, k):
        if k == self.at and self.eps:
            self.velocity = self.velocity.copy()
            self.velocity[0, 3] += self.eps
        return super().step(k)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-13
    at = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    orc = oracle_lib.load()
    seq = replay.Sequence(2, n)
    A = rm.VisionReplay(seq, OracleVisionStages(orc), lba_lag=3)
    ta = A.run(n)
    B = Perturbed(seq, OracleVisionStages(orc), lba_lag=3, eps=eps, at=at)
    tb = B.run(n)
    d = np.linalg.norm(ta["p"] - tb["p"], axis=1)
    for k in range(n):
        print("frame %2d  |dp| %.3e  %s" % (k, d[k], "" if k == 0 or d[k - 1] == 0 else "x%.2f" % (d[k] / d[k - 1])))
    print("same integer decisions:", A.stats["n_matches"] == B.stats["n_matches"] and A.stats["n_inliers"] == B.stats["n_inliers"])
