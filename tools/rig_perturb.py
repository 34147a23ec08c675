"""Is the float divergence of the 4-camera replay a property of the algorithm or of the HIP path?  The ORACLE replay twice:
as is, and with frame `kp`'s velocity moved by `eps` (1e-12 m/s: below anything either implementation rounds to).  CPU only.
    python tools/rig_perturb.py kb8 4 1500 5 [frames=12] [kp=5] [eps=1e-12]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import oracle_lib
from tests.replay_oracle import OracleRigStages
from vieo_slam_amd import replay, replay_modes as rm

rig, nc, nfeat, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 12
kp = int(sys.argv[6]) if len(sys.argv) > 6 else 5
eps = float(sys.argv[7]) if len(sys.argv) > 7 else 1e-12
FIELD = sys.argv[8] if len(sys.argv) > 8 else "v"
orc = oracle_lib.load()
seq = rm.RigSequence(seed, n, rig, nc)


def run(perturb):
    R = rm.RigReplay(seq, OracleRigStages(orc, nfeat, nc), nfeat, lba_lag=8)
    fin = R._finish_frame

    def finish(k, f, t0):
        if perturb and k == kp:
            if FIELD == "H":  # one diagonal entry of the frame's marginal prior, relative
                f.prior[1].reshape(-1)[0] *= 1.0 + eps
            else:
                f.nav[FIELD][0] += eps
        return fin(k, f, t0)
    R._finish_frame = finish
    return R.run(n), R

ta, Ra = run(False)
tb, Rb = run(True)
print("%s x%d seed %d: oracle replay vs the same replay with v_x of frame %d moved by %.0e m/s; first different integer decision: %s" % (
    rig, nc, seed, kp, eps, replay.first_decision_flip(Ra.stats, Rb.stats)))
for k in range(n):
    print("frame %2d  |dp| %.3e  |dv| %.3e  |dbg| %.3e  |dba| %.3e" % (k, np.linalg.norm(ta[k]["p"] - tb[k]["p"]), np.linalg.norm(ta[k]["v"] - tb[k]["v"]),
                                                             np.linalg.norm(ta[k]["dbg"] - tb[k]["dbg"]), np.linalg.norm(ta[k]["dba"] - tb[k]["dba"])))
