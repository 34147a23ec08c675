"""tools/rig_perturb.py on the HIP paths: (a) the one-call tracker against itself with frame kp's velocity moved by eps, (b) the STAGED
HIP replay (one host-pointer call per stage) against the oracle -- which of the two leaves the oracle, and does the path amplify?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import oracle_lib
from tests.replay_oracle import OracleRigStages
from vieo_slam_amd import replay, replay_modes as rm

rig, nc, nfeat, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 12
kp, eps = 5, 1e-12
seq = rm.RigSequence(seed, n, rig, nc)


def run(mk, perturb):
    R = mk()
    fin = R._finish_frame

    def finish(k, f, t0):
        if perturb and k == kp:
            f.nav["v"][0] += eps
        return fin(k, f, t0)
    R._finish_frame = finish
    t = R.run(n)
    if hasattr(R, "close"):
        R.close()
    return t, R

def show(tag, ta, tb):
    print(tag)
    for k in range(n):
        print("  frame %2d  |dp| %.3e  |dv| %.3e  |dbg| %.3e  |dba| %.3e" % (k, np.linalg.norm(ta[k]["p"] - tb[k]["p"]), np.linalg.norm(ta[k]["v"] - tb[k]["v"]),
                                                                   np.linalg.norm(ta[k]["dbg"] - tb[k]["dbg"]), np.linalg.norm(ta[k]["dba"] - tb[k]["dba"])))

trk = lambda: rm.RigTrackerReplay(seq, rm.HipRigStages(nfeat, nc), nfeat, lba_lag=8, prefetch=False)
stg = lambda: rm.RigReplay(seq, rm.HipRigStages(nfeat, nc), nfeat, lba_lag=8)
orc = oracle_lib.load()
ora = lambda: rm.RigReplay(seq, OracleRigStages(orc, nfeat, nc), nfeat, lba_lag=8)
t0, _ = run(trk, False)
t1, _ = run(trk, True)
show("one-call tracker vs itself with v_x of frame %d moved by %.0e:" % (kp, eps), t0, t1)
ts, _ = run(stg, False)
to, _ = run(ora, False)
show("staged HIP replay vs oracle:", ts, to)
show("one-call tracker vs staged HIP replay:", t0, ts)
