"""Per-call differences of the float-valued stages (pre-integration, isInFrustum depths, stereo depths, local BA) on IDENTICAL inputs along
the oracle's replay of a rig sequence."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import oracle_lib
from tests.replay_oracle import OracleRigStages
from vieo_slam_amd import replay_modes as rm

rig, nc, nfeat, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 12
orc = oracle_lib.load()
seq = rm.RigSequence(seed, n, rig, nc)
S = OracleRigStages(orc, nfeat, nc)
Hs = rm.HipRigStages(nfeat, nc)
log = []


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

o_pre = S.preintegrate
def pre(noise, samples, ti, tj, bg, ba):
    ro = o_pre(noise, samples, ti, tj, bg, ba)
    rh = Hs.preintegrate(noise, samples, ti, tj, bg, ba)
    d = {k: rel(rh[0][k], ro[0][k]) for k in ro[0].dtype.names if ro[0][k].dtype.kind == "f"}
    log.append(("preintegrate [%.3f, %.3f] %d samples" % (ti, tj, len(samples)), max(d.values()), max(d, key=d.get)))
    return ro
S.preintegrate = pre
o_fr = S.in_frustum
def fr(F, P):
    ro = o_fr(F, P)
    rh = Hs.in_frustum(F, P)
    d = {k: rel(rh[k], ro[k]) for k in ro.dtype.names if ro[k].dtype.kind == "f"}
    log.append(("isInFrustum %d points" % len(P), max(d.values()), max(d, key=d.get)))
    return ro
S.in_frustum = fr
o_lba = S.lba_vio
def lba(*a):
    ro = o_lba(*a)
    rh = Hs.lba_vio(*a)
    log.append(("local BA: trials hip %d oracle %d" % (int(rh[3]["lm_trials"]), int(ro[3]["lm_trials"])), max(rel(rh[0]["p"], ro[0]["p"]), rel(rh[1], ro[1])), "poses / points"))
    return ro
S.lba_vio = lba
R = rm.RigReplay(seq, S, nfeat, lba_lag=8)
R.run(n)
for name, v, which in log:
    print("%-52s max relative difference %.2e (%s)" % (name, v, which))
