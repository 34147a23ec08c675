"""Per-call difference of PoseOptimization(VIO) on IDENTICAL inputs along the oracle's replay of a rig sequence: every call of the
oracle's optimiser is repeated on the HIP kernel with the same frame record and observations."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import oracle_lib
from tests.replay_oracle import OracleRigStages
from vieo_slam_amd import replay_modes as rm
from vieo_slam_amd.optimizer import Optimizer

rig, nc, nfeat, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 12
orc = oracle_lib.load()
seq = rm.RigSequence(seed, n, rig, nc)
S = OracleRigStages(orc, nfeat, nc)
orig = S.pose_vio
calls = []


def both(F, obs):
    ro, oo = orig(F, obs)
    rh, oh = Optimizer.PoseOptimizationVIO(F.copy(), obs.copy())
    a, b = rh["base"]["nav"], ro["base"]["nav"]
    calls.append((len(obs), int(F[0]["last_has_prior"]), int(F[0]["compute_marg"]), int(rh["base"]["lm_iterations"]), int(ro["base"]["lm_iterations"]),
                  float(np.linalg.norm(a["p"] - b["p"])), float(np.linalg.norm(a["v"] - b["v"])), float(np.linalg.norm(a["dbg"] - b["dbg"])),
                  float(np.linalg.norm(a["dba"] - b["dba"])), bool(np.array_equal(oo, oh)),
                  float(np.abs(rh["H_marg"] - ro["H_marg"]).max() / max(np.abs(ro["H_marg"]).max(), 1e-300))))
    return ro, oo
S.pose_vio = both
R = rm.RigReplay(seq, S, nfeat, lba_lag=8)
R.run(n)
print("%s x%d seed %d: HIP vs oracle PoseOptimization on the SAME inputs (the oracle's replay), two calls per frame" % (rig, nc, seed))
for i, c in enumerate(calls):
    print("frame %2d call %d  obs %4d prior %d marg %d  LM iterations hip %2d oracle %2d  |dp| %.2e |dv| %.2e |dbg| %.2e |dba| %.2e  outliers equal %s  dH %.1e" % (
        (1 + i // 2, 1 + i % 2) + c))
