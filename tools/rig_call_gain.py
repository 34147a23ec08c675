"""Gain of ONE PoseOptimization call of the oracle's replay with respect to its inputs: the call is repeated ON THE ORACLE with single
inputs moved by 1e-10 (the size of the HIP / oracle differences that reach it).  CPU only.
    python tools/rig_call_gain.py kb8 4 1500 5 [frame=9] [call=1]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import oracle_lib
from tests.replay_oracle import OracleRigStages
from vieo_slam_amd import replay_modes as rm

rig, nc, nfeat, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
fr = int(sys.argv[5]) if len(sys.argv) > 5 else 9
call = int(sys.argv[6]) if len(sys.argv) > 6 else 1
orc = oracle_lib.load()
seq = rm.RigSequence(seed, fr + 1, rig, nc)
S = OracleRigStages(orc, nfeat, nc)
log = []
orig = S.pose_vio
def pv(F, obs):
    r, o = orig(F, obs)
    log.append((F.copy(), obs.copy()))
    return r, o
S.pose_vio = pv
rm.RigReplay(seq, S, nfeat, lba_lag=8).run(fr + 1)
F, obs = log[2 * (fr - 1) + call - 1]
r0, o0 = orc.pose_optimization_vio(F, obs)
print("frame %d call %d: %d observations, prior %d; reference solve: %d iterations, %d trials" % (fr, call, len(obs), int(F[0]["last_has_prior"]),
                                                                                         int(r0["base"]["lm_iterations"]), int(r0["base"]["reserved"])))
def d(r):
    a, b = r["base"]["nav"], r0["base"]["nav"]
    return "p %.1e v %.1e dbg %.1e dba %.1e (iterations %d, trials %d)" % (np.abs(a["p"] - b["p"]).max(), np.abs(a["v"] - b["v"]).max(), np.abs(a["dbg"] - b["dbg"]).max(),
                                                                      np.abs(a["dba"] - b["dba"]).max(), int(r["base"]["lm_iterations"]), int(r["base"]["reserved"]))
eps = 1e-10
for name, f in (("start state p_x", lambda G: G[0]["base"]["nav"]["p"].__setitem__(0, G[0]["base"]["nav"]["p"][0] + eps)),
                ("start state v_x", lambda G: G[0]["base"]["nav"]["v"].__setitem__(0, G[0]["base"]["nav"]["v"][0] + eps)),
                ("last state p_x (+ prior mean)", lambda G: (G[0]["nav_last"]["p"].__setitem__(0, G[0]["nav_last"]["p"][0] + eps), G[0]["nav_prior"]["p"].__setitem__(0, G[0]["nav_prior"]["p"][0] + eps))),
                ("last state v_x (+ prior mean)", lambda G: (G[0]["nav_last"]["v"].__setitem__(0, G[0]["nav_last"]["v"][0] + eps), G[0]["nav_prior"]["v"].__setitem__(0, G[0]["nav_prior"]["v"][0] + eps))),
                ("last state dba_x (+ prior mean)", lambda G: (G[0]["nav_last"]["dba"].__setitem__(0, G[0]["nav_last"]["dba"][0] + eps), G[0]["nav_prior"]["dba"].__setitem__(0, G[0]["nav_prior"]["dba"][0] + eps))),
                ("last state ba_x (+ prior mean)", lambda G: (G[0]["nav_last"]["ba"].__setitem__(0, G[0]["nav_last"]["ba"][0] + eps), G[0]["nav_prior"]["ba"].__setitem__(0, G[0]["nav_prior"]["ba"][0] + eps))),
                ("last state v_x only (not the prior mean)", lambda G: G[0]["nav_last"]["v"].__setitem__(0, G[0]["nav_last"]["v"][0] + eps)),
                ("H_prior[0][0] x (1 + 1e-12)", lambda G: G[0]["H_prior"].reshape(-1).__setitem__(0, G[0]["H_prior"].reshape(-1)[0] * (1 + 1e-12))),
                ("pre-integrated velocity vij_x", lambda G: G[0]["imu"]["vij"].__setitem__(0, G[0]["imu"]["vij"][0] + eps))):
    G = F.copy()
    f(G)
    r, o = orc.pose_optimization_vio(G, obs)
    print("  %-44s + 1e-10 -> output moves by %s%s" % (name, d(r), "" if np.array_equal(o, o0) else "  OUTLIER SET DIFFERS"))
