"""Where the one-call rig tracker leaves the oracle's replay BEFORE any integer decision differs (verdict r5 item 6):
per frame |p_hip - p_oracle|, per local BA the difference of its outputs and its trial counts.
    python tools/rig_drift.py kb8 4 1500 5 [frames=16] [lag=8]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import oracle_lib
from tests.replay_oracle import OracleRigStages
from vieo_slam_amd import replay, replay_modes as rm, synth_ba

rig, nc, nfeat, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = int(sys.argv[5]) if len(sys.argv) > 5 else 16
lag = int(sys.argv[6]) if len(sys.argv) > 6 else 8
orc = oracle_lib.load()
seq = rm.RigSequence(seed, n, rig, nc)


def hook(R, log):
    orig = R.local_ba

    def lb(apply=True):
        job = orig(apply)
        log.append(dict(navs=job["navs"].copy(), Xo=job["Xo"].copy(), erase=job["erase"].copy(), trials=int(job["res"]["lm_trials"]),
                        its=int(job["res"]["lm_iterations"]), chi=(float(job["res"]["chi2_initial"]), float(job["res"]["chi2_final"])),
                        n_kf=len(job["kfs"]), n_pts=len(job["pts"]), n_obs=len(job["rows"]), kfs_in=job["kfs"].copy()))
        return job
    R.local_ba = lb


lo, lh = [], []
Ro = rm.RigReplay(seq, OracleRigStages(orc, nfeat, nc), nfeat, lba_lag=lag)
hook(Ro, lo)
Ro.stats["keep_marg"] = True
to = Ro.run(n)
Rh = rm.RigTrackerReplay(seq, rm.HipRigStages(nfeat, nc), nfeat, lba_lag=lag, prefetch=False)
hook(Rh, lh)
Rh.stats["keep_marg"] = True
th = Rh.run(n)
Rh.close()
flip = replay.first_decision_flip(Rh.stats, Ro.stats)
print("%s x%d, %d features, seed %d, %d frames, write-back %d frames behind; first frame with a different integer decision: %s" % (rig, nc, nfeat, seed, n, lag, flip))
d = np.linalg.norm(th["p"] - to["p"], axis=1)
for k in range(n):
    m = Rh.stats["n_matches"][k - 1] if 0 < k <= len(Rh.stats["n_matches"]) else None
    mo = Ro.stats["n_matches"][k - 1] if 0 < k <= len(Ro.stats["n_matches"]) else None
    print("frame %2d  |dp| %.3e  rot %.3e  |dv| %.3e |dbg| %.3e  matches hip %s oracle %s  inliers %s / %s  LM iterations hip %s oracle %s" % (
        k, d[k], synth_ba.pose_error(th[k], to[k])[1], np.linalg.norm(th[k]["v"] - to[k]["v"]), np.linalg.norm(th[k]["dbg"] - to[k]["dbg"]), m, mo,
        Rh.stats["n_inliers"][k - 1] if 0 < k <= len(Rh.stats["n_inliers"]) else None,
        Ro.stats["n_inliers"][k - 1] if 0 < k <= len(Ro.stats["n_inliers"]) else None,
        Rh.stats["lm_iterations"][k - 1] if 0 < k <= len(Rh.stats.get("lm_iterations", [])) else None,
        Ro.stats["lm_iterations"][k - 1] if 0 < k <= len(Ro.stats.get("lm_iterations", [])) else None))
ca, cb = Rh.stats.get("assignment_crc", []), Ro.stats.get("assignment_crc", [])
for k, (a, b) in enumerate(zip(ca, cb)):
    if a != b:
        print("frame %d: the first frame whose key -> point assignment (%s) or outlier set (%s) differs from the oracle's" % (
            k + 1, "differs" if a[0] != b[0] else "same", "differs" if a[1] != b[1] else "same"))
        break
else:
    print("assignments and outlier sets equal in all %d frames" % len(ca))
print("marginal prior of frame k (H_marg, Optimizer.h:663-813): relative difference hip vs oracle, and its conditioning")
for k, (a, b) in enumerate(zip(Rh.stats.get("H_marg", []), Ro.stats.get("H_marg", []))):
    if a is None or b is None:
        continue
    A, B = np.array(a).reshape(15, 15), np.array(b).reshape(15, 15)
    ev = np.linalg.eigvalsh(0.5 * (B + B.T))
    print("frame %2d  |dH| / |H| %.2e   eigenvalues of H_marg: min %.3e max %.3e  (condition %.1e)" % (
        k + 1, np.linalg.norm(A - B) / np.linalg.norm(B), ev.min(), ev.max(), ev.max() / max(abs(ev.min()), 1e-300)))
for i, (a, b) in enumerate(zip(lh, lo)):
    same_in = np.array_equal(a["kfs_in"]["nav"]["p"], b["kfs_in"]["nav"]["p"])
    din = float(np.abs(a["kfs_in"]["nav"]["p"] - b["kfs_in"]["nav"]["p"]).max()) if a["kfs_in"].shape == b["kfs_in"].shape else float("nan")
    dn = float(np.abs(a["navs"]["p"] - b["navs"]["p"]).max()) if a["navs"].shape == b["navs"].shape else float("nan")
    dx = float(np.abs(a["Xo"] - b["Xo"]).max()) if a["Xo"].shape == b["Xo"].shape else float("nan")
    print("local BA %d: %d key frames, %d points, %d observations | input poses differ by %.2e | trials hip %d oracle %d, iterations %d / %d, "
          "chi2 %.6g -> %.6g (oracle %.6g -> %.6g) | output poses differ by %.2e, points by %.2e, erase flags equal %s" % (
              i, a["n_kf"], a["n_pts"], a["n_obs"], din, a["trials"], b["trials"], a["its"], b["its"], a["chi"][0], a["chi"][1], b["chi"][0], b["chi"][1],
              dn, dx, np.array_equal(a["erase"], b["erase"])))
