import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vieo_slam_amd import replay, replay_modes as rm, synth_ba
# (1) which call of the resident test fails
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import test_resident_frame as T
from vieo_slam_amd.matching import ORBmatcher, compute_stereo_matches, compute_stereo_matches_resident
eL, eR, kl, dl, kr, dr = T._hip_frame(1000)
ur_r, dp_r = compute_stereo_matches_resident(eL, eR, T.BASELINE, T.BF)
ur_h, dp_h = compute_stereo_matches(eL, eR, kl, dl, kr, dr, T.BASELINE, T.BF)
sc_ = np.asarray(eL.GetScaleFactors(), np.float32)
M = ORBmatcher(0.9, True)
for th, dt in ((7.0, (0.02, -0.01, 0.03)), (15.0, (0.05, 0.02, -0.2))):
    pts, cam = T._points_and_cam(kl, dl, dp_r, sc_, th, dt)
    for name, fn in (("res", lambda: M.search_last_frame_resident(eL, pts, cam)),
                     ("host", lambda: M.SearchByProjectionLastFrame(M.project_last_frame(pts, cam), kl, ur_h, dl, None, T.BOUNDS)),
                     ("res_ur", lambda: M.search_last_frame_resident(eL, pts, cam, uright=ur_h)),
                     ("res2", lambda: M.search_last_frame_resident(eL, pts, cam)),
                     ("res_ur_mod", lambda: M.search_last_frame_resident(eL, pts, cam, uright=ur_h + np.float32(0))),
                     ("res3", lambda: M.search_last_frame_resident(eL, pts, cam))):
        try:
            n, a = fn()
            print(th, name, n, int((a >= 0).sum()))
        except Exception as e:
            print(th, name, "ERR", str(e)[-80:])
# (2) rig: staged HIP vs one call, frame by frame
seed, rig, nc, nfeat, n, lag = 3, "radtan", 2, 1200, 32, 3
seq = rm.RigSequence(seed, n, rig, nc)
Rh = rm.RigReplay(seq, rm.HipRigStages(nfeat, nc), nfeat, lba_lag=lag); th_ = Rh.run(n)
Rt = rm.RigTrackerReplay(seq, rm.HipRigStages(nfeat, nc), nfeat, lba_lag=lag); tt = Rt.run(n); Rt.close()
from vieo_slam_amd._lib import lib
lib().vieo_pose_set_replicas(0)
Ru = rm.RigTrackerReplay(seq, rm.HipRigStages(nfeat, nc), nfeat, lba_lag=lag); tu = Ru.run(n); Ru.close()
lib().vieo_pose_set_replicas(1)
for k in range(n):
    print(k, "staged-vs-call %.2e" % np.linalg.norm(th_[k]["p"] - tt[k]["p"]), "call-vs-call(no replicas) %.2e" % np.linalg.norm(tu[k]["p"] - tt[k]["p"]),
          Rh.stats["n_matches"][k - 1] if k else "", Rt.stats["n_matches"][k - 1] if k else "", Rh.stats["n_inliers"][k - 1] if k else "", Rt.stats["n_inliers"][k - 1] if k else "")
