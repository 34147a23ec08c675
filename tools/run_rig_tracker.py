"""Runs the one-call rig tracker on one rendered rig frame pair N times (profiling harness: rocprofv3 --kernel-trace)."""
import sys
import time

import numpy as np

from vieo_slam_amd import synth_scene as sc
from vieo_slam_amd.pipeline_rig import RigFrontEnd
from vieo_slam_amd.tracker import Tracker, rig_params


def inputs(fe, fr0, mps, case):
    from vieo_slam_amd.map_point import FRUSTUM_POINT_DTYPE
    pts = fe.last_frame_points(fr0, mps)
    has = mps["key_mp"] >= 0
    pts["reserved"][has, 0] = mps["first_key"][mps["key_mp"][has]] + 1
    z = fr0.fe["group_p3d"][np.nonzero(fr0.fe["group_good"])[0]][:, 2].astype(np.float32)
    last_depth = np.full(fr0.N, np.inf, np.float32)
    last_depth[has] = z[mps["key_mp"][has]]
    _, P = fe._frustum(np.eye(3, 4), mps, case["pose0"])
    return pts, last_depth, np.ascontiguousarray(P, FRUSTUM_POINT_DTYPE), mps["first_key"].astype(np.int32)


def main():
    rig, nc, nfeat, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    scene = sc.RigScene(5, rig, nc)
    case = sc.make_rig_tracking_case(5, scene)
    fe = RigFrontEnd(scene, nfeat)
    fr0 = fe.make_frame(case["images0"])
    mps = fe.make_map_points(fr0, case["pose0"][2], case["pose0"][3])
    pts, ld, P, alias = inputs(fe, fr0, mps, case)
    prm, rg = rig_params(scene, nfeat, max_local_points=len(P) + 10)
    trk = Tracker(prm, rg)
    nav = case["vio"][0]["nav_last"]
    ms = []
    for k in range(n):
        t0 = time.perf_counter()
        o, v = trk.track(None, None, case["imu_samples"], 0.0, case["dt_frame"], nav, nav, None, pts, ld, P, mps["desc"],
                         alias, 1, images=case["images1"])
        ms.append((time.perf_counter() - t0) * 1e3)
    print("rig %s x%d %d feats: %d keys, %d stereo groups, matches %d + %d, call %.2f ms (GPU %.2f), wall %.2f ms" % (
        rig, nc, nfeat, int(o["n_keys"]), int(o["n_groups"]), int(o["n_matches_last"]), int(o["n_matches_local"]),
        float(o["ms_host"]), float(o["ms_gpu"]), np.median(ms)))
    trk.close()


if __name__ == "__main__":
    main()
