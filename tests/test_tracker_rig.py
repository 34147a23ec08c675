"""The one-call frame tracker for distorted camera rigs (vieo_tracker_create_rig + vieo_track_frame; BASELINE configs[3] /
[4] and the reference's default EuRoC_VIO_dist* set-up): ExtractORB x n_cams -> ComputeStereoFishEyeMatches ->
PredictNavStateByIMU -> SearchByProjection(last frame, camera loop) -> PoseOptimization(VIO, rig) -> isInFrustum ->
SearchByProjection(local map) -> PoseOptimization(VIO, marg) as ONE call with one host synchronisation, against the
stage-by-stage chain of pipeline_rig.py (every stage one C-ABI call) and against the same chain on the CPU oracle."""
import numpy as np
import pytest

from vieo_slam_amd import synth_ba
from vieo_slam_amd import synth_scene as sc


def _tracker_inputs(fe, fr0, mps, case):
    """mLastFrame flattened for the tracker: one entry per key of fr0 in mvKeys order; the keys of a stereo group all
    hold the group's map point and name the group's first key as their table entry; the local map = all map points."""
    from vieo_slam_amd.map_point import FRUSTUM_POINT_DTYPE
    pts = fe.last_frame_points(fr0, mps)
    has = mps["key_mp"] >= 0
    pts["reserved"][has, 0] = mps["first_key"][mps["key_mp"][has]] + 1
    z = fr0.fe["group_p3d"][np.nonzero(fr0.fe["group_good"])[0]][:, 2].astype(np.float32)  # a depth per map point
    last_depth = np.full(fr0.N, np.inf, np.float32)
    last_depth[has] = z[mps["key_mp"][has]]
    _, P = fe._frustum(np.eye(3, 4), mps, case["pose0"])
    P = np.ascontiguousarray(P, FRUSTUM_POINT_DTYPE)
    alias = mps["first_key"].astype(np.int32)
    return pts, last_depth, z, P, alias


@pytest.mark.gpu
@pytest.mark.parametrize("rig,nc,nfeat,seed", [("radtan", 2, 1200, 3), ("kb8", 2, 1500, 4), ("kb8", 4, 1500, 5),
                                               ("radtan", 4, 1200, 6)])
def test_gpu_rig_tracker_equals_the_staged_chain_and_the_oracle(oracle, rig, nc, nfeat, seed):
    from tests.test_pipeline_rig import OracleStages
    from vieo_slam_amd.pipeline_rig import RigFrontEnd
    from vieo_slam_amd.tracker import Tracker, rig_params
    scene = sc.RigScene(seed, rig, nc)
    case = sc.make_rig_tracking_case(seed, scene)
    fe = RigFrontEnd(scene, nfeat)
    fr0 = fe.make_frame(case["images0"])
    Ri, pi, Rwc0, twc0 = case["pose0"]
    mps = fe.make_map_points(fr0, Rwc0, twc0)
    pts, last_depth, z, P, alias = _tracker_inputs(fe, fr0, mps, case)
    prm, rg = rig_params(scene, nfeat, max_local_points=len(P) + 10, th_depth=float(case["vio"][0]["th_depth"]))
    trk = Tracker(prm, rg)
    assert trk.key_cap >= fr0.N
    nav_i = case["vio"][0]["nav_last"]
    o, v = trk.track(None, None, case["imu_samples"], 0.0, case["dt_frame"], nav_i, nav_i, None, pts, last_depth, P,
                     mps["desc"], alias, 1, images=case["images1"])
    assert int(o["status"]) == 0 and int(o["stereo_status"]) == 0 and int(o["widened"]) == 0
    # the prediction is the pre-integrated motion: close to the truth
    dt, dr = synth_ba.pose_error(o["nav_pred"], case["truth"])
    assert dt < 0.05 and dr < 0.02, (dt, dr)
    pred = (o["nav_pred"].copy(), o["imu"].copy())
    for name, stages in (("hip", None), ("oracle", OracleStages(oracle, nfeat, nc))):
        fs = RigFrontEnd(scene, nfeat, stages=stages)
        ref = fs.track(case, pred=pred, track_depth=z)
        fr1 = ref["fr1"]
        # Frame::Frame: mvKeys / mDescriptors / vdepth_ and the stereo tables
        assert int(o["n_keys"]) == fr1.N and np.array_equal(o["cam_first"][:nc + 1], fr1.cam_first)
        assert np.array_equal(v["keys"].view(np.uint8), fr1.keys.view(np.uint8)) and np.array_equal(v["desc"], fr1.desc)
        assert np.array_equal(o["mono_index"][:nc], fr1.mono) and (v["uright"] == -1).all()
        assert np.array_equal(v["key_group"], fr1.fe["key_group"]) and np.array_equal(v["group_idx"], fr1.fe["group_idx"])
        assert np.array_equal(v["group_good"], fr1.fe["group_good"]) and int(o["n_stereo_matches"]) == fr1.fe["n_matches"]
        assert np.array_equal(v["depth"] < 0, fr1.depth < 0) and np.allclose(v["depth"], fr1.depth, rtol=1e-6, atol=0)
        # the searches: which map point every key holds after TrackLocalMap
        assert int(o["n_matches_last"]) == ref["n1"] and int(o["n_matches_local"]) == ref["n2"], name
        tab = v["point_ref"].astype(np.int64)
        cap = int(o["key_cap"])
        held = np.full(fr1.N, -1, np.int64)
        a = (tab >= 0) & (tab < cap)
        held[a] = mps["key_mp"][tab[a]]
        held[tab >= cap] = tab[tab >= cap] - cap
        assert np.array_equal(held, ref["mp_ref"]), name
        out2 = np.zeros(fr1.N, np.uint8)
        out2[ref["idx2"]] = ref["o2"]
        assert np.array_equal(v["outlier"], out2), name
        for which, r in (("first", ref["r1"]), ("second", ref["r2"])):
            dt, dr = synth_ba.pose_error(o[which]["base"]["nav"], r["base"]["nav"])
            assert dt < 1e-4 and dr < 1e-4, (name, which, dt, dr)
            assert int(o[which]["base"]["n_inliers"]) == int(r["base"]["n_inliers"]), (name, which)
        Ho, Hh = ref["r2"]["H_marg"].reshape(15, 15), o["second"]["H_marg"].reshape(15, 15)
        assert int(o["second"]["has_marg"]) == 1 and np.allclose(Ho, Hh, rtol=1e-5, atol=1e-5 * np.abs(Ho).max())
    # and the frame is tracked: the optimised pose is the true one
    gdt, gdr = synth_ba.pose_error(o["second"]["base"]["nav"], case["truth"])
    assert gdt < 1e-2 and gdr < 5e-3 and int(o["second"]["base"]["n_inliers"]) > 60, (gdt, gdr)
    # a second call on the same handle (everything re-uploaded) gives the same frame
    o2, v2 = trk.track(None, None, case["imu_samples"], 0.0, case["dt_frame"], nav_i, nav_i, None, pts, last_depth, P,
                       mps["desc"], alias, 1, images=case["images1"])
    assert np.array_equal(v2["point_ref"], tab.astype(np.int32)) and o2["second"]["base"]["nav"].tobytes() == o["second"]["base"]["nav"].tobytes()
    print("rig tracker %s x%d: %.2f ms in the call (GPU %.2f)" % (rig, nc, float(o2["ms_host"]), float(o2["ms_gpu"])))
    trk.close()


@pytest.mark.gpu
def test_gpu_rig_tracker_edge_cases(monkeypatch):
    """No map yet (no last-frame points, no local map): the frame's extraction and stereo outputs are valid, the status is
    LOST (TrackWithIMU's `nmatches < 10`, Tracking.cc:311) after the widened search.  A stereo group table too small for
    the frame: stereo_status reports it, no depths, the call itself succeeds."""
    from vieo_slam_amd.ba_types import LAST_FRAME_POINT_DTYPE
    from vieo_slam_amd.map_point import FRUSTUM_POINT_DTYPE
    from vieo_slam_amd.pipeline_rig import RigFrontEnd
    from vieo_slam_amd.tracker import Tracker, rig_params
    scene = sc.RigScene(9, "kb8", 2)
    case = sc.make_rig_tracking_case(9, scene)
    prm, rg = rig_params(scene, 1500, max_local_points=256)
    trk = Tracker(prm, rg)
    nav = case["vio"][0]["nav_last"]
    none_p, none_c = np.zeros(0, LAST_FRAME_POINT_DTYPE), np.zeros(0, FRUSTUM_POINT_DTYPE)
    o, v = trk.track(None, None, case["imu_samples"], 0.0, case["dt_frame"], nav, nav, None, none_p, np.zeros(0, np.float32),
                     none_c, np.zeros((0, 32), np.uint8), np.zeros(0, np.int32), 1, images=case["images1"])
    assert int(o["status"]) == 2 and int(o["widened"]) == 1 and int(o["n_matches_last"]) == 0
    fr = RigFrontEnd(scene, 1500).make_frame(case["images1"])
    assert int(o["n_keys"]) == fr.N and np.array_equal(v["keys"].view(np.uint8), fr.keys.view(np.uint8))
    assert np.array_equal(v["key_group"], fr.fe["key_group"]) and int(o["n_groups"]) == len(fr.fe["group_good"]) > 100
    assert (v["point_ref"] == -1).all()
    trk.close()
    monkeypatch.setenv("VIEO_FE_GCAP", "64")
    trk = Tracker(prm, rg)
    o, v = trk.track(None, None, case["imu_samples"], 0.0, case["dt_frame"], nav, nav, None, none_p, np.zeros(0, np.float32),
                     none_c, np.zeros((0, 32), np.uint8), np.zeros(0, np.int32), 1, images=case["images1"])
    assert int(o["stereo_status"]) != 0 and int(o["n_groups"]) == 0 and (v["depth"] < 0).all() and int(o["n_keys"]) == fr.N
    trk.close()


@pytest.mark.gpu
def test_gpu_rig_tracker_repeats_the_frame_when_a_replica_never_arrives(monkeypatch):
    """The two optimisations of a rig frame run on 16 replica workgroups each; one that never becomes resident
    (VIEO_POSE_REPLICA_DROP) used to fail the frame with VIEO_E_HIP after one time-out per exchange.  Now the waiting
    replicas give up once, and the tracker repeats the tail of the chain with one workgroup per optimisation: the frame is
    late, not lost, and carries the one-workgroup result."""
    import time
    from vieo_slam_amd._lib import lib
    from vieo_slam_amd.pipeline_rig import RigFrontEnd
    from vieo_slam_amd.tracker import Tracker, rig_params
    seed, rig, nc, nfeat = 5, "kb8", 4, 1500
    scene = sc.RigScene(seed, rig, nc)
    case = sc.make_rig_tracking_case(seed, scene)
    fe = RigFrontEnd(scene, nfeat)
    fr0 = fe.make_frame(case["images0"])
    Ri, pi, Rwc0, twc0 = case["pose0"]
    mps = fe.make_map_points(fr0, Rwc0, twc0)
    pts, last_depth, z, P, alias = _tracker_inputs(fe, fr0, mps, case)
    prm, rg = rig_params(scene, nfeat, max_local_points=len(P) + 10, th_depth=float(case["vio"][0]["th_depth"]))
    nav_i = case["vio"][0]["nav_last"]
    args = (None, None, case["imu_samples"], 0.0, case["dt_frame"], nav_i, nav_i, None, pts, last_depth, P, mps["desc"], alias, 1)
    L = lib()
    was = L.vieo_pose_set_replicas(0)
    try:
        trk = Tracker(prm, rg)
        o1, v1 = trk.track(*args, images=case["images1"])
        one = (o1["first"].tobytes(), o1["second"].tobytes(), v1["point_ref"].copy(), v1["outlier"].copy())
        L.vieo_pose_set_replicas(1)
        monkeypatch.setenv("VIEO_POSE_REPLICA_DROP", "1")
        t0 = time.perf_counter()
        od, vd = trk.track(*args, images=case["images1"])
        dt_fail = time.perf_counter() - t0
        monkeypatch.delenv("VIEO_POSE_REPLICA_DROP")
        st = trk.stats()
        assert int(od["status"]) == 0 and int(od["first"]["base"]["status"]) == 0 and int(od["second"]["base"]["status"]) == 0
        assert (od["first"].tobytes(), od["second"].tobytes()) == one[:2]
        assert np.array_equal(vd["point_ref"], one[2]) and np.array_equal(vd["outlier"], one[3])
        assert st["replica_repeats"] == 1 and dt_fail < 5.0, (st, dt_fail)
        o16, v16 = trk.track(*args, images=case["images1"])  # and the replicated form works again right after
        assert int(o16["second"]["base"]["status"]) == 0 and trk.stats()["replica_repeats"] == 1
        dt, dr = synth_ba.pose_error(o16["second"]["base"]["nav"], o1["second"]["base"]["nav"])
        assert dt < 1e-9 and dr < 1e-7
        trk.close()
    finally:
        L.vieo_pose_set_replicas(was)
    print("rig tracker, replica dropped: frame back after %.2f s" % dt_fail)
