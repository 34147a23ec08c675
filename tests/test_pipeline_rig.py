"""GPU: the front-end chain of a distorted camera-rig frame (BASELINE configs[3] / [4]) -- ORBextractor x n_cams ->
ComputeStereoFishEyeMatches -> SearchByProjection(last frame) -> PoseOptimization(VIO, rig) -> isInFrustum ->
SearchByProjection(local map) -> PoseOptimization(VIO, marg) -- through the C-ABI, every stage re-evaluated by the
CPU oracle on the same inputs (vieo_slam_amd/pipeline_rig.py records them)."""
import numpy as np
import pytest

from vieo_slam_amd import synth_ba
from vieo_slam_amd import synth_scene as sc



class OracleStages:
    def __init__(self, oracle, nfeatures, n_cams):
        self.o = oracle
        self.ext = [oracle.extractor(nfeatures) for _ in range(n_cams)]

    def extract(self, c, image, lapping):
        return self.ext[c](image, lapping)

    def fisheye(self, params, keys, descs, mono):
        return self.o.stereo_fisheye(params, keys, descs, mono)

    def project_last_frame(self, pts, cam, rig):
        return self.o.sbp_project_last_frame(pts, cam, rig)

    def search(self, mode, q, keys, ur, desc, taken, bounds, cam_first, nn, ori=True):
        return self.o.search_by_projection(mode, q, keys, ur, desc, taken, bounds, nn_ratio=nn, check_ori=ori,
                                           cam_first=cam_first)

    def pose_vio(self, F, obs):
        return self.o.pose_optimization_vio(F, obs)

    def in_frustum(self, F, P):
        return self.o.is_in_frustum(F, P)


def test_oracle_rig_chain_tracks(oracle):
    """CPU: the chain itself (host glue + oracle stages) recovers the true pose of a KB8 stereo frame"""
    from vieo_slam_amd.pipeline_rig import RigFrontEnd
    scene = sc.RigScene(4, "kb8", 2)
    case = sc.make_rig_tracking_case(4, scene)
    fe = RigFrontEnd(scene, 1500, stages=OracleStages(oracle, 1500, 2))
    out = fe.track(case, np.random.default_rng(4))
    gdt, gdr = synth_ba.pose_error(out["r2"]["base"]["nav"], case["truth"])
    assert gdt < 1e-2 and gdr < 5e-3 and out["n1"] > 300 and out["r2"]["base"]["n_inliers"] > 300
    assert (out["fr1"].mono == 0).all() and out["fr0"].fe["group_good"].sum() > 100


@pytest.mark.gpu
@pytest.mark.parametrize("rig,nc,nfeat,seed", [("radtan", 2, 1200, 3), ("kb8", 2, 1500, 4), ("kb8", 4, 1500, 5),
                                               ("radtan", 4, 1200, 6)])
def test_rig_frontend_chain_stage_by_stage(oracle, rig, nc, nfeat, seed):
    from vieo_slam_amd.pipeline_rig import RigFrontEnd
    scene = sc.RigScene(seed, rig, nc)
    case = sc.make_rig_tracking_case(seed, scene)
    fe = RigFrontEnd(scene, nfeat)
    out = fe.track(case, np.random.default_rng(seed))
    O = OracleStages(oracle, nfeat, nc)
    seen = {}
    for name, inp, got in fe.trace:
        seen[name] = seen.get(name, 0) + 1
        if name == "extract":
            mono, k, d = O.extract(inp["c"], inp["image"], inp["lapping"])
            assert mono == got[0] and len(k) == len(got[1]) > 0.8 * nfeat
            assert np.array_equal(k.view(np.uint8), got[1].view(np.uint8)) and np.array_equal(d, got[2])
        elif name == "fisheye":
            ref = O.fisheye(fe.fparams, inp["keys"], inp["descs"], inp["mono"])
            assert np.array_equal(ref["key_group"], got["key_group"]) and np.array_equal(ref["group_idx"], got["group_idx"])
            assert np.array_equal(ref["group_good"], got["group_good"]) and ref["n_matches"] == got["n_matches"] > 100
            g = ref["group_good"]
            assert np.allclose(ref["group_p3d"][g], got["group_p3d"][g], rtol=1e-9, atol=1e-12)
            assert np.allclose(ref["depth"], got["depth"], rtol=1e-6)
        elif name == "project_last_frame":
            ref = O.project_last_frame(inp["pts"], inp["cam"], fe.rig)
            assert np.array_equal(ref.view(np.uint8), got.view(np.uint8)) and ((ref["flags"] & 1) > 0).sum() > 150
        elif name == "search":
            fr = inp["fr"]
            n, a = O.search(inp["mode"], inp["q"], fr.keys, fr.uright, fr.desc, inp["taken"], fe.bounds, fr.cam_first,
                            inp["nn"])
            assert n == got[0] and np.array_equal(a, got[1]) and n > (60 if inp["mode"] == 0 else 5), (inp["mode"], n)
        elif name == "in_frustum":
            ref = O.in_frustum(inp["F"], inp["P"])
            assert ref.tobytes() == got.tobytes()
        elif name == "pose_vio":
            r, o = O.pose_vio(inp["F"], inp["obs"])
            dt, dr = synth_ba.pose_error(r["base"]["nav"], got[0]["base"]["nav"])
            assert dt < 1e-4 and dr < 1e-4, (dt, dr)
            assert r["base"]["n_inliers"] == got[0]["base"]["n_inliers"] and np.array_equal(o, got[1])
            if inp["F"][0]["compute_marg"]:
                Ho, Hh = r["H_marg"].reshape(15, 15), got[0]["H_marg"].reshape(15, 15)
                assert np.allclose(Ho, Hh, rtol=1e-5, atol=1e-5 * np.abs(Ho).max())
    assert seen == dict(extract=2 * nc, fisheye=2, project_last_frame=1, search=2, pose_vio=2, in_frustum=1)
    # and the chain tracks: the optimised pose is the true one
    gdt, gdr = synth_ba.pose_error(out["r2"]["base"]["nav"], case["truth"])
    assert gdt < 1e-2 and gdr < 5e-3, (gdt, gdr)
    assert out["r2"]["base"]["n_inliers"] > 60


@pytest.mark.gpu
@pytest.mark.parametrize("rig,nc,nfeat,seed", [("kb8", 4, 1500, 7), ("radtan", 2, 1200, 8)])
def test_batched_rig_pipeline_equals_the_staged_chain(rig, nc, nfeat, seed):
    """pipeline_rig_batch.RigFramePipeline -- a batch of rig frames device-resident through the whole tracking chain (what
    bench.py's batched rig leg times) -- against the stage-by-stage chain of RigFrontEnd on the same inputs (the chain
    tests/test_tracker_rig.py and the test above tie to the oracle): held map points, both optimisations."""
    from vieo_slam_amd.pipeline_rig import RigFrontEnd
    from vieo_slam_amd.pipeline_rig_batch import RigFramePipeline
    scene = sc.RigScene(seed, rig, nc)
    cases = [sc.make_rig_tracking_case(seed + 10 * i, scene) for i in range(2)]
    B = 5  # frames 2.. are noisy replicas of the two base cases
    P = RigFramePipeline(scene, cases, nfeat, B, seed=3)
    P.step()
    P.step()  # a second pass over the resident batch gives the same answer
    R = P.results()
    assert (R["hdr"][:, 3] == 0).all()
    for b in range(2):
        case, pr = cases[b], P.prep[b]
        F = P.f1_host[b]
        fs = RigFrontEnd(scene, nfeat)
        ref = fs.track(case, pred=(F["base"]["nav"].copy(), F["imu"].copy()), track_depth=pr["z"])
        fr1 = ref["fr1"]
        N = int(R["first"][b, nc])
        assert N == fr1.N and np.array_equal(R["keys"][b, :N].view(np.uint8), fr1.keys.view(np.uint8))
        assert np.allclose(R["depth"][b, :N], fr1.depth, rtol=1e-6, atol=0)
        tab = R["mp_ref"][b, :N].astype(np.int64)
        held = np.full(N, -1, np.int64)
        a = (tab >= 0) & (tab < P.kc)
        held[a] = pr["mps"]["key_mp"][tab[a]]
        held[tab >= P.kc] = tab[tab >= P.kc] - P.kc
        assert np.array_equal(held, ref["mp_ref"]), b
        for name, got in (("r1", R["r1"][b]), ("r2", R["r2"][b])):
            dt, dr = synth_ba.pose_error(ref[name]["base"]["nav"], got["base"]["nav"])
            assert dt < 1e-4 and dr < 1e-4, (b, name, dt, dr)
            assert int(ref[name]["base"]["n_inliers"]) == int(got["base"]["n_inliers"]), (b, name)
    for b in range(B):  # every frame of the batch is tracked
        gdt, gdr = synth_ba.pose_error(R["r2"][b]["base"]["nav"], P.truth[b])
        assert gdt < 1e-2 and gdr < 5e-3 and int(R["r2"][b]["base"]["n_inliers"]) > 60, (b, gdt, gdr)
    P.close_all()
