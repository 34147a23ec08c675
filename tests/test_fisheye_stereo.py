"""Frame::ComputeStereoFishEyeMatches (a10, reference src/Frame.cc:613-779): oracle known-answer tests (CPU) and
HIP-vs-oracle parity (GPU): the group tables (integers) must be identical, 3-D points / depths agree to rounding."""
import numpy as np
import pytest

from vieo_slam_amd import synth_ba, synth_fisheye


# ------------------------------------------------------------------ oracle KATs (CPU)
@pytest.mark.parametrize("rig", ["radtan", "kb8"])
def test_unproject_inverts_project(oracle, rig):
    cams, (W, H) = synth_ba.camera_rig(rig)
    rng = np.random.default_rng(3)
    for c in cams:
        for _ in range(50):
            uv = np.array([rng.uniform(10, W - 10), rng.uniform(10, H - 10)], np.float32)
            if rig == "kb8" and np.hypot(uv[0] - c["cx"], uv[1] - c["cy"]) > 230:
                continue  # theta_d beyond ~1.2 rad: the plane form (tan theta) is not meant for the fisheye rim
            P = oracle.cam_unproject(c, uv)
            assert P[2] == 1.0
            u, v = synth_ba.project_camera(c, P * rng.uniform(0.5, 20))
            assert abs(u - uv[0]) < 2e-3 and abs(v - uv[1]) < 2e-3, (uv, u, v)


def test_null_vector_matches_numpy_svd(oracle):
    rng = np.random.default_rng(5)
    for m in (4, 6, 8):
        for _ in range(20):
            x = rng.normal(0, 1, 4)
            A = rng.normal(0, 1, (m, 4))
            A -= np.outer(A @ x, x) / (x @ x)           # exact null vector x ...
            A += rng.normal(0, 1e-6, A.shape)           # ... up to noise
            v = oracle.null_vector4(A)
            ref = np.linalg.svd(A)[2][3]
            assert min(np.linalg.norm(v - ref), np.linalg.norm(v + ref)) < 1e-9
    # rank-deficient by more than one: still a unit vector in the null space
    A = np.zeros((4, 4))
    A[0, 0] = A[1, 1] = 1
    v = oracle.null_vector4(A)
    assert abs(np.linalg.norm(v) - 1) < 1e-12 and np.allclose(A @ v, 0)


def _check_truth(case, out, min_frac):
    """matched keys carry the depth of their scene point in their own camera"""
    nc = len(case["keys"])
    off = np.cumsum([0] + [len(k) for k in case["keys"]])
    n_ok = n_bad = 0
    for c in range(nc):
        Tcr = case["Tcr"][c].reshape(3, 4)
        for k, p in enumerate(case["owner"][c]):
            d = out["depth"][off[c] + k]
            if d < 0:
                continue
            if p < 0:
                n_bad += 1
                continue
            z = Tcr[2, :3] @ case["X"][p] + Tcr[2, 3]
            if abs(d - z) < (0.1 + 0.1 * z) * z:  # depth uncertainty grows with z / (f * baseline)
                n_ok += 1
            else:
                n_bad += 1
    assert n_ok > min_frac * (n_ok + n_bad) and n_ok >= 30, (n_ok, n_bad)
    return n_ok, n_bad


@pytest.mark.parametrize("rig,seed", [("radtan", 1), ("kb8", 2)])
def test_oracle_recovers_scene_depths(oracle, rig, seed):
    # (the 0.9998 parallax gate of an 11 cm baseline ends near 5.5 m)
    case = synth_fisheye.make_fisheye_case(seed, rig=rig, n_points=300, duplicates=0.0, far_frac=0.0, z_max=5.0,
                                           noise=0.05)
    out = oracle.stereo_fisheye(case["params"], case["keys"], case["descs"], case["num_mono"])
    _check_truth(case, out, 0.95)
    G = out["group_idx"]
    assert len(G) > 50 and out["n_matches"] > 50
    # table invariants: every mapped key is the member its group lists; good groups have >= 2 members
    off = np.cumsum([0] + [len(k) for k in case["keys"]])
    for c in range(len(case["keys"])):
        for k in range(len(case["keys"][c])):
            g = out["key_group"][off[c] + k]
            if g >= 0:
                assert G[g, c] == k
    assert ((G >= 0).sum(1)[out["group_good"]] >= 2).all()


def test_oracle_far_points_and_second_threshold(oracle):
    # only far points: the 0.9998 gate rejects (almost) everything, the second pass (1 - 1e-6) keeps more
    case = synth_fisheye.make_fisheye_case(7, rig="radtan", n_points=200, far_frac=1.0, duplicates=0.0)
    out = oracle.stereo_fisheye(case["params"], case["keys"], case["descs"], case["num_mono"])
    case2 = synth_fisheye.make_fisheye_case(7, rig="radtan", n_points=200, far_frac=1.0, duplicates=0.0,
                                            th_far_pts=20.0)
    out2 = oracle.stereo_fisheye(case2["params"], case2["keys"], case2["descs"], case2["num_mono"])
    assert out2["n_matches"] <= out["n_matches"]  # a far-point limit tightens both thresholds


def test_generator_reaches_every_bookkeeping_branch(oracle):
    """the cases of the parity test drive the replacement and both contradiction branches of FillMatchesFromPair"""
    oracle.fisheye_branch_counts()
    tot = np.zeros(5, np.int64)
    for seed, kw in ((14, dict(duplicates=0.3)), (18, dict(flip_bits=60, duplicates=0.4))):
        case = synth_fisheye.make_fisheye_case(seed, rig="kb8", **kw)
        oracle.stereo_fisheye(case["params"], case["keys"], case["descs"], case["num_mono"])
        tot += np.array(oracle.fisheye_branch_counts())
    assert (tot > 0).all(), tot


# ------------------------------------------------------------------ parity (GPU)
def _parity(oracle, case):
    from vieo_slam_amd.matching import compute_stereo_fisheye_matches
    o = oracle.stereo_fisheye(case["params"], case["keys"], case["descs"], case["num_mono"])
    h = compute_stereo_fisheye_matches(case["params"], case["keys"], case["descs"], case["num_mono"])
    assert np.array_equal(o["group_idx"], h["group_idx"])
    assert np.array_equal(o["group_good"], h["group_good"])
    assert np.array_equal(o["key_group"], h["key_group"])
    assert o["n_matches"] == h["n_matches"]
    good = o["group_good"]
    if good.any():
        a, b = o["group_p3d"][good], h["group_p3d"][good]
        assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(a).max())  # FP64, same algorithm; libm vs device trig
    assert np.array_equal(o["depth"] < 0, h["depth"] < 0)
    assert np.allclose(o["depth"], h["depth"], rtol=1e-6, atol=0)
    return o, h


@pytest.mark.gpu
@pytest.mark.parametrize("rig,seed,kw", [
    ("radtan", 11, {}), ("radtan", 12, dict(th_far_pts=30.0)), ("kb8", 13, {}), ("kb8", 14, dict(duplicates=0.3)),
    ("kb8", 15, dict(n_points=1500, distractors=0.5)), ("kb8", 16, dict(num_mono=[5, 0, 17, 3])),
    ("radtan", 17, dict(far_frac=1.0)), ("kb8", 18, dict(flip_bits=60, duplicates=0.4)),
])
def test_fisheye_stereo_parity(oracle, rig, seed, kw):
    case = synth_fisheye.make_fisheye_case(seed, rig=rig, **kw)
    o, h = _parity(oracle, case)
    if not kw.get("far_frac"):
        _check_truth(case, h, 0.8)


@pytest.mark.gpu
def test_fisheye_stereo_edge_cases(oracle):
    from vieo_slam_amd._lib import lib
    from vieo_slam_amd.matching import compute_stereo_fisheye_matches, fisheye_call
    case = synth_fisheye.make_fisheye_case(21, rig="kb8", n_points=120)
    # a camera without usable rows (num_mono >= n): its pairs are skipped (Frame.cc:623)
    case["num_mono"] = np.array([0, len(case["keys"][1]), 0, 0], np.int32)
    _parity(oracle, case)
    # an empty camera
    case = synth_fisheye.make_fisheye_case(22, rig="kb8", n_points=120)
    case["keys"][2] = case["keys"][2][:0]
    case["descs"][2] = case["descs"][2][:0]
    _parity(oracle, case)
    # a single train row: knnMatch returns one neighbour, size() >= 2 fails
    case = synth_fisheye.make_fisheye_case(23, rig="radtan", n_points=60)
    case["keys"][1], case["descs"][1] = case["keys"][1][:1], case["descs"][1][:1]
    o, h = _parity(oracle, case)
    assert len(h["group_idx"]) == 0 and (h["depth"] < 0).all()
    # capacity
    case = synth_fisheye.make_fisheye_case(24, rig="kb8", n_points=200)
    rc, _ = fisheye_call(lib().vieo_stereo_fisheye_match, case["params"], case["keys"], case["descs"],
                         case["num_mono"], group_capacity=3)
    assert rc != 0


@pytest.mark.gpu
def test_fisheye_device_batch_matches_oracle_and_walks_in_parallel(oracle):
    """The device-resident stage (vieo_stereo_fisheye_match_batch_device) on a batch of different rig frames at once --
    2-camera Radtan, 4-camera KB8, frames with look-alike descriptors (replacement / contradiction branches) -- against
    the oracle frame by frame, mvKeys / mDescriptors included; the speculative-parallel walk of the group tables must
    take far fewer wavefront steps than rows on ordinary frames."""
    from vieo_slam_amd.matching import FisheyeStereoDevice, compute_stereo_fisheye_matches, fisheye_last_walk
    for rig, seeds, kws in (("radtan", (31, 32, 33), ({}, dict(duplicates=0.3), dict(n_points=900, distractors=0.5))),
                            ("kb8", (34, 35, 36), ({}, dict(flip_bits=60, duplicates=0.4), dict(num_mono=[5, 0, 17, 3])))):
        cases = [synth_fisheye.make_fisheye_case(s, rig=rig, **kw) for s, kw in zip(seeds, kws)]
        cap = max(len(k) for c in cases for k in c["keys"]) + 7
        dev = FisheyeStereoDevice(cases[0]["params"], cap, max_frames=len(cases))
        outs = dev.match_batch([(c["keys"], c["descs"], c["num_mono"]) for c in cases])
        for c, h in zip(cases, outs):
            o = oracle.stereo_fisheye(c["params"], c["keys"], c["descs"], c["num_mono"])
            assert h["hdr"][3] == 0
            assert np.array_equal(o["group_idx"], h["group_idx"]) and np.array_equal(o["group_good"], h["group_good"])
            assert np.array_equal(o["key_group"], h["key_group"]) and o["n_matches"] == h["n_matches"]
            good = o["group_good"]
            if good.any():
                a, b = o["group_p3d"][good], h["group_p3d"][good]
                assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(a).max())
            assert np.array_equal(o["depth"] < 0, h["depth"] < 0) and np.allclose(o["depth"], h["depth"], rtol=1e-6, atol=0)
            assert np.array_equal(h["keys"].view(np.uint8), np.concatenate(c["keys"]).view(np.uint8))
            assert np.array_equal(h["desc"], np.concatenate(c["descs"])) and (h["uright"] == -1).all()
            assert np.array_equal(h["cam_first"], np.concatenate([[0], np.cumsum([len(k) for k in c["keys"]])]))
        # an ordinary frame: most rows are independent
        rows, steps = int(outs[0]["hdr"][5]), int(outs[0]["hdr"][6])
        assert rows >= 30 and 0 < steps <= rows // 8 + 8, (rig, rows, steps)
        dev.close()
    # the host-pointer entry reports its walk as well
    c = synth_fisheye.make_fisheye_case(37, rig="kb8")
    compute_stereo_fisheye_matches(c["params"], c["keys"], c["descs"], c["num_mono"])
    rows, steps = fisheye_last_walk()
    assert rows > 100 and 0 < steps < rows


@pytest.mark.gpu
def test_fisheye_device_batch_ragged_frames(oracle):
    """One batch whose frames differ in what they have: a full frame, a frame with an empty camera, a frame whose
    second camera has no usable rows (num_mono >= n) and a frame with a single key in camera 1 (knnMatch returns one
    neighbour) -- each equals its own oracle run."""
    from vieo_slam_amd.matching import FisheyeStereoDevice
    cases = [synth_fisheye.make_fisheye_case(41 + i, rig="kb8", n_points=150) for i in range(4)]
    cases[1]["keys"][2], cases[1]["descs"][2] = cases[1]["keys"][2][:0], cases[1]["descs"][2][:0]
    cases[2]["num_mono"] = np.array([0, len(cases[2]["keys"][1]), 0, 0], np.int32)
    cases[3]["keys"][1], cases[3]["descs"][1] = cases[3]["keys"][1][:1], cases[3]["descs"][1][:1]
    cap = max(len(k) for c in cases for k in c["keys"]) + 3
    dev = FisheyeStereoDevice(cases[0]["params"], cap, max_frames=4)
    outs = dev.match_batch([(c["keys"], c["descs"], c["num_mono"]) for c in cases])
    for c, h in zip(cases, outs):
        o = oracle.stereo_fisheye(c["params"], c["keys"], c["descs"], c["num_mono"])
        assert np.array_equal(o["group_idx"], h["group_idx"]) and np.array_equal(o["group_good"], h["group_good"])
        assert np.array_equal(o["key_group"], h["key_group"]) and o["n_matches"] == h["n_matches"]
        assert np.array_equal(o["depth"] < 0, h["depth"] < 0) and np.allclose(o["depth"], h["depth"], rtol=1e-6, atol=0)
    dev.close()
