"""Projection search (tracking-side ORBmatcher::SearchByProjection): CPU known-answer tests of
the oracle and GPU parity tests (bit-exact assignments)."""
import numpy as np
import pytest

from vieo_slam_amd import frontend, synth
from vieo_slam_amd.ba_types import PROJ_QUERY_DTYPE

K = (synth.EUROC_FX, synth.EUROC_FX, 367.4517211914062, 252.2008514404297)
BF, BASELINE = synth.EUROC_BF, synth.EUROC_BF / synth.EUROC_FX
BOUNDS = np.array([0, 752, 0, 480], np.float32)


def _frame(oracle, seed):
    """keys/desc/uright/depth of a synthetic stereo frame (via the oracle)."""
    left, right, _ = synth.synth_stereo_pair(seed)
    eL, eR = oracle.extractor(1200), oracle.extractor(1200)
    _, kl, dl = eL(left)
    _, kr, dr = eR(right)
    ur, dp = oracle.stereo_match(eL, eR, kl, dl, kr, dr, BASELINE, BF)
    return kl, dl, ur, dp, np.array(eL.scale_factors(), np.float32)


def _scenario(oracle, seed, th=7.0, dt=(0.02, -0.01, 0.03), observed=True):
    """current frame = stereo frame `seed`; 'last frame' = the same keys unprojected with the
    identity pose, current pose = small translation -> projections land near their keys."""
    kl, dl, ur, dp, sc = _frame(oracle, seed)
    Xw, ok = frontend.unproject_stereo(kl, dp, K, np.eye(3), np.zeros(3))
    pts = frontend.make_last_frame_points(kl, dl, Xw, ok, observed)
    Tl = frontend.pose_to_Tcw(np.eye(3), np.zeros(3))
    Tc = frontend.pose_to_Tcw(np.eye(3), np.array(dt))
    cam = frontend.make_sbp_camera(Tc, Tl, K, BOUNDS, BF, BASELINE, th, sc)
    return kl, dl, ur, pts, cam


def test_oracle_sbp_last_frame_finds_own_keys(oracle):
    kl, dl, ur, pts, cam = _scenario(oracle, 1000, dt=(0.0, 0.0, 0.0))
    q = oracle.sbp_project_last_frame(pts, cam)
    valid = (q["flags"] & 1) > 0
    assert valid.sum() > 300
    # zero motion: every valid query projects onto its own key (float32 round trip)
    assert np.max(np.abs(q["u"][valid] - kl["x"][valid])) < 1e-2
    n, assign = oracle.search_by_projection(0, q, kl, ur, dl, None, BOUNDS)
    got = np.nonzero(assign >= 0)[0]
    assert n == len(got) and n > 0.9 * valid.sum()
    assert np.mean(assign[got] == got) > 0.98  # matched to itself (distance 0)
    assert np.all(q["level_min"][valid] == kl["octave"][valid] - 1)


def test_oracle_sbp_forward_backward_levels(oracle):
    kl, dl, ur, pts, cam = _scenario(oracle, 1001, dt=(0, 0, 0.0))
    for tz, lo, hi in ((0.5, "zero", "oct"), (-0.5, "oct", "none")):
        Tc = frontend.pose_to_Tcw(np.eye(3), np.array([0, 0, tz]))
        cam[0]["Tcw_cur"] = Tc.reshape(-1)
        q = oracle.sbp_project_last_frame(pts, cam)
        v = (q["flags"] & 1) > 0
        # camera moved along +z of the last frame: tlrcr.z = +tz (Tlrcr = Tlrw * Tcrw^-1)
        if lo == "zero":
            assert np.all(q["level_min"][v] == 0) and np.all(q["level_max"][v] == pts["octave"][v])
        else:
            assert np.all(q["level_min"][v] == pts["octave"][v]) and np.all(q["level_max"][v] == -1)


def test_oracle_sequential_claiming_and_rotation_filter(oracle):
    kl, dl, ur, pts, cam = _scenario(oracle, 1002)
    q = oracle.sbp_project_last_frame(pts, cam)
    n, a = oracle.search_by_projection(0, q, kl, ur, dl, None, BOUNDS)
    used = a[a >= 0]
    assert len(used) == len(set(used.tolist()))  # a query is placed at most once
    # rotation check off: no erasures, count >= the filtered count
    n2, a2 = oracle.search_by_projection(0, q, kl, ur, dl, None, BOUNDS, check_ori=False)
    assert not np.any(a2 == -2) and n2 >= n
    # pre-claimed keypoints are never overwritten
    taken = np.zeros(len(kl), np.uint8)
    taken[::3] = 1
    n3, a3 = oracle.search_by_projection(0, q, kl, ur, dl, taken, BOUNDS)
    assert np.all(a3[::3] == -1)
    # unobserved (temporal) map points do not claim: later queries may overwrite the slot
    pts0 = pts.copy()
    pts0["flags"] = (pts0["flags"] & 1)
    q0 = oracle.sbp_project_last_frame(pts0, cam)
    n4, a4 = oracle.search_by_projection(0, q0, kl, ur, dl, None, BOUNDS, check_ori=False)
    assert n4 >= n2


def test_oracle_local_map_ratio_rule(oracle):
    kl, dl, ur, pts, cam = _scenario(oracle, 1003)
    q = oracle.sbp_project_last_frame(pts, cam)
    q["level_min"] = pts["octave"] - 1  # isInFrustum-style windows: levels [L-1, L]
    q["level_max"] = pts["octave"]
    n_loose, _ = oracle.search_by_projection(1, q, kl, ur, dl, None, BOUNDS, nn_ratio=1.0)
    n_tight, _ = oracle.search_by_projection(1, q, kl, ur, dl, None, BOUNDS, nn_ratio=0.3)
    assert n_loose >= n_tight > 0


# ------------------------------------------------------------------ GPU parity
def _hip_matcher(nn=0.6, ori=True):
    from vieo_slam_amd.matching import ORBmatcher
    return ORBmatcher(nn, ori)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th,dt", [(1000, 7.0, (0.02, -0.01, 0.03)), (1001, 15.0, (0.05, 0.02, 0.3)),
                                        (1002, 7.0, (0.0, 0.0, -0.4)), (1003, 14.0, (-0.1, 0.05, 0.0))])
def test_gpu_sbp_last_frame_parity(oracle, seed, th, dt):
    kl, dl, ur, pts, cam = _scenario(oracle, seed, th, dt)
    oq = oracle.sbp_project_last_frame(pts, cam)
    m = _hip_matcher()
    hq = m.project_last_frame(pts, cam)
    assert np.array_equal(oq.view(np.uint8), hq.view(np.uint8))
    for taken in (None, (np.arange(len(kl)) % 5 == 0).astype(np.uint8)):
        on, oa = oracle.search_by_projection(0, oq, kl, ur, dl, taken, BOUNDS)
        hn, ha = m.SearchByProjectionLastFrame(hq, kl, ur, dl, taken, BOUNDS)
        assert on == hn and np.array_equal(oa, ha)
        assert on > 100
    on, oa = oracle.search_by_projection(0, oq, kl, ur, dl, None, BOUNDS, check_ori=False)
    hn, ha = _hip_matcher(0.6, False).SearchByProjectionLastFrame(hq, kl, ur, dl, None, BOUNDS)
    assert on == hn and np.array_equal(oa, ha)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,nn", [(1004, 0.8), (1005, 0.6), (1006, 0.9)])
def test_gpu_sbp_local_map_parity(oracle, seed, nn):
    kl, dl, ur, pts, cam = _scenario(oracle, seed, th=1.0)
    q = oracle.sbp_project_last_frame(pts, cam)
    rng = np.random.default_rng(seed)
    q["level_min"] = pts["octave"] - 1
    q["level_max"] = pts["octave"]
    q["radius"] = np.where(rng.random(len(q)) < 0.5, 2.5, 4.0).astype(np.float32) * cam[0]["scale"][pts["octave"]] * 3
    q = q[rng.permutation(len(q))]  # local map points come in arbitrary order
    for taken in (None, (rng.random(len(kl)) < 0.3).astype(np.uint8)):
        on, oa = oracle.search_by_projection(1, q, kl, ur, dl, taken, BOUNDS, nn_ratio=nn)
        hn, ha = _hip_matcher(nn).SearchByProjectionLocalMap(q, kl, ur, dl, taken, BOUNDS)
        assert on == hn and np.array_equal(oa, ha)
        assert on > 50


@pytest.mark.gpu
def test_gpu_sbp_edge_cases(oracle):
    kl, dl, ur, pts, cam = _scenario(oracle, 1007)
    m = _hip_matcher()
    q = oracle.sbp_project_last_frame(pts, cam)
    # no queries / no valid queries / all keys claimed
    hn, ha = m.SearchByProjectionLastFrame(q[:0], kl, ur, dl, None, BOUNDS)
    assert hn == 0 and np.all(ha == -1)
    q0 = q.copy()
    q0["flags"] = 0
    hn, ha = m.SearchByProjectionLastFrame(q0, kl, ur, dl, None, BOUNDS)
    assert hn == 0 and np.all(ha == -1)
    taken = np.ones(len(kl), np.uint8)
    on, oa = oracle.search_by_projection(0, q, kl, ur, dl, taken, BOUNDS)
    hn, ha = m.SearchByProjectionLastFrame(q, kl, ur, dl, taken, BOUNDS)
    assert on == hn == 0 and np.array_equal(oa, ha)
    # windows hanging over the image border
    q2 = q.copy()
    q2["u"] = np.where(np.arange(len(q)) % 2 == 0, 2.0, 750.0)
    on, oa = oracle.search_by_projection(0, q2, kl, ur, dl, None, BOUNDS)
    hn, ha = m.SearchByProjectionLastFrame(q2, kl, ur, dl, None, BOUNDS)
    assert on == hn and np.array_equal(oa, ha)


def test_oracle_reloc_variant_rules(oracle):
    """a14 (ORBmatcher.cc:1471-1606): keys holding any map point are skipped even when that point has no
    observations, the stereo coordinate is not checked, acceptance is dist <= ORBdist."""
    kl, dl, ur, pts, cam = _scenario(oracle, 1010, observed=False)
    q = oracle.sbp_project_last_frame(pts, cam)
    q["level_min"], q["level_max"] = pts["octave"] - 1, pts["octave"] + 1
    n0, a0 = oracle.search_by_projection(2, q, kl, ur, dl, None, BOUNDS, nn_ratio=100.0)
    assert n0 > 100
    # mode 0 lets a second query re-claim a key whose holder has no observations; mode 2 never does
    claimed = a0[a0 >= 0]
    assert len(np.unique(claimed)) == len(claimed)
    # a stricter ORBdist only removes matches
    n1, a1 = oracle.search_by_projection(2, q, kl, ur, dl, None, BOUNDS, nn_ratio=30.0, check_ori=False)
    n2, a2 = oracle.search_by_projection(2, q, kl, ur, dl, None, BOUNDS, nn_ratio=100.0, check_ori=False)
    assert n1 <= n2 and np.all((a1 < 0) | (a1 == a2))
    # ruining the stereo coordinates changes nothing (no gate in this variant)
    n3, a3 = oracle.search_by_projection(2, q, kl, np.full_like(ur, 5.0), dl, None, BOUNDS, nn_ratio=100.0)
    assert n3 == n0 and np.array_equal(a3, a0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,orbdist", [(1011, 100), (1012, 64), (1013, 50)])
def test_gpu_sbp_reloc_parity(oracle, seed, orbdist):
    kl, dl, ur, pts, cam = _scenario(oracle, seed, th=10.0, observed=(seed % 2 == 0))
    q = oracle.sbp_project_last_frame(pts, cam)
    q["level_min"], q["level_max"] = pts["octave"] - 1, pts["octave"] + 1
    rng = np.random.default_rng(seed)
    m = _hip_matcher()
    for taken in (None, (rng.random(len(kl)) < 0.4).astype(np.uint8)):
        on, oa = oracle.search_by_projection(2, q, kl, ur, dl, taken, BOUNDS, nn_ratio=float(orbdist))
        hn, ha = m.SearchByProjectionKeyFrame(q, kl, ur, dl, taken, BOUNDS, orbdist)
        assert on == hn and np.array_equal(oa, ha)
        assert on > 30


@pytest.mark.gpu
def test_gpu_sbp_window_size_classes_and_tiny_batches(oracle):
    """k_sbp_candidates takes four queries per wavefront when a window spans at most 16 grid columns and 64 entries and
    a wavefront per query otherwise: one frame that mixes narrow windows, windows wider than 16 columns and windows
    with more than 64 entries, then query counts that do not fill a wavefront's four rows (1, 3, 17)."""
    kl, dl, ur, pts, cam = _scenario(oracle, 1030, th=7.0)
    q = oracle.sbp_project_last_frame(pts, cam)
    rng = np.random.default_rng(1030)
    kind = rng.integers(0, 3, len(q))
    # narrow windows over every level / 13 columns with the projection's own level range (more than 64 entries, a
    # third of them candidates) / 18 columns likewise (the library keeps at most 128 candidates per query)
    q["level_min"] = np.where(kind == 0, 0, q["level_min"])
    q["level_max"] = np.where(kind == 0, -1, q["level_max"])
    q["radius"] = np.where(kind == 0, 6.0, np.where(kind == 1, 70.0, 100.0)).astype(np.float32)
    m = _hip_matcher(0.9)
    on, oa = oracle.search_by_projection(0, q, kl, ur, dl, None, BOUNDS, nn_ratio=0.9)
    hn, ha = m.SearchByProjectionLastFrame(q, kl, ur, dl, None, BOUNDS)
    assert on == hn and np.array_equal(oa, ha) and on > 50
    for n in (1, 3, 17):
        on, oa = oracle.search_by_projection(0, q[:n], kl, ur, dl, None, BOUNDS, nn_ratio=0.9)
        hn, ha = m.SearchByProjectionLastFrame(q[:n], kl, ur, dl, None, BOUNDS)
        assert on == hn and np.array_equal(oa, ha)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_gpu_sbp_assignment_dependency_chains(oracle, mode, monkeypatch):
    """The optimistic-parallel assignment (k_sbp_assign_par) against the sequential semantics where they bite: look-alike
    descriptors and wide windows make every query want the keys its predecessors take, so the claims settle only after
    many rounds.  Three ways through the library -- the parallel fixed point, the sequential replay alone
    (VIEO_SBP_ASSIGN=seq), the parallel form giving up after one round and handing the frame to the replay -- all equal the
    oracle bit for bit; unobserved points (no blocking) and pre-claimed keys mixed in."""
    kl, dl, ur, pts, cam = _scenario(oracle, 1020 + mode, th=25.0 if mode != 1 else 6.0)
    rng = np.random.default_rng(77 + mode)
    # a few descriptor families: within a family the distances are 0..6 bits, so the order of claims decides who gets what
    fam = rng.integers(0, 256, (12, 32), dtype=np.uint8)
    dl2 = fam[rng.integers(0, 12, len(kl))].copy()
    flip = rng.integers(0, 256, len(kl))
    dl2[np.arange(len(kl)), flip % 32] ^= (1 << (flip % 8)).astype(np.uint8)
    pts2 = pts.copy()
    pts2["desc"] = fam[rng.integers(0, 12, len(pts))]
    unobs = rng.random(len(pts)) < 0.15
    pts2["flags"] = np.where(unobs, pts2["flags"] & 1, pts2["flags"])
    q = oracle.sbp_project_last_frame(pts2, cam)
    if mode == 1:
        q["level_min"], q["level_max"] = pts["octave"] - 1, pts["octave"]
        q["radius"] = (np.float32(4.0) * cam[0]["scale"][pts["octave"]] * 6).astype(np.float32)
        q = q[rng.permutation(len(q))]
    elif mode == 2:
        q["level_min"], q["level_max"] = pts["octave"] - 1, pts["octave"] + 1
    taken = (rng.random(len(kl)) < 0.1).astype(np.uint8)
    m = _hip_matcher(0.9 if mode != 1 else 0.95)
    call = {0: m.SearchByProjectionLastFrame, 1: m.SearchByProjectionLocalMap,
            2: lambda *a: m.SearchByProjectionKeyFrame(*a, 100)}[mode]
    on, oa = oracle.search_by_projection(mode, q, kl, ur, dl2, taken, BOUNDS, nn_ratio=(0.9, 0.95, 100.0)[mode])
    assert on > 100
    results = {}
    # (the parallel form exists twice: flat over the candidate pairs -- the default for a call of few frames and for
    # rigs -- and one thread per query for large batches; VIEO_SBP_FLAT picks)
    for name, env in (("parallel", {}), ("flat", {"VIEO_SBP_FLAT": "1"}), ("per query", {"VIEO_SBP_FLAT": "0"}),
                      ("sequential", {"VIEO_SBP_ASSIGN": "seq"}), ("fallback", {"VIEO_SBP_MAX_ROUNDS": "1"}),
                      ("fallback per query", {"VIEO_SBP_MAX_ROUNDS": "1", "VIEO_SBP_FLAT": "0"})):
        monkeypatch.delenv("VIEO_SBP_ASSIGN", raising=False)
        monkeypatch.delenv("VIEO_SBP_MAX_ROUNDS", raising=False)
        monkeypatch.delenv("VIEO_SBP_FLAT", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        results[name] = call(q, kl, ur, dl2, taken, BOUNDS)
    for name, (hn, ha) in results.items():
        assert hn == on and np.array_equal(ha, oa), name


@pytest.mark.gpu
def test_gpu_sbp_many_keys_in_one_camera(oracle, monkeypatch):
    """A one-camera frame with ~7 000 keys: its per-key state alone fills the LDS the flat assignment would need for the
    candidate lists, so the call takes the thread-per-query form even when the flat one is asked for -- same answer."""
    kl, dl, ur, pts, cam = _scenario(oracle, 1050, th=7.0)
    reps = max(2, 7000 // len(kl) + 1)
    rng = np.random.default_rng(1050)
    kl2, dl2, ur2 = np.tile(kl, reps), np.tile(dl, (reps, 1)), np.tile(ur, reps)
    kl2["x"] += rng.uniform(-1.5, 1.5, len(kl2)).astype(np.float32)
    kl2["y"] += rng.uniform(-1.5, 1.5, len(kl2)).astype(np.float32)
    flip = rng.integers(0, 256, len(dl2))
    dl2[np.arange(len(dl2)), flip % 32] ^= (1 << (flip % 8)).astype(np.uint8)
    assert 6500 < len(kl2) <= 8192
    q = oracle.sbp_project_last_frame(pts, cam)
    on, oa = oracle.search_by_projection(0, q, kl2, ur2, dl2, None, BOUNDS, nn_ratio=0.9)
    assert on > 100
    m = _hip_matcher(0.9)
    for env in ({}, {"VIEO_SBP_FLAT": "1"}, {"VIEO_SBP_FLAT": "0"}):
        monkeypatch.delenv("VIEO_SBP_FLAT", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        hn, ha = m.SearchByProjectionLastFrame(q, kl2, ur2, dl2, None, BOUNDS)
        assert hn == on and np.array_equal(ha, oa), env
    monkeypatch.delenv("VIEO_SBP_FLAT", raising=False)


@pytest.mark.gpu
def test_gpu_sbp_more_wide_windows_than_a_block_lists(oracle, monkeypatch):
    """k_sbp_candidates leaves wide / crowded windows to a second pass through a per-block list of 1024 entries; a frame
    with more of them than the eight blocks can list (relocalisation-size windows, rig local maps with p_cap * n_cams
    queries) makes every block walk ITS OWN share of the queries again.  19 000 queries, all of them wider than 16 grid
    columns, against the oracle and against the sequential replay."""
    monkeypatch.setenv("VIEO_SBP_BLOCKS", "8")  # (a single frame would get 32 blocks: 33 000 wide queries to overflow)
    kl, dl, ur, pts, cam = _scenario(oracle, 1040, th=7.0)
    q1 = oracle.sbp_project_last_frame(pts, cam)
    rng = np.random.default_rng(1040)
    q = np.tile(q1, 16)
    q["u"] += rng.uniform(-30, 30, len(q)).astype(np.float32)
    q["v"] += rng.uniform(-30, 30, len(q)).astype(np.float32)
    q["radius"] = rng.uniform(96.0, 110.0, len(q)).astype(np.float32)
    oc = np.tile(pts["octave"], 16)
    q["level_min"], q["level_max"] = oc, oc           # one level: ~20 candidates out of a window of ~130 keys
    q["flags"] = np.where(rng.random(len(q)) < 0.5, q["flags"] & ~2, q["flags"])
    assert ((q["flags"] & 1) > 0).sum() > 8 * 1024 + 1500
    m = _hip_matcher(0.9)
    on, oa = oracle.search_by_projection(0, q, kl, ur, dl, None, BOUNDS, nn_ratio=0.9)
    assert on > 100
    for env in ({}, {"VIEO_SBP_FLAT": "0"}, {"VIEO_SBP_ASSIGN": "seq"}):  # (19 000 queries: the flat form hands over to the replay)
        monkeypatch.delenv("VIEO_SBP_ASSIGN", raising=False)
        monkeypatch.delenv("VIEO_SBP_FLAT", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        hn, ha = m.SearchByProjectionLastFrame(q, kl, ur, dl, None, BOUNDS)
        assert hn == on and np.array_equal(ha, oa), env
    monkeypatch.delenv("VIEO_SBP_ASSIGN", raising=False)
    monkeypatch.delenv("VIEO_SBP_FLAT", raising=False)
    # and with only three of the eight blocks over their list (block b owns the queries with (q >> 4) % 8 == b)
    q2 = q.copy()
    q2["radius"] = np.where(((np.arange(len(q)) >> 4) % 8) < 3, q["radius"], 6.0).astype(np.float32)
    on, oa = oracle.search_by_projection(0, q2, kl, ur, dl, None, BOUNDS, nn_ratio=0.9)
    hn, ha = m.SearchByProjectionLastFrame(q2, kl, ur, dl, None, BOUNDS)
    assert hn == on and np.array_equal(ha, oa)
