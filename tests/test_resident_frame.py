"""The resident frame (include/vieo_hot.h, round 5): the entries that read a frame's keys / descriptors / pyramid /
uright where vieo_orb_extract left them must return what the host-pointer entries and the oracle return, bit for bit,
and must refuse a handle that holds another frame."""
import numpy as np
import pytest

from vieo_slam_amd import frontend, synth

K = (synth.EUROC_FX, synth.EUROC_FX, 367.4517211914062, 252.2008514404297)
BF, BASELINE = synth.EUROC_BF, synth.EUROC_BF / synth.EUROC_FX
BOUNDS = np.array([0, 752, 0, 480], np.float32)


def _hip_frame(seed, nfeat=1200):
    from vieo_slam_amd.orb_extractor import ORBextractor
    left, right, _ = synth.synth_stereo_pair(seed)
    eL, eR = ORBextractor(nfeat, 1.2, 8, 20, 7), ORBextractor(nfeat, 1.2, 8, 20, 7)
    _, kl, dl = eL(left)
    _, kr, dr = eR(right)
    return eL, eR, kl, dl, kr, dr


def _points_and_cam(kl, dl, dp, sc, th=7.0, dt=(0.02, -0.01, 0.03)):
    Xw, ok = frontend.unproject_stereo(kl, dp, K, np.eye(3), np.zeros(3))
    pts = frontend.make_last_frame_points(kl, dl, Xw, ok, True)
    Tl = frontend.pose_to_Tcw(np.eye(3), np.zeros(3))
    Tc = frontend.pose_to_Tcw(np.eye(3), np.array(dt))
    return pts, frontend.make_sbp_camera(Tc, Tl, K, BOUNDS, BF, BASELINE, th, sc)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1000, 1003])
def test_gpu_resident_stereo_and_searches_equal_host_pointer_entries_and_oracle(oracle, seed):
    from vieo_slam_amd.matching import ORBmatcher, compute_stereo_matches, compute_stereo_matches_resident
    eL, eR, kl, dl, kr, dr = _hip_frame(seed)
    assert eL.holds(kl) and eR.holds(kr) and eL.resident_keys() == len(kl)
    assert not eL.holds(kr) and not eL.holds(kl[:-1])
    ur_r, dp_r = compute_stereo_matches_resident(eL, eR, BASELINE, BF)
    ur_h, dp_h = compute_stereo_matches(eL, eR, kl, dl, kr, dr, BASELINE, BF)
    assert np.array_equal(ur_r.view(np.uint32), ur_h.view(np.uint32)) and np.array_equal(dp_r.view(np.uint32), dp_h.view(np.uint32))
    oL, oR = oracle.extractor(1200), oracle.extractor(1200)
    left, right, _ = synth.synth_stereo_pair(seed)
    _, okl, odl = oL(left)
    _, okr, odr = oR(right)
    our, odp = oracle.stereo_match(oL, oR, okl, odl, okr, odr, BASELINE, BF)
    assert np.array_equal(ur_r.view(np.uint32), our.view(np.uint32))
    sc = np.asarray(eL.GetScaleFactors(), np.float32)
    M = ORBmatcher(0.9, True)
    for th, dt in ((7.0, (0.02, -0.01, 0.03)), (15.0, (0.05, 0.02, -0.2))):
        pts, cam = _points_and_cam(kl, dl, dp_r, sc, th, dt)
        # (1) the last-frame overload as one call, twice: the second call reads the kept grid
        for rep in range(2):
            n_r, a_r = M.search_last_frame_resident(eL, pts, cam)
            q = M.project_last_frame(pts, cam)
            n_h, a_h = M.SearchByProjectionLastFrame(q, kl, ur_h, dl, None, BOUNDS)
            assert n_r == n_h and np.array_equal(a_r, a_h), (th, rep)
        oq = oracle.sbp_project_last_frame(pts, cam)
        n_o, a_o = oracle.search_by_projection(0, oq, okl, our, odl, None, BOUNDS, nn_ratio=0.9)
        assert n_r == n_o and np.array_equal(a_r, a_o)
        # (2) the local-map overload on caller-built queries, taken flags from the first search
        taken = (a_r >= 0).astype(np.uint8)
        q2 = q.copy()
        q2["radius"] *= 1.5
        M2 = ORBmatcher(0.8, True)
        n2_r, a2_r = M2.search_resident(1, eL, q2, taken, BOUNDS)
        n2_h, a2_h = M2.SearchByProjectionLocalMap(q2, kl, ur_h, dl, taken, BOUNDS)
        assert n2_r == n2_h and np.array_equal(a2_r, a2_h)
        # (3) caller-supplied uright (a frame whose stereo stage ran through the host-pointer entry)
        n3, a3 = M.search_last_frame_resident(eL, pts, cam, uright=ur_h)
        assert n3 == n_h and np.array_equal(a3, a_h)
        n4, a4 = M.search_last_frame_resident(eL, pts, cam)  # (and the resident grid is rebuilt after it)
        assert n4 == n_h and np.array_equal(a4, a_h)


@pytest.mark.gpu
def test_gpu_resident_entries_follow_the_handle_and_refuse_stale_frames(oracle):
    from vieo_slam_amd._lib import VieoError
    from vieo_slam_amd.matching import ORBmatcher, compute_stereo_matches_resident
    eL, eR, kl, dl, kr, dr = _hip_frame(1001)
    sc = np.asarray(eL.GetScaleFactors(), np.float32)
    M = ORBmatcher(0.9, True)
    pts, cam = _points_and_cam(kl, dl, np.full(len(kl), 3.0, np.float32), sc)
    with pytest.raises(VieoError):  # no resident uright yet
        M.search_last_frame_resident(eL, pts, cam)
    ur1, dp1 = compute_stereo_matches_resident(eL, eR, BASELINE, BF)
    n1, a1 = M.search_last_frame_resident(eL, pts, cam)
    # the next frame through the same handles: the old keys are gone, uright and grid belong to the old epoch
    left2, right2, _ = synth.synth_stereo_pair(1002)
    _, kl2, dl2 = eL(left2)
    assert not eL.holds(kl) and eL.holds(kl2)
    with pytest.raises(VieoError):
        M.search_last_frame_resident(eL, pts, cam)
    _, kr2, dr2 = eR(right2)
    ur2, dp2 = compute_stereo_matches_resident(eL, eR, BASELINE, BF)
    n2, a2 = M.search_last_frame_resident(eL, pts, cam)
    q = M.project_last_frame(pts, cam)
    n_h, a_h = M.SearchByProjectionLastFrame(q, kl2, ur2, dl2, None, BOUNDS)
    assert n2 == n_h and np.array_equal(a2, a_h) and len(a2) == len(kl2)
    # empty inputs
    n0, a0 = M.search_last_frame_resident(eL, pts[:0], cam)
    assert n0 == 0 and np.all(a0 == -1)


@pytest.mark.gpu
def test_gpu_extract_strided_and_repeated_still_bit_exact(oracle):
    """vieo_orb_extract now stages the image through the handle's pinned plane and brings counts | keys | descriptors
    back as one block: strided views, width not a multiple of 16, repeated calls."""
    from vieo_slam_amd.orb_extractor import ORBextractor
    e = ORBextractor(1000, 1.2, 8, 20, 7)
    o = oracle.extractor(1000)
    for seed, (w, h) in ((7, (752, 480)), (8, (500, 333)), (9, (752, 480))):
        big = np.zeros((h, w + 37), np.uint8)
        big[:, :w] = synth.synth_image(seed, w, h)
        view = big[:, :w]
        mono, kps, desc = e(view)
        omono, okps, odesc = o(np.ascontiguousarray(view))
        assert mono == omono and np.array_equal(kps.view(np.uint8), okps.view(np.uint8)) and np.array_equal(desc, odesc)
        assert e.holds(kps)
