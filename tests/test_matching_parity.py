"""GPU parity: HIP Hamming matching (knn-2, rectified stereo) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

from vieo_slam_amd import synth
from vieo_slam_amd._lib import DeviceBuffer, check, lib

pytestmark = pytest.mark.gpu
BF, BASELINE = synth.EUROC_BF, synth.EUROC_BF / synth.EUROC_FX


def _hip(nfeat=1200):
    from vieo_slam_amd.orb_extractor import ORBextractor
    return ORBextractor(nfeat, 1.2, 8, 20, 7)


@pytest.mark.parametrize("nq,nt", [(1200, 1200), (1500, 1000), (1, 1), (7, 1), (65, 130), (3, 0), (300, 33), (257, 31), (40, 2100)])
def test_knn2_parity(oracle, nq, nt):
    from vieo_slam_amd.matching import knn_match2
    q = synth.synth_descriptors(nq, seed=7, n_dup=nq // 2)
    t = synth.synth_descriptors(max(nt, 1), seed=8, n_dup=nt // 2)[:nt]
    if nt > 50:
        t[40] = t[10]
        q[0] = t[10]
    oi, od = oracle.knn2(q, t) if nt > 0 else (np.full((nq, 2), -1, np.int32),
                                               np.full((nq, 2), np.iinfo(np.int32).max, np.int32))
    hi, hd = knn_match2(q, t)
    assert np.array_equal(oi, hi) and np.array_equal(od, hd)


def test_knn2_extreme_rows(oracle):
    """all-zero / all-one rows (|a| = 0, 256; a.b = 0, 256) and exact duplicates among the train rows: the matrix-core form
    computes |a| + |b| - 2 a.b, its packed keys must order exactly like the distances + indices"""
    from vieo_slam_amd.matching import knn_match2
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (70, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (45, 32), dtype=np.uint8)
    q[0], q[1], q[2] = 0, 255, t[7]
    t[0], t[1], t[20], t[21], t[44] = 255, 0, t[7], 0, 255
    oi, od = oracle.knn2(q, t)
    hi, hd = knn_match2(q, t)
    assert np.array_equal(oi, hi) and np.array_equal(od, hd)
    assert od[0, 0] == 0 and oi[0, 0] == 1 and od[1, 0] == 0 and oi[1, 0] == 0 and oi[2].tolist() == [7, 20]


def test_knn2_real_descriptors_with_ties(oracle):
    from vieo_slam_amd.matching import knn_match2
    left, right, _ = synth.synth_stereo_pair(1001)
    _, _, dl = oracle.extractor(1200)(left)
    _, _, dr = oracle.extractor(1200)(right)
    oi, od = oracle.knn2(dl, dr)
    hi, hd = knn_match2(dl, dr)
    assert np.array_equal(oi, hi) and np.array_equal(od, hd)


@pytest.mark.parametrize("seed", [1000, 1001, 1002])
def test_stereo_rectified_host_api(oracle, seed):
    from vieo_slam_amd.matching import compute_stereo_matches
    left, right, _ = synth.synth_stereo_pair(seed)
    oL, oR = oracle.extractor(1200), oracle.extractor(1200)
    _, kl, dl = oL(left)
    _, kr, dr = oR(right)
    our, odp = oracle.stereo_match(oL, oR, kl, dl, kr, dr, BASELINE, BF)
    hL, hR = _hip(), _hip()
    _, hkl, hdl = hL(left)
    _, hkr, hdr = hR(right)
    assert np.array_equal(hkl, kl) and np.array_equal(hdr, dr)
    hur, hdp = compute_stereo_matches(hL, hR, hkl, hdl, hkr, hdr, BASELINE, BF)
    assert (our >= 0).sum() > 300
    assert np.array_equal(our.view(np.uint32), hur.view(np.uint32))
    assert np.array_equal(odp.view(np.uint32), hdp.view(np.uint32))


def test_stereo_rectified_edge_cases(oracle):
    from vieo_slam_amd.matching import compute_stereo_matches
    left, right, _ = synth.synth_stereo_pair(1003)
    oL, oR = oracle.extractor(1200), oracle.extractor(1200)
    _, kl, dl = oL(left)
    _, kr, dr = oR(right)
    hL, hR = _hip(), _hip()
    hL(left), hR(right)
    # no right keys at all; a handful of right keys; unrelated right image (few/no matches)
    for sel in (slice(0, 0), slice(0, 5)):
        our, odp = oracle.stereo_match(oL, oR, kl, dl, kr[sel], dr[sel], BASELINE, BF)
        hur, hdp = compute_stereo_matches(hL, hR, kl, dl, kr[sel], dr[sel], BASELINE, BF)
        assert np.array_equal(our.view(np.uint32), hur.view(np.uint32))
        assert np.array_equal(odp.view(np.uint32), hdp.view(np.uint32))
    other = synth.synth_image(2000)
    _, k2, d2 = oR(other)
    hR(other)
    our, odp = oracle.stereo_match(oL, oR, kl, dl, k2, d2, BASELINE, BF)
    hur, hdp = compute_stereo_matches(hL, hR, kl, dl, k2, d2, BASELINE, BF)
    assert np.array_equal(our.view(np.uint32), hur.view(np.uint32))
    assert np.array_equal(odp.view(np.uint32), hdp.view(np.uint32))


def test_stereo_rectified_batch_device(oracle):
    from vieo_slam_amd.orb_extractor import KEYPOINT_DTYPE
    F = 4
    imgs = np.zeros((F, 2, 480, 752), np.uint8)
    for f in range(F):
        imgs[f, 0], imgs[f, 1], _ = synth.synth_stereo_pair(1040 + f)
    h = _hip()
    cap = h.max_keypoints()
    d_img = DeviceBuffer(imgs.nbytes)
    d_img.upload(imgs)
    n_img = 2 * F
    d_kp, d_desc, d_cnt = DeviceBuffer(n_img * cap * 28), DeviceBuffer(n_img * cap * 32), DeviceBuffer(n_img * 8)
    d_ur, d_dp = DeviceBuffer(F * cap * 4), DeviceBuffer(F * cap * 4)
    h.extract_batch_device(d_img.ptr, n_img, 752, 480, 752, 752 * 480, d_kp.ptr, d_desc.ptr, cap, d_cnt.ptr)
    check(lib().vieo_stereo_match_rectified_batch_device(h._h, F, d_kp.ptr, d_desc.ptr, d_cnt.ptr, cap,
                                                         BASELINE, BF, d_ur.ptr, d_dp.ptr))
    h.sync()
    cnt = d_cnt.download(np.int32, (n_img, 2))
    ur = d_ur.download(np.float32, (F, cap))
    dp = d_dp.download(np.float32, (F, cap))
    oL, oR = oracle.extractor(1200), oracle.extractor(1200)
    for f in range(F):
        _, kl, dl = oL(imgs[f, 0])
        _, kr, dr = oR(imgs[f, 1])
        our, odp = oracle.stereo_match(oL, oR, kl, dl, kr, dr, BASELINE, BF)
        n = cnt[2 * f, 0]
        assert n == len(kl)
        assert np.array_equal(our.view(np.uint32), ur[f, :n].view(np.uint32)), f
        assert np.array_equal(odp.view(np.uint32), dp[f, :n].view(np.uint32)), f
