"""CPU known-answer tests pinning the matching oracle (oracle/matching.cc)."""
import numpy as np

from vieo_slam_amd import synth

BF, BASELINE = synth.EUROC_BF, synth.EUROC_BF / synth.EUROC_FX


def _popcount_dist(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def test_descriptor_distance_is_hamming(oracle):
    d = synth.synth_descriptors(64, seed=7, n_dup=20)
    for i in range(0, 64, 3):
        for j in range(0, 64, 5):
            assert oracle.descriptor_distance(d[i], d[j]) == _popcount_dist(d[i], d[j])
    assert oracle.descriptor_distance(d[3], d[3]) == 0
    assert oracle.descriptor_distance(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256


def test_knn2_matches_bruteforce_with_stable_ties(oracle):
    q = synth.synth_descriptors(150, seed=7, n_dup=60)
    t = synth.synth_descriptors(170, seed=8, n_dup=60)
    t[40] = t[10]  # exact duplicate rows -> distance ties, lower index must come first
    t[99] = t[10]
    q[5] = t[10]
    idx, dist = oracle.knn2(q, t)
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2).astype(np.int32)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]
    assert np.array_equal(idx, order)
    assert np.array_equal(dist, np.take_along_axis(D, order, 1))
    assert list(idx[5]) == [10, 40] and list(dist[5]) == [0, 0]
    # fewer than 2 train rows
    idx1, dist1 = oracle.knn2(q[:3], t[:1])
    assert np.all(idx1[:, 0] == 0) and np.all(idx1[:, 1] == -1)
    assert np.all(dist1[:, 1] == np.iinfo(np.int32).max)


def test_stereo_matcher_recovers_planted_disparity(oracle):
    left, right, disp = synth.synth_stereo_pair(1000)
    eL, eR = oracle.extractor(1200), oracle.extractor(1200)
    _, kl, dl = eL(left)
    _, kr, dr = eR(right)
    ur, dp = oracle.stereo_match(eL, eR, kl, dl, kr, dr, BASELINE, BF)
    ok = ur >= 0
    assert ok.sum() > 300, ok.sum()
    # matched keys: u_L - u_R close to the planted disparity at the key, depth = bf / disparity
    x = np.clip(np.rint(kl["x"][ok]).astype(int), 0, 751)
    y = np.clip(np.rint(kl["y"][ok]).astype(int), 0, 479)
    err = np.abs((kl["x"][ok] - ur[ok]) - disp[y, x])
    assert np.median(err) < 0.6 and np.mean(err < 2.0) > 0.85
    assert np.allclose(dp[ok], np.float32(BF) / (kl["x"][ok] - ur[ok]), rtol=1e-6)
    assert np.all(dp[~ok] == -1) and np.all(ur[~ok] == -1)
    assert np.all(kl["x"][ok] - ur[ok] > 0) and np.all(kl["x"][ok] - ur[ok] < BF / BASELINE)
