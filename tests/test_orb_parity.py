"""GPU parity tests: HIP extractor (through the C-ABI) vs the CPU oracle, bit-exact.

Stage-wise (pyramid, blur, FAST candidates, quadtree selection) and end to end (keypoints,
descriptors, monoIndex), on seeded synthetic frames at the BASELINE.json sizes, odd sizes and the
edge cases the reference handles (empty image, lapping area, min-threshold fallback cells).
"""
import os

import numpy as np
import pytest

from vieo_slam_amd import synth
from vieo_slam_amd._lib import DeviceBuffer

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _hip(nfeat=1200, scale=1.2, nlevels=8, ini=20, mn=7):
    from vieo_slam_amd.orb_extractor import ORBextractor
    return ORBextractor(nfeat, scale, nlevels, ini, mn)


def _assert_same(ref, got):
    (m0, k0, d0), (m1, k1, d1) = ref, got
    assert m0 == m1
    assert len(k0) == len(k1)
    for f in k0.dtype.names:
        assert np.array_equal(k0[f].view(np.uint32), k1[f].view(np.uint32)), f
    assert np.array_equal(d0, d1)


def test_tables_match(oracle):
    for nf in (1000, 1200, 1500):
        o, h = oracle.extractor(nf), _hip(nf)
        assert list(h.features_per_level()) == o.features_per_level()
        assert np.array_equal(h.GetScaleFactors(), np.array(o.scale_factors(), np.float32))
        assert h.GetLevels() == 8 and abs(h.GetScaleFactor() - 1.2) < 1e-6
        s = h.GetScaleFactors()
        assert np.array_equal(h.GetInverseScaleFactors(), np.float32(1) / s)
        assert np.array_equal(h.GetScaleSigmaSquares(), s * s)
        assert np.array_equal(h.GetInverseScaleSigmaSquares(), np.float32(1) / (s * s))


def test_stagewise_euroc(oracle):
    img = synth.synth_image(1000)
    o, h = oracle.extractor(1200), _hip(1200)
    ref = o(img)
    got = h(img)
    for l in range(8):
        assert h.level_size(l) == o.level_size(l)
        assert np.array_equal(h.image_pyramid(l), o.plane(l, 0)), "pyramid level %d" % l
        assert np.array_equal(h.image_pyramid(l, with_border=True), o.plane(l, 2))
        assert np.array_equal(h.tap_blurred(l), o.plane(l, 1)), "blur level %d" % l
        assert np.array_equal(h.tap_candidates(l), o.candidates(l)), "FAST level %d" % l
        ok, hk = o.level_keys(l), h.tap_level_keys(l)
        assert len(ok) == len(hk), "quadtree count level %d" % l
        for f in ("x", "y", "response", "size", "octave"):
            assert np.array_equal(ok[f], hk[f]), (l, f)
    _assert_same(ref, got)


@pytest.mark.parametrize("seed,w,h,nfeat", [(1001, 752, 480, 1200), (1002, 752, 480, 1000),
                                            (1003, 512, 512, 1500), (1004, 641, 479, 700),
                                            (1005, 320, 240, 500), (1006, 1280, 720, 2000)])
def test_end_to_end_sizes(oracle, seed, w, h, nfeat):
    img = synth.synth_image(seed, w, h)
    _assert_same(oracle.extractor(nfeat)(img), _hip(nfeat)(img))


def test_lapping_area(oracle):
    img = synth.synth_image(1001, 512, 512)
    o, h = oracle.extractor(1500), _hip(1500)
    for lap in ((0, 511), (200, 300), (600, 700)):
        _assert_same(o(img, lapping=lap), h(img, pvLappingArea=lap))


def test_min_threshold_fallback_and_flat_regions(oracle):
    # low-contrast frame: most cells find nothing at iniThFAST=20 and fall back to minThFAST=7
    img = (synth.synth_image_f32(1010) - 128.0) * 0.12 + 128.0
    img = synth.quantise(img.astype(np.float32), 1010)
    _assert_same(oracle.extractor(1200)(img), _hip(1200)(img))
    # half of the frame perfectly flat: empty cells, unbalanced quadtree
    img2 = synth.synth_image(1011)
    img2[:, 376:] = 90
    _assert_same(oracle.extractor(1200)(img2), _hip(1200)(img2))
    # completely flat: zero keypoints
    flat = np.full((480, 752), 77, np.uint8)
    m, k, d = _hip(1200)(flat)
    assert (m, len(k), len(d)) == (0, 0, 0)
    assert len(oracle.extractor(1200)(flat)[1]) == 0


def test_noise_image_many_candidates(oracle):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (480, 752), dtype=np.uint8)  # ~10^4 candidates per level
    _assert_same(oracle.extractor(1200)(img), _hip(1200)(img))


def test_empty_image_returns_minus_one():
    m, k, d = _hip(1200)(np.zeros((0, 0), np.uint8))
    assert m == -1 and len(k) == 0


def test_strided_input_and_reuse_across_sizes(oracle):
    h = _hip(1200)
    big = synth.synth_image(1020, 800, 500)
    view = big[10:490, 20:772]  # non-contiguous rows (stride 800)
    _assert_same(oracle.extractor(1200)(np.ascontiguousarray(view)), h(view))
    small = synth.synth_image(1021, 400, 300)
    _assert_same(oracle.extractor(1200)(small), h(small))  # same handle, new geometry
    _assert_same(oracle.extractor(1200)(np.ascontiguousarray(view)), h(view))


def test_batch_device_matches_oracle_and_single(oracle):
    from vieo_slam_amd.orb_extractor import KEYPOINT_DTYPE
    B, w, hgt = 6, 752, 480
    imgs = np.stack([synth.synth_image(1030 + i) for i in range(B)])
    h = _hip(1200)
    cap = h.max_keypoints()
    d_img = DeviceBuffer(imgs.nbytes)
    d_img.upload(imgs)
    d_kp, d_desc, d_cnt = DeviceBuffer(B * cap * 28), DeviceBuffer(B * cap * 32), DeviceBuffer(B * 8)
    h.extract_batch_device(d_img.ptr, B, w, hgt, w, w * hgt, d_kp.ptr, d_desc.ptr, cap, d_cnt.ptr)
    h.sync()
    cnt = d_cnt.download(np.int32, (B, 2))
    kps = d_kp.download(KEYPOINT_DTYPE, (B, cap))
    desc = d_desc.download(np.uint8, (B, cap, 32))
    o = oracle.extractor(1200)
    for i in range(B):
        n = cnt[i, 0]
        _assert_same(o(imgs[i]), (int(cnt[i, 1]), kps[i, :n], desc[i, :n]))
    # idempotence: a second run over the same resident batch gives the same bytes
    h.extract_batch_device(d_img.ptr, B, w, hgt, w, w * hgt, d_kp.ptr, d_desc.ptr, cap, d_cnt.ptr)
    h.sync()
    assert np.array_equal(cnt, d_cnt.download(np.int32, (B, 2)))
    kps2 = d_kp.download(KEYPOINT_DTYPE, (B, cap))
    for i in range(B):
        assert np.array_equal(kps[i, :cnt[i, 0]], kps2[i, :cnt[i, 0]])


def test_matches_committed_golden():
    g = np.load(os.path.join(GOLD, "orb_golden.npz"))
    for tag, (seed, w, h, nfeat, lap) in {"euroc": (1000, 752, 480, 1200, None),
                                          "tumvi": (1001, 512, 512, 1500, (0, 511))}.items():
        mono, kps, desc = _hip(nfeat)(synth.synth_image(seed, w, h), pvLappingArea=lap)
        assert mono == int(g[tag + "_mono"])
        assert np.array_equal(kps.view(np.uint8), g[tag + "_kps"].view(np.uint8))
        assert np.array_equal(desc, g[tag + "_desc"])


def test_rotated_frames_cover_all_quadrants(oracle):
    """The steered-BRIEF path (IC angle -> cos / sin of sincosf_exact.h -> 256 rotated taps) on frames whose key-point
    angles fill every quadrant: a frame, its 90 / 180 / 270 degree rotations and its mirror image.  Descriptors and
    angles stay bit-equal to the oracle, and the angles of the four rotations together populate all 12 bins of 30
    degrees (a synthetic frame alone leaves some nearly empty)."""
    base = synth.synth_image(1042, 640, 480)
    frames = [base, np.ascontiguousarray(np.rot90(base, 1)), np.ascontiguousarray(np.rot90(base, 2)),
              np.ascontiguousarray(np.rot90(base, 3)), np.ascontiguousarray(base[:, ::-1])]
    hist = np.zeros(12, int)
    for img in frames:
        o, h = oracle.extractor(1000), _hip(1000)
        ref, got = o(img), h(img)
        _assert_same(ref, got)
        assert len(ref[1]) > 800
        hist += np.bincount((ref[1]["angle"] // 30).astype(int) % 12, minlength=12)
    assert hist.min() > 100, hist
