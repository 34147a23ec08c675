"""Known-answer tests that pin the CPU oracle (the reference holds no golden vectors for this
path, SURVEY.md 8c: these are independent analytic expectations + constants of the reference)."""
import math
import os

import numpy as np
import pytest

from vieo_slam_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_cv_round_half_even(oracle):
    assert [oracle.cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == \
        [0, 2, 2, 0, -2, 2, 3]


def test_fast_atan2_axes_and_accuracy(oracle):
    assert oracle.fast_atan2(0.0, 1.0) == 0.0
    assert abs(oracle.fast_atan2(1.0, 0.0) - 90.0) < 1e-4
    assert abs(oracle.fast_atan2(0.0, -1.0) - 180.0) < 1e-4
    assert abs(oracle.fast_atan2(-1.0, 0.0) - 270.0) < 1e-4
    rng = np.random.default_rng(3)
    for _ in range(2000):
        y, x = rng.integers(-200000, 200000, 2)
        ref = math.degrees(math.atan2(y, x)) % 360.0
        got = oracle.fast_atan2(float(y), float(x))
        d = abs(got - ref)
        assert min(d, 360 - d) < 0.02  # published accuracy of cv::fastAtan2 is ~0.3 deg


def test_resize_identity_and_constant(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert np.array_equal(oracle.resize(img, 53, 37), img)
    const = np.full((40, 60), 173, np.uint8)
    assert np.all(oracle.resize(const, 50, 33) == 173)


def test_resize_ramp_monotone_and_bounds(oracle):
    ramp = np.tile(np.arange(0, 240, 2, dtype=np.uint8), (30, 1))  # 120 wide
    out = oracle.resize(ramp, 100, 25)
    assert out.min() >= ramp.min() and out.max() <= ramp.max()
    assert np.all(np.diff(out[5].astype(int)) >= 0)
    # interior samples of a linear ramp are reproduced to fixed-point accuracy
    xs = (np.arange(100) + 0.5) * 1.2 - 0.5
    expect = 2.0 * xs
    assert np.all(np.abs(out[5, 2:-2].astype(float) - expect[2:-2]) <= 1.0)


def test_blur_constant_and_impulse(oracle):
    const = np.full((31, 45), 201, np.uint8)
    assert np.all(oracle.blur(const) == 201)  # Q8.8 kernel sums to exactly 256
    k = np.array([18, 34, 48, 56, 48, 34, 18])
    img = np.zeros((21, 21), np.uint8)
    img[10, 10] = 255
    out = oracle.blur(img)
    for dy in range(-3, 4):
        for dx in range(-3, 4):
            expect = (int(k[dy + 3]) * (255 * int(k[dx + 3])) + 32768) >> 16
            assert out[10 + dy, 10 + dx] == expect
    assert out[10, 14] == 0 and out[14, 10] == 0
    # REFLECT_101 at the border: a bright corner pixel is mirrored, not replicated
    img = np.zeros((21, 21), np.uint8)
    img[0, 0] = 255
    out = oracle.blur(img)
    assert out[0, 0] == (56 * (255 * 56) + 32768) >> 16
    assert out[0, 1] == (56 * (255 * 48) + 32768) >> 16


def _ring_image(center, arc_val, other_val, arc_start, arc_len, size=15):
    off = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3),
           (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    img = np.full((size, size), other_val, np.uint8)
    c = size // 2
    img[c, c] = center
    for k in range(16):
        dx, dy = off[k]
        img[c + dy, c + dx] = other_val
    for k in range(arc_start, arc_start + arc_len):
        dx, dy = off[k % 16]
        img[c + dy, c + dx] = arc_val
    return img, c


def test_fast_arc_of_nine(oracle):
    # 9 contiguous brighter ring pixels -> corner, score = min(x - v) - 1
    for start in range(16):
        img, c = _ring_image(100, 200, 100, start, 9)
        kp = oracle.fast(img, 20)
        assert [tuple(k) for k in kp if (k[0], k[1]) == (c, c)] == [(c, c, 99)]
    # only 8 contiguous -> not a corner at the centre
    img, c = _ring_image(100, 200, 100, 3, 8)
    assert not any((k[0], k[1]) == (c, c) for k in oracle.fast(img, 20))
    # darker arc, threshold boundary: v - x = 21 > 20 is a corner, 20 is not
    img, c = _ring_image(100, 79, 100, 5, 9)
    assert any((k[0], k[1], k[2]) == (c, c, 20) for k in oracle.fast(img, 20))
    img, c = _ring_image(100, 80, 100, 5, 9)
    assert not any((k[0], k[1]) == (c, c) for k in oracle.fast(img, 20))
    assert any((k[0], k[1], k[2]) == (c, c, 19) for k in oracle.fast(img, 7))


def test_fast_border_and_nms(oracle):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    kp = oracle.fast(img, 20)
    assert len(kp) > 0
    assert kp[:, 0].min() >= 3 and kp[:, 0].max() <= 50 - 4
    assert kp[:, 1].min() >= 3 and kp[:, 1].max() <= 40 - 4
    # row-major output order
    lin = kp[:, 1] * 50 + kp[:, 0]
    assert np.all(np.diff(lin) > 0)
    # strict 8-neighbour maxima: no two keypoints adjacent
    s = set((int(a), int(b)) for a, b, _ in kp)
    for (x, y) in s:
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                if dx or dy:
                    assert (x + dx, y + dy) not in s


def test_extractor_tables(oracle):
    assert oracle.extractor(1200).features_per_level() == [261, 217, 181, 151, 126, 105, 87, 72]
    assert oracle.extractor(1000).features_per_level() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert oracle.extractor(1500).features_per_level() == [326, 271, 226, 189, 157, 131, 109, 91]
    e = oracle.extractor(1200)
    assert [e.L.vo_orb_umax(e.h, v) for v in range(16)] == \
        [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    sf = e.scale_factors()
    assert sf[0] == 1.0 and abs(sf[7] - 1.2 ** 7) < 1e-5


def test_extractor_pyramid_sizes_and_invariants(oracle):
    e = oracle.extractor(1200)
    img = synth.synth_image(1000)
    mono, kps, desc = e(img)
    assert mono == 0
    sizes = [e.level_size(l) for l in range(8)]
    assert sizes == [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193),
                     (252, 161), (210, 134)]  # SURVEY.md 8 (float32 arithmetic of the ctor)
    assert 1200 <= len(kps) <= 1200 + 3 * 8
    assert np.all(np.diff(kps["octave"]) >= 0)  # levels concatenated in order
    for l in range(8):
        k = kps[kps["octave"] == l]
        w, h = sizes[l]
        s = np.float32(e.scale_factors()[l])
        assert np.all(k["x"] >= np.float32(19) * s) and np.all(k["x"] <= np.float32(w - 20) * s)
        assert np.all(k["y"] >= np.float32(19) * s) and np.all(k["y"] <= np.float32(h - 20) * s)
        assert np.all(k["size"] == np.float32(int(31 * s)))
    assert np.all((kps["angle"] >= 0) & (kps["angle"] <= 360))
    assert np.all(kps["response"] >= 7) and np.all(kps["class_id"] == -1)
    assert desc.shape == (len(kps), 32) and desc.any()
    # bordered plane = REFLECT_101 of the ROI
    b = e.plane(1, 2)
    p = e.plane(1, 0)
    assert np.array_equal(b[19:-19, 19:-19], p)
    assert np.array_equal(b[19:-19, 0], p[:, 19]) and np.array_equal(b[0, 19:-19], p[19, :])


def test_extractor_lapping_area(oracle):
    e = oracle.extractor(600)
    img = synth.synth_image(1001, 512, 512)
    mono0, k0, d0 = e(img)
    # whole image is lapping area (TUM-VI, TUM_VI_512_VIO.yaml:88-91): everything written from
    # the back -> reversed order, monoIndex 0
    mono, k1, d1 = e(img, lapping=(0, 511))
    assert mono == 0 and len(k1) == len(k0)
    assert np.array_equal(k1, k0[::-1]) and np.array_equal(d1, d0[::-1])
    # split at x in [200, 300]: mono keys first (original order), stereo keys reversed at the end
    mono, k2, d2 = e(img, lapping=(200, 300))
    st = (k0["x"] >= 200) & (k0["x"] <= 300)
    assert mono == int((~st).sum())
    assert np.array_equal(k2[:mono], k0[~st]) and np.array_equal(k2[mono:], k0[st][::-1])
    assert np.array_equal(d2[:mono], d0[~st]) and np.array_equal(d2[mono:], d0[st][::-1])


def test_extractor_empty_image(oracle):
    e = oracle.extractor(500)
    mono, k, d = e(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(k) == 0


def test_oracle_matches_committed_golden(oracle):
    """Pins the oracle itself: seeded synthetic frames -> committed keypoints/descriptors
    (tests/golden/make_golden.py wrote them)."""
    g = np.load(os.path.join(GOLD, "orb_golden.npz"))
    for tag, (seed, w, h, nfeat, lap) in {"euroc": (1000, 752, 480, 1200, None),
                                          "tumvi": (1001, 512, 512, 1500, (0, 511))}.items():
        e = oracle.extractor(nfeat)
        mono, kps, desc = e(synth.synth_image(seed, w, h), lapping=lap)
        assert mono == int(g[tag + "_mono"])
        assert np.array_equal(kps.view(np.uint8), g[tag + "_kps"].view(np.uint8))
        assert np.array_equal(desc, g[tag + "_desc"])


# ---------------------------------------------------------------- round 6: the unpinned oracle tightened from the inside
# (VERDICT r5 item 10).  None of this pins the oracle against a real OpenCV -- there is none in the image -- but it takes
# "a typo in a constant" off the list: the constants are DERIVED here, not asserted as literals.

def test_gaussian_taps_follow_from_sigma_by_error_diffusion(oracle):
    """cv::GaussianBlur(8U, 7 x 7, sigma = 2) uses a Q8.8 kernel whose coefficients sum to exactly 256: the real-valued
    kernel exp(-x^2 / (2 sigma^2)) / sum, scaled by 256 and rounded WITH the rounding error carried to the next tap
    (plain rounding gives 18 34 49 55 49 34 18 = 257).  The oracle's impulse response must be that kernel's outer product
    with the vertical pass's round-to-nearest."""
    sigma, n = 2.0, 7
    x = np.arange(n) - (n - 1) / 2
    k = np.exp(-x * x / (2 * sigma * sigma))
    k /= k.sum()
    taps, err = [], 0.0
    for i in range(n):
        v = k[i] * 256 + err
        t = int(math.floor(v + 0.5))
        err = v - t
        taps.append(t)
    assert sum(taps) == 256 and taps == taps[::-1]
    plain = [int(math.floor(v * 256 + 0.5)) for v in k]
    assert sum(plain) != 256  # (the diffusion is what makes a constant image come out unchanged)
    img = np.zeros((23, 23), np.uint8)
    img[11, 11] = 200
    out = oracle.blur(img)
    for dy in range(-3, 4):
        for dx in range(-3, 4):
            assert out[11 + dy, 11 + dx] == (taps[dy + 3] * (200 * taps[dx + 3]) + 32768) >> 16, (dy, dx)


def _resize_linear_8u(src, dw, dh):
    """cv::resize(INTER_LINEAR, CV_8UC1) written out independently of oracle/ocv_prims.hpp from SURVEY.md Appendix A: float
    source coordinates, 11-bit coefficients by round-half-even, the horizontal pass in int, the vertical pass
    ((b0 (r0 >> 4)) >> 16) + ((b1 (r1 >> 4)) >> 16) + 2 >> 2."""
    sh, sw = src.shape
    def coeffs(sn, dn):
        scale = sn / dn
        idx, w0, w1 = [], [], []
        for d in range(dn):
            f = np.float32((d + 0.5) * scale - 0.5)
            s = int(math.floor(f))
            f = np.float32(f - s)
            if s < 0:
                s, f = 0, np.float32(0)
            if s >= sn - 1:
                s, f = sn - 1, np.float32(0)
            idx.append(s)
            w1.append(int(np.rint(np.float32(f * np.float32(2048)))))
            w0.append(int(np.rint(np.float32((np.float32(1) - f) * np.float32(2048)))))
        return idx, w0, w1
    xi, xa, xb = coeffs(sw, dw)
    yi, ya, yb = coeffs(sh, dh)
    s = src.astype(np.int64)
    rows = np.zeros((sh, dw), np.int64)
    for d in range(dw):
        rows[:, d] = s[:, xi[d]] * xa[d] + s[:, min(xi[d] + 1, sw - 1)] * xb[d]
    out = np.zeros((dh, dw), np.uint8)
    for d in range(dh):
        r0, r1 = rows[yi[d]], rows[min(yi[d] + 1, sh - 1)]
        out[d] = ((((ya[d] * (r0 >> 4)) >> 16) + ((yb[d] * (r1 >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
    return out


def test_resize_ramps_at_the_seven_euroc_level_ratios(oracle):
    """Level l + 1 is resized from level l (ORBextractor.cc:1070): the seven (source, destination) size pairs of a 752 x 480
    image at scale 1.2.  On a linear ramp bilinear interpolation is exact up to the fixed-point rounding, and the whole plane
    must equal the independent restatement above byte for byte (ramps in x, in y, and a random plane)."""
    s, sizes = np.float32(1.0), []
    for _ in range(8):
        inv = np.float32(1.0) / s
        sizes.append((int(np.rint(np.float32(752) * inv)), int(np.rint(np.float32(480) * inv))))
        s = np.float32(s * np.float64(np.float32(1.2)))
    assert sizes[0] == (752, 480) and sizes[-1] == (210, 134)
    rng = np.random.default_rng(11)
    for (sw, sh), (dw, dh) in zip(sizes[:-1], sizes[1:]):
        rx = np.tile((np.arange(sw) * 255.0 / (sw - 1)).astype(np.uint8), (sh, 1))
        ry = np.tile((np.arange(sh) * 255.0 / (sh - 1)).astype(np.uint8)[:, None], (1, sw))
        rnd = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        for name, img in (("x ramp", rx), ("y ramp", ry), ("random", rnd)):
            got = oracle.resize(img, dw, dh)
            assert np.array_equal(got, _resize_linear_8u(img, dw, dh)), (name, sw, sh, dw, dh)
        # the ramp itself: within one grey level of the real-valued interpolation of the (quantised) source ramp
        xs = np.clip((np.arange(dw) + 0.5) * (sw / dw) - 0.5, 0, sw - 1)
        expect = np.interp(xs, np.arange(sw), rx[0].astype(float))
        assert np.abs(oracle.resize(rx, dw, dh)[dh // 2].astype(float) - expect).max() <= 1.0


def test_fast_atan2_within_its_documented_accuracy_over_the_moment_range(oracle):
    """IC_Angle feeds fastAtan2((float)m_01, (float)m_10) with integer moments of the radius-15 disc (|m| <= 255 x sum |u| over
    the disc ~ 1.2e6).  cv::fastAtan2's documented accuracy is 0.3 degrees; the restated polynomial has to stay inside it on a
    dense grid of small moments (where the quantisation is coarsest), on the same grid scaled up to the largest moments, and
    along the octant boundaries where the polynomial's branches meet."""
    def err(y, x):
        ref = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(oracle.fast_atan2(float(y), float(x)) - ref)
        return min(d, 360 - d)
    worst = 0.0
    for y in range(-120, 121):
        for x in range(-120, 121):
            if x == 0 and y == 0:
                continue
            worst = max(worst, err(y, x))
    for scale in (97, 9973):  # up to ~1.2e6
        for y in range(-120, 121, 3):
            for x in range(-120, 121, 3):
                if x or y:
                    worst = max(worst, err(y * scale, x * scale + (y % 7)))
    for m in (1, 2, 1000, 1199999):  # |x| = |y| and the axes: the branch boundaries
        for sy in (-1, 1):
            for sx in (-1, 1):
                worst = max(worst, err(sy * m, sx * m), err(sy * m, sx * (m + 1)), err(sy * (m + 1), sx * m))
        worst = max(worst, err(0, m), err(0, -m), err(m, 0), err(-m, 0))
    assert worst < 0.3, worst
    assert worst < 0.02, worst  # (what the polynomial actually achieves: an order of magnitude inside the documented bound)
    assert oracle.fast_atan2(0.0, 0.0) == 0.0  # OpenCV returns 0 for the null vector
