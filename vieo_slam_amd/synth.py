"""Seeded synthetic inputs for the hot path: no dataset exists in the container or
on the GPU box, so parity tests and bench.py feed both the oracle and the HIP path with these.

Images are corner-rich 8-bit frames: 3-octave value noise + 400 random rectangles/discs +
uniform noise.  A stereo pair warps the left frame by a piecewise-planar disparity map with
EuRoC's bf (Examples/Stereo/EuRoC/EuRoC_VIO.yaml:75) so the rectified matcher has true answers.
"""
import numpy as np

EUROC_W, EUROC_H = 752, 480
TUMVI_W, TUMVI_H = 512, 512
EUROC_BF = 47.90639384423901
EUROC_FX = 435.2046959714599


def _value_noise(rng, w, h, cell, amp):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.uniform(-1.0, 1.0, (gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32)
    x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return amp * (a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx)


def synth_image_f32(seed, w=EUROC_W, h=EUROC_H, n_shapes=400):
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 128.0, np.float32)
    for cell, amp in ((64, 50.0), (16, 25.0), (4, 10.0)):
        img += _value_noise(rng, w, h, cell, amp)
    for _ in range(n_shapes):
        kind = rng.integers(0, 3)
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        grey = rng.uniform(10, 245)
        rw, rh = rng.uniform(4, 40), rng.uniform(4, 40)
        t = rng.uniform(0, np.pi)
        r = rng.uniform(3, 25)
        ext = int(np.ceil(max(np.hypot(rw, rh), r))) + 1
        xa, xb = max(0, int(cx) - ext), min(w, int(cx) + ext + 1)
        ya, yb = max(0, int(cy) - ext), min(h, int(cy) + ext + 1)
        if xa >= xb or ya >= yb:
            continue
        yy, xx = np.mgrid[ya:yb, xa:xb].astype(np.float32)
        if kind == 0:  # axis-aligned rectangle
            m = (np.abs(xx - cx) < rw) & (np.abs(yy - cy) < rh)
        elif kind == 1:  # rotated rectangle
            c, s = np.cos(t), np.sin(t)
            u = (xx - cx) * c + (yy - cy) * s
            v = -(xx - cx) * s + (yy - cy) * c
            m = (np.abs(u) < rw) & (np.abs(v) < rh)
        else:  # disc
            m = (xx - cx) ** 2 + (yy - cy) ** 2 < r * r
        sub = img[ya:yb, xa:xb]
        sub[m] = grey + 0.15 * (sub[m] - 128.0)
    return img


def quantise(img_f32, seed):
    rng = np.random.default_rng(seed + 7919)
    n = rng.integers(-3, 4, img_f32.shape).astype(np.float32)
    return np.clip(np.rint(img_f32 + n), 0, 255).astype(np.uint8)


def synth_image(seed, w=EUROC_W, h=EUROC_H):
    """One uint8 frame, seed convention of SURVEY 8d: seed = 1000 + frame."""
    return quantise(synth_image_f32(seed, w, h), seed)


def synth_disparity(seed, w=EUROC_W, h=EUROC_H, bf=EUROC_BF):
    """Piecewise-planar depth (1..15 m) -> disparity in pixels, per left pixel."""
    rng = np.random.default_rng(seed + 104729)
    depth = np.full((h, w), 8.0, np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for _ in range(12):
        x0, y0 = rng.uniform(0, w), rng.uniform(0, h)
        rw, rh = rng.uniform(40, 250), rng.uniform(40, 200)
        z0 = rng.uniform(1.0, 15.0)
        gx, gy = rng.uniform(-0.004, 0.004), rng.uniform(-0.004, 0.004)
        m = (np.abs(xx - x0) < rw) & (np.abs(yy - y0) < rh)
        depth[m] = np.clip(z0 + gx * (xx[m] - x0) + gy * (yy[m] - y0), 1.0, 15.0)
    return (bf / depth).astype(np.float32)


def synth_stereo_pair(seed, w=EUROC_W, h=EUROC_H, bf=EUROC_BF):
    """(left, right, disparity).  right(x, y) = left(x + d(x, y), y) with linear interpolation,
    i.e. a left pixel at u_L appears at about u_R = u_L - d in the right frame."""
    left_f = synth_image_f32(seed, w, h)
    disp = synth_disparity(seed, w, h, bf)
    xs = np.arange(w, dtype=np.float32)[None, :] + disp
    x0 = np.clip(np.floor(xs).astype(np.int32), 0, w - 1)
    x1 = np.clip(x0 + 1, 0, w - 1)
    f = np.clip(xs - np.floor(xs), 0, 1)
    rows = np.arange(h)[:, None]
    right_f = left_f[rows, x0] * (1 - f) + left_f[rows, x1] * f
    return quantise(left_f, seed), quantise(right_f, seed + 500000), disp


def synth_descriptors(n, seed=7, n_dup=0, max_flip=60):
    """Random 256-bit rows (n x 32 uint8) with `n_dup` planted near-duplicates of earlier rows at
    Hamming distance 0..max_flip."""
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for k in range(n_dup):
        src = rng.integers(0, n)
        dst = rng.integers(0, n)
        row = np.unpackbits(d[src])
        flips = rng.choice(256, rng.integers(0, max_flip + 1), replace=False)
        row[flips] ^= 1
        d[dst] = np.packbits(row)
    return d
