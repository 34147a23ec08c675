"""CPU checks of the product's host-side/portable logic against the oracle:
 - the data-parallel quadtree formulation (the exact text of the HIP kernel, host-emulated)
 - the device sin/cos routine's arithmetic (host-compiled) against libm
 - the C-ABI library loads, exports every declared symbol and refuses to compute without a GPU
"""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.emul import build as emul_build
from vieo_slam_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, I = ctypes.c_void_p, ctypes.c_int


@pytest.fixture(scope="module")
def qt():
    L = ctypes.CDLL(emul_build.quadtree())
    L.emul_distribute.argtypes = [P, I, I, I, I, I, I, P, I]

    def run(xyr, minX, maxX, minY, maxY, N):
        xyr = np.ascontiguousarray(xyr, np.int32)
        out = np.zeros((N + 64, 3), np.int32)
        n = L.emul_distribute(xyr.ctypes.data, len(xyr), minX, maxX, minY, maxY, N,
                              out.ctypes.data, N + 64)
        assert n >= 0, n
        return out[:n].copy()
    return run


def _random_keys(rng, W, H, K):
    xy = np.unique(np.stack([rng.integers(0, W, K), rng.integers(0, H, K)], 1), axis=0)
    rng.shuffle(xy)
    r = rng.integers(7, 255, (len(xy), 1))
    return np.ascontiguousarray(np.concatenate([xy, r], 1).astype(np.int32))


def test_quadtree_formulation_random(oracle, qt):
    rng = np.random.default_rng(11)
    for trial in range(150):
        W = int(rng.integers(60, 900))
        H = int(rng.integers(60, min(int(W * 1.9), 700)))
        K = int(rng.integers(0, 4000))
        N = int(rng.integers(1, 400))
        xyr = _random_keys(rng, W, H, K)
        a = oracle.distribute(xyr, 16, 16 + W, 16, 16 + H, N)
        b = qt(xyr, 16, 16 + W, 16, 16 + H, N)
        assert np.array_equal(a, b), (trial, W, H, K, N)


def test_quadtree_formulation_edge_cases(oracle, qt):
    # empty, single key, all keys in one column, duplicate responses (first max wins), N=1
    cases = [np.zeros((0, 3), np.int32),
             np.array([[5, 7, 30]], np.int32),
             np.array([[10, y, 50] for y in range(0, 200, 2)], np.int32),
             np.array([[x, y, 40] for x in range(0, 60, 3) for y in range(0, 60, 3)], np.int32)]
    for xyr in cases:
        for N in (1, 5, 50, 300):
            a = oracle.distribute(xyr, 16, 16 + 300, 16, 16 + 200, N)
            b = qt(xyr, 16, 16 + 300, 16, 16 + 200, N)
            assert np.array_equal(a, b)


def test_quadtree_formulation_real_candidates(oracle, qt):
    e = oracle.extractor(1200)
    feats = e.features_per_level()
    for seed in (1000, 1003):
        e(synth.synth_image(seed))
        for l in range(8):
            w, h = e.level_size(l)
            c = e.candidates(l)
            a = oracle.distribute(c, 16, w - 16, 16, h - 16, feats[l])
            b = qt(c, 16, w - 16, 16, h - 16, feats[l])
            assert np.array_equal(a, b), (seed, l)
    assert e.tie_count() > 0  # the (size, pointer) tie rule is exercised by real data


def test_sincos_arithmetic_matches_libm():
    L = ctypes.CDLL(emul_build.sincos())
    L.emul_sincosf_sweep.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, P]
    L.emul_sincosf_sweep.restype = ctypes.c_long
    hi = np.array([6.3], np.float32).view(np.uint32)[0]
    n = ctypes.c_long()
    # every 16th float in [0, 6.3] (68 M values, < 1 s); the full sweep is the slow test below
    bad = L.emul_sincosf_sweep(0, int(hi), 16, ctypes.byref(n))
    assert n.value > 60_000_000 and bad == 0


@pytest.mark.slow
def test_sincos_arithmetic_exhaustive():
    L = ctypes.CDLL(emul_build.sincos())
    L.emul_sincosf_sweep.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, P]
    L.emul_sincosf_sweep.restype = ctypes.c_long
    hi = np.array([6.3], np.float32).view(np.uint32)[0]
    n = ctypes.c_long()
    assert L.emul_sincosf_sweep(0, int(hi), 1, ctypes.byref(n)) == 0
    assert n.value == int(hi) + 1


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vieo_hot.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(vieo_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libvieo_hot.so does not export %s" % name
    assert declared == set(_lib.declared_symbols())


def test_fails_loudly_without_gpu():
    L = _lib.lib()
    if L.vieo_device_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = L.vieo_orb_create(ctypes.byref(h), 1200, 1.2, 8, 20, 7)
    assert rc == _lib.VIEO_E_NO_DEVICE
    assert b"no CPU fallback" in L.vieo_last_error() or b"not gfx950" in L.vieo_last_error()
    from vieo_slam_amd.orb_extractor import ORBextractor
    with pytest.raises(_lib.VieoError):
        ORBextractor(1200, 1.2, 8, 20, 7)
