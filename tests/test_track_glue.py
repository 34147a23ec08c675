"""The device-side glue of the chained frame (track_glue.hip) on batches of several frames against numpy: the held-entry
marks of Tracking::SearchLocalPoints and the observation gathering with the `close` bit taken from the tracked depth."""
import numpy as np
import pytest

from vieo_slam_amd.ba_types import POSE_OBS_DTYPE, VIO_FRAME_DTYPE
from vieo_slam_amd.orb_extractor import KEYPOINT_DTYPE


@pytest.mark.gpu
def test_mark_held_and_build_obs_depth_batches():
    from vieo_slam_amd._lib import DeviceBuffer, check, lib
    rng = np.random.default_rng(5)
    B, key_cap, p_cap = 5, 300, 700
    counts = np.zeros((2 * B, 2), np.int32)        # the extractor's layout: image 2 f is frame f's left image
    counts[0::2, 0] = rng.integers(0, key_cap + 1, B)
    counts[2, 0] = 0                                # an empty frame
    mp_ref = rng.integers(-1, p_cap, (B, key_cap)).astype(np.int32)
    mp_ref[rng.random((B, key_cap)) < 0.5] = -1
    keys = np.zeros((2 * B, key_cap), KEYPOINT_DTYPE)
    keys["x"], keys["y"] = rng.uniform(0, 752, keys.shape), rng.uniform(0, 480, keys.shape)
    keys["octave"] = rng.integers(0, 8, keys.shape)
    ur = np.where(rng.random((B, key_cap)) < 0.6, rng.uniform(0, 752, (B, key_cap)), -1).astype(np.float32)
    xyz = rng.normal(0, 3, (B, p_cap, 3)).astype(np.float32)
    depth = rng.uniform(0.5, 40, (B, p_cap)).astype(np.float32)
    depth[rng.random((B, p_cap)) < 0.1] = np.inf
    isig = (np.float32(1) / (np.float32(1.2) ** np.arange(8, dtype=np.float32)) ** 2).astype(np.float32)
    frames = np.zeros(B, VIO_FRAME_DTYPE)
    D = DeviceBuffer
    bufs = {}
    for name, a in (("counts", counts), ("mp_ref", mp_ref), ("keys", keys), ("ur", ur), ("xyz", xyz), ("depth", depth),
                    ("isig", isig), ("frames", frames)):
        bufs[name] = D(a.nbytes)
        bufs[name].upload(a)
    d_held, d_obs, d_okey = D(B * p_cap), D(B * key_cap * 32), D(B * key_cap * 4)
    L = lib()
    check(L.vieo_track_mark_held_batch_device(bufs["mp_ref"].ptr, bufs["counts"].ptr, key_cap, B, 0, 2, d_held.ptr, p_cap, None))
    close = 12.5
    check(L.vieo_track_build_obs_depth_batch_device(bufs["mp_ref"].ptr, bufs["xyz"].ptr, bufs["depth"].ptr, close, p_cap,
                                                    bufs["keys"].ptr, bufs["ur"].ptr, bufs["counts"].ptr, key_cap, B, 0, 2,
                                                    bufs["isig"].ptr, d_obs.ptr, d_okey.ptr, bufs["frames"].ptr, 1, None))
    check(L.vieo_device_synchronize())
    held = d_held.download(np.uint8, (B, p_cap))
    obs = d_obs.download(POSE_OBS_DTYPE, (B, key_cap))
    okey = d_okey.download(np.int32, (B, key_cap))
    fr = bufs["frames"].download(VIO_FRAME_DTYPE, (B,))
    for f in range(B):
        N = counts[2 * f, 0]
        m = mp_ref[f, :N]
        exp_held = np.zeros(p_cap, np.uint8)
        exp_held[m[m >= 0]] = 1
        assert np.array_equal(held[f], exp_held), f
        idx = np.nonzero(m >= 0)[0]
        assert fr[f]["base"]["n_obs"] == len(idx) and fr[f]["base"]["obs_begin"] == f * key_cap
        o = obs[f, :len(idx)]
        assert np.array_equal(okey[f, :len(idx)], idx)
        assert np.array_equal(o["Xw"], xyz[f, m[idx]])
        assert np.array_equal(o["u"], keys["x"][2 * f, idx]) and np.array_equal(o["v"], keys["y"][2 * f, idx])
        assert np.array_equal(o["ur"], ur[f, idx])
        assert np.array_equal(o["inv_sigma2"], isig[keys["octave"][2 * f, idx]])
        assert np.array_equal(o["flags"], (depth[f, m[idx]] < np.float32(close)).astype(np.int32))
