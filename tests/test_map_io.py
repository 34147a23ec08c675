"""The binary sparse map of System::SaveMap / LoadMap (vieo_slam_amd/map_io.py): byte-level known answers for the
record layouts the reference writes (src/System.cc:272-421, src/FrameBase.cpp:297-360, src/KeyFrame.cc:163-201) and a
round trip."""
import struct

import numpy as np
import pytest

from vieo_slam_amd import map_io
from vieo_slam_amd.ba_types import NAVSTATE_DTYPE
from vieo_slam_amd.orb_extractor import KEYPOINT_DTYPE


def _kf(rng, kid, n_keys, prev=None, distort=False):
    keys = np.zeros(n_keys, KEYPOINT_DTYPE)
    keys["x"], keys["y"] = rng.uniform(0, 752, n_keys), rng.uniform(0, 480, n_keys)
    keys["octave"], keys["angle"], keys["size"] = rng.integers(0, 8, n_keys), rng.uniform(0, 360, n_keys), 31
    keys["class_id"] = -1
    nav = np.zeros((), NAVSTATE_DTYPE)
    nav["p"], nav["q"], nav["v"] = rng.normal(size=3), (0.5, 0.5, -0.5, 0.5), rng.normal(size=3)
    nav["bg"], nav["ba"], nav["dbg"], nav["dba"] = (rng.normal(size=3) for _ in range(4))
    kf = dict(id=kid, prev_id=prev, timestamp=100.0 + kid, usedistort=distort,
              cameras=[(0, np.array([435.2, 435.2, 367.4, 252.2], np.float32))] if not distort else
              [(2, np.arange(8, dtype=np.float32)), (2, np.arange(8, dtype=np.float32) + 1)],
              keys=keys, th_depth=3.85, descriptors=rng.integers(0, 256, (n_keys, 32), dtype=np.uint8),
              depth=rng.uniform(-1, 5, n_keys).astype(np.float32), uright=rng.uniform(-1, 700, n_keys).astype(np.float32),
              baseline=0.11, n_levels=8, scale_factor=1.2, image_size=(752, 480), nav=nav,
              imu_list=rng.normal(size=(10, 7)), enc_list=rng.normal(size=(4, 3)), state=2,
              matches=np.where(rng.random(n_keys) < 0.5, rng.integers(0, 50, n_keys), -1).astype(np.int64).astype(np.uint64),
              parent_id=prev, loop_ids=[0] if kid == 2 else [])
    if distort:
        kf["mapn2in"] = np.stack([np.arange(n_keys) % 2, np.arange(n_keys) // 2], 1).astype(np.uint64)
        kf["stereo_points"] = rng.normal(size=(5, 3))
        kf["good_matches"] = np.array([1, 0, 1, 1, 0], bool)
        kf["camidx2idxs"] = rng.integers(0, 5, (7, 3)).astype(np.uint64)
    else:
        kf["keys_un"] = keys.copy()
    return kf


def _map(rng, distort=False, sensor=3):
    kfs = [_kf(rng, i, 20 + 3 * i, prev=i - 1 if i else None, distort=distort) for i in range(3)]
    mps = [dict(id=m, ref_kf_id=m % 3, pos=rng.normal(size=3).astype(np.float32),
                observations=[(k, [int(rng.integers(0, 20))] + ([3] if k == 1 and distort else [])) for k in range(3)
                              if rng.random() < 0.8]) for m in range(12)]
    m = dict(sensor_type=sensor, keyframes=kfs, mappoints=mps)
    if sensor & 1:
        m["enc_params"] = dict(vscale=0.001, rc=0.28, Sigma=np.diag([1e-4, 2e-4]), Sigmam=np.diag(np.arange(1, 7) * 1e-6),
                               dt_cov_noise_fixed=1, freq_ref=0.0)
    if sensor & 2:
        m["imu_params"] = dict(multiply_g=1.0, ref_g=9.81, Sigma_g=np.eye(3) * 1e-6, Sigma_a=np.eye(3) * 4e-6,
                               Sigma_bg=np.eye(3) * 1e-9, Sigma_ba=np.eye(3) * 9e-6, dt_cov_noise_fixed=1, freq_ref=0.0)
        m["gravity"] = np.array([0, 0, -9.81], np.float32)
    return m


def _same(a, b):
    if isinstance(a, dict):
        return a.keys() <= b.keys() | {"keys_un"} and all(_same(a[k], b[k]) for k in a if k in b)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        a, b = np.asarray(a), np.asarray(b)
        if a.dtype.names:
            return a.tobytes() == np.asarray(b, a.dtype).tobytes()
        return a.shape == b.shape and np.array_equal(a, b.astype(a.dtype))
    if isinstance(a, float):
        return a == b or abs(a - b) < 1e-6 * max(1, abs(a))  # float32 fields of the file
    return a == b


@pytest.mark.parametrize("distort,sensor", [(False, 3), (True, 2), (False, 0)])
def test_round_trip(tmp_path, distort, sensor):
    m = _map(np.random.default_rng(5), distort, sensor)
    path = tmp_path / "map.bin"
    n = map_io.save_map(str(path), m)
    assert n == path.stat().st_size
    r = map_io.load_map(str(path))
    assert _same(m, r), "round trip"
    # a second save of what was loaded is byte-identical
    path2 = tmp_path / "map2.bin"
    map_io.save_map(str(path2), r)
    assert path.read_bytes() == path2.read_bytes()


def test_byte_layout_known_answers(tmp_path):
    """offsets and sizes of the records as the reference's C++ writes them on x86-64"""
    m = _map(np.random.default_rng(6), False, 2)
    path = tmp_path / "m.bin"
    map_io.save_map(str(path), m)
    b = path.read_bytes()
    assert b[0] == 2                                                     # char sensorType: IMU
    imu_param_bytes = 8 + 8 + 4 * 72 + 4 + 8                             # mdMultiplyG, mdRefG, 4 Matrix3d, int, double
    o = 1 + imu_param_bytes
    assert struct.unpack_from("<3f", b, o) == (0.0, 0.0, np.float32(-9.81))   # gravity cv::Mat(3, 1, CV_32F)
    o += 12
    assert struct.unpack_from("<Q", b, o)[0] == 3                        # size_t NKFs
    o += 8
    assert struct.unpack_from("<QQ", b, o) == (0, map_io.ULONG_MAX)      # nid_ of key frame 0, no previous one
    o += 16
    kf = m["keyframes"][0]
    N = len(kf["keys"])
    assert struct.unpack_from("<d?B", b, o) == (100.0, False, 1)         # timestamp_, usedistort_, number of cameras
    o += 10
    assert struct.unpack_from("<iB", b, o) == (0, 4)                     # CameraModel kPinhole, 4 parameters
    o += 5 + 16
    assert struct.unpack_from("<i", b, o)[0] == N
    o += 4
    assert b[o:o + 28] == kf["keys"][:1].tobytes()                       # cv::KeyPoint: 28 bytes
    frame_bytes = 2 * N * 28 + 4 + 4 + N * 32 + 2 * N * 4 + 4 + 4 + 4 + 4 + 4 + 8
    #             mvKeys + mvKeysUn, mapn2in_ count, mThDepth, descriptors, vdepth_ + vuright_, v3dpoints_ count,
    #             mapcamidx2idxs_ count, baseline, n levels, scale factor, sz_dims_[2]
    o += frame_bytes
    nav = np.frombuffer(b, "<f8", 22, o)
    assert np.array_equal(nav[:3], kf["nav"]["p"]) and np.array_equal(nav[3:7], [0.5, -0.5, 0.5, 0.5])  # q as x, y, z, w
    o += 22 * 8
    assert struct.unpack_from("<Q", b, o)[0] == 4                        # encoder samples: mv[2], mtm each
    o += 8 + 4 * 24
    assert struct.unpack_from("<Q", b, o)[0] == 10                       # IMU samples: mtm, ma, mw each
    o += 8 + 10 * 56
    assert b[o] == 2                                                     # mState
    o += 1
    assert struct.unpack_from("<Q", b, o)[0] == N                        # NMPMatches, then one u64 per key
    first = struct.unpack_from("<Q", b, o + 8)[0]
    assert first == int(kf["matches"][0]) and (first == map_io.ULONG_MAX or first < 50)


def test_truncated_or_foreign_files_are_rejected(tmp_path):
    m = _map(np.random.default_rng(7))
    path = tmp_path / "m.bin"
    map_io.save_map(str(path), m)
    b = path.read_bytes()
    with pytest.raises(ValueError):
        map_io.load_map(b[:len(b) // 2])
    with pytest.raises(ValueError):
        map_io.load_map(b + b"\0")
    with pytest.raises(ValueError):
        map_io.load_map(b"\x09" + b[1:])
