"""The reference's binary sparse-map file (the data format on the storage side of the path): what
System::SaveMap(filename, bPCL = false, bUseTbc, bSaveBadKF = false) writes and System::LoadMap reads
(reference src/System.cc:272-421, FrameBase::write / read src/FrameBase.cpp:223-366, KeyFrame::write / read
src/KeyFrame.cc:138-201, MapPoint::write src/MapPoint.cc:43-56, IMUData / EncData writeParam + write
src/Odom/OdomData.h:66-101,171-198, Serialize common/serialize/serialize.h + .cpp).

Host code like the reference's (plain little-endian structs, x86-64 sizes: size_t / unsigned long = 8 bytes, bool = 1
byte, cv::KeyPoint = 28 bytes, the CameraModel enum = 4 bytes, Eigen matrices column-major).  The objects are plain
dicts / numpy arrays in the layouts the hot-path entry points take (orb_extractor.KEYPOINT_DTYPE, ba_types
.NAVSTATE_DTYPE), so a loaded map feeds the searches and the bundle adjustments directly.

File layout:
    char sensorType (bit 0 encoder, bit 1 IMU); [EncData params]; [IMUData params; gravity float32[3]]
    size_t NKFs; per key frame: nid_ (u64), previous key frame's nid_ (u64, ULONG_MAX: none), FrameBase::write,
        NavState (p, q xyzw, v, bg, ba, dbg, dba: 22 doubles), encoder list, IMU list, mState (char),
        size_t NMPMatches, the matched map points' mnId (u64, ULONG_MAX: none)
    size_t NMPs; per map point: mnId, reference key frame's nid_, float32 xyz, size_t Nobs,
        per observation: key frame nid_, size_t n, n key indices (u64)
    per key frame again (same order): parent's nid_ (ULONG_MAX: none), size_t nLoops, loop key frames' nid_
"""
import io
import struct

import numpy as np

from .ba_types import NAVSTATE_DTYPE
from .orb_extractor import KEYPOINT_DTYPE

ULONG_MAX = 0xFFFFFFFFFFFFFFFF
SENSOR_ENC, SENSOR_IMU = 1, 2


class _W:
    def __init__(self):
        self.b = io.BytesIO()

    def raw(self, fmt, *v):
        self.b.write(struct.pack("<" + fmt, *v))

    def arr(self, a, dtype):
        self.b.write(np.ascontiguousarray(a, dtype).tobytes())

    def eig(self, m):  # Serialize::writeEigMat: column-major doubles
        self.b.write(np.asfortranarray(np.asarray(m, np.float64)).tobytes(order="F"))


class _R:
    def __init__(self, data):
        self.d, self.o = memoryview(data), 0

    def raw(self, fmt):
        fmt = "<" + fmt
        n = struct.calcsize(fmt)
        if self.o + n > len(self.d):
            raise ValueError("map file truncated at byte %d" % self.o)
        v = struct.unpack_from(fmt, self.d, self.o)
        self.o += n
        return v if len(v) > 1 else v[0]

    def arr(self, dtype, count):
        dtype = np.dtype(dtype)
        n = dtype.itemsize * count
        if self.o + n > len(self.d):
            raise ValueError("map file truncated at byte %d" % self.o)
        a = np.frombuffer(self.d, dtype, count, self.o).copy()
        self.o += n
        return a

    def eig(self, rows, cols):
        return self.arr("<f8", rows * cols).reshape(cols, rows).T.copy()


def _write_framebase(w, kf):
    """FrameBase::write (FrameBase.cpp:297-360)"""
    w.raw("d", kf["timestamp"])
    w.raw("?", bool(kf["usedistort"]))
    cams = kf["cameras"]
    w.raw("B", len(cams))
    for model, params in cams:  # CameraModel enum (int), uint8 count, float parameters
        w.raw("i", int(model))
        w.raw("B", len(params))
        w.arr(params, "<f4")
    keys = np.ascontiguousarray(kf["keys"], KEYPOINT_DTYPE)
    N = len(keys)
    w.raw("i", N)
    w.b.write(keys.tobytes())
    if not kf["usedistort"]:
        w.b.write(np.ascontiguousarray(kf["keys_un"], KEYPOINT_DTYPE).tobytes())
    n2in = np.asarray(kf.get("mapn2in", np.zeros((0, 2), np.uint64)), np.uint64).reshape(-1, 2)
    w.raw("i", len(n2in))
    w.arr(n2in, "<u8")  # pair<size_t, size_t>
    w.raw("f", kf["th_depth"])
    desc = np.ascontiguousarray(kf["descriptors"], np.uint8).reshape(N, 32)
    w.b.write(desc.tobytes())  # Serialize::writeMat, N x 32 CV_8UC1
    w.arr(kf["depth"], "<f4")
    w.arr(kf["uright"], "<f4")
    p3d = np.asarray(kf.get("stereo_points", np.zeros((0, 3))), np.float64).reshape(-1, 3)
    w.raw("i", len(p3d))
    w.arr(p3d, "<f8")  # aligned_vector<Vector3d>, each written whole
    w.arr(np.asarray(kf.get("good_matches", np.zeros(len(p3d), bool)), bool).astype(np.uint8), "u1")
    c2i = np.asarray(kf.get("camidx2idxs", np.zeros((0, 3), np.uint64)), np.uint64).reshape(-1, 3)
    w.raw("i", len(c2i))
    w.arr(c2i, "<u8")  # pair<pair<size_t, size_t>, size_t> in the container's order
    w.raw("f", kf["baseline"])
    w.raw("i", int(kf["n_levels"]))
    w.raw("f", kf["scale_factor"])
    w.raw("ii", int(kf["image_size"][0]), int(kf["image_size"][1]))


def _read_framebase(r):
    kf = {}
    kf["timestamp"] = r.raw("d")
    kf["usedistort"] = bool(r.raw("?"))
    cams = []
    for _ in range(r.raw("B")):
        model = r.raw("i")
        cams.append((model, r.arr("<f4", r.raw("B"))))
    kf["cameras"] = cams
    N = r.raw("i")
    kf["keys"] = r.arr(KEYPOINT_DTYPE, N)
    if not kf["usedistort"]:
        kf["keys_un"] = r.arr(KEYPOINT_DTYPE, N)
    kf["mapn2in"] = r.arr("<u8", 2 * r.raw("i")).reshape(-1, 2)
    kf["th_depth"] = r.raw("f")
    kf["descriptors"] = r.arr("u1", N * 32).reshape(N, 32)
    kf["depth"] = r.arr("<f4", N)
    kf["uright"] = r.arr("<f4", N)
    n = r.raw("i")
    kf["stereo_points"] = r.arr("<f8", 3 * n).reshape(n, 3)
    kf["good_matches"] = r.arr("u1", n).astype(bool)
    kf["camidx2idxs"] = r.arr("<u8", 3 * r.raw("i")).reshape(-1, 3)
    kf["baseline"] = r.raw("f")
    kf["n_levels"] = r.raw("i")
    kf["scale_factor"] = r.raw("f")
    kf["image_size"] = r.raw("ii")
    return kf


def _write_nav(w, nav):
    """KeyFrame::write (KeyFrame.cc:166-184): p, q as Eigen coeffs (x, y, z, w), v, bg, ba, dbg, dba"""
    n = np.asarray(nav, NAVSTATE_DTYPE).reshape(())
    w.arr(n["p"], "<f8")
    q = n["q"]
    w.arr([q[1], q[2], q[3], q[0]], "<f8")
    for f in ("v", "bg", "ba", "dbg", "dba"):
        w.arr(n[f], "<f8")


def _read_nav(r):
    nav = np.zeros((), NAVSTATE_DTYPE)
    nav["p"] = r.arr("<f8", 3)
    x, y, z, qw = r.arr("<f8", 4)
    nav["q"] = (qw, x, y, z)
    for f in ("v", "bg", "ba", "dbg", "dba"):
        nav[f] = r.arr("<f8", 3)
    return nav


def save_map(path_or_file, m):
    """m: dict(sensor_type, [enc_params], [imu_params, gravity], keyframes [...], mappoints [...]).
    key frame: dict(id, prev_id (None), <FrameBase fields>, nav NAVSTATE, enc_list float64[n, 3] (vl, vr, t),
    imu_list float64[n, 7] (t, a xyz, w xyz), state (int), matches uint64[N] (ULONG_MAX: none), parent_id (None),
    loop_ids [...]); map point: dict(id, ref_kf_id, pos float32[3], observations [(kf_id, [key indices])])."""
    w = _W()
    st = int(m["sensor_type"])
    w.raw("b", st)
    if st & SENSOR_ENC:  # EncData::writeParam
        e = m["enc_params"]
        w.raw("dd", e["vscale"], e["rc"])
        w.eig(np.asarray(e["Sigma"]).reshape(2, 2))
        w.eig(np.asarray(e["Sigmam"]).reshape(6, 6))
        w.raw("id", int(e["dt_cov_noise_fixed"]), e["freq_ref"])
    if st & SENSOR_IMU:  # IMUData::writeParam + gravity (cv::Mat 3x1 CV_32F)
        p = m["imu_params"]
        w.raw("dd", p["multiply_g"], p["ref_g"])
        for k in ("Sigma_g", "Sigma_a", "Sigma_bg", "Sigma_ba"):
            w.eig(np.asarray(p[k]).reshape(3, 3))
        w.raw("id", int(p["dt_cov_noise_fixed"]), p["freq_ref"])
        w.arr(m["gravity"], "<f4")
    kfs = m["keyframes"]
    w.raw("Q", len(kfs))
    for kf in kfs:
        w.raw("Q", int(kf["id"]))
        w.raw("Q", ULONG_MAX if kf.get("prev_id") is None else int(kf["prev_id"]))
        _write_framebase(w, kf)
        _write_nav(w, kf["nav"])
        enc = np.asarray(kf.get("enc_list", np.zeros((0, 3))), np.float64).reshape(-1, 3)
        w.raw("Q", len(enc))
        w.arr(enc, "<f8")  # EncData::write: mv[2], mtm
        imu = np.asarray(kf.get("imu_list", np.zeros((0, 7))), np.float64).reshape(-1, 7)
        w.raw("Q", len(imu))
        w.arr(imu, "<f8")  # IMUData::write: mtm, ma, mw
        w.raw("b", int(kf.get("state", 2)))
        mt = np.asarray(kf["matches"], np.uint64)
        w.raw("Q", len(mt))
        w.arr(mt, "<u8")
    mps = m["mappoints"]
    w.raw("Q", len(mps))
    for p in mps:
        w.raw("QQ", int(p["id"]), int(p["ref_kf_id"]))
        w.arr(p["pos"], "<f4")
        obs = p["observations"]
        w.raw("Q", len(obs))
        for kid, idxs in obs:
            w.raw("QQ", int(kid), len(idxs))
            w.arr(idxs, "<u8")
    for kf in kfs:
        w.raw("Q", ULONG_MAX if kf.get("parent_id") is None else int(kf["parent_id"]))
        loops = list(kf.get("loop_ids", []))
        w.raw("Q", len(loops))
        w.arr(loops, "<u8")
    data = w.b.getvalue()
    if hasattr(path_or_file, "write"):
        path_or_file.write(data)
    else:
        with open(path_or_file, "wb") as f:
            f.write(data)
    return len(data)


def load_map(path_or_bytes):
    """The inverse of save_map (System::LoadMap reads the same fields in the same order)."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        data = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    r = _R(data)
    m = {"sensor_type": r.raw("b")}
    st = m["sensor_type"]
    if st < 0 or st > 3:
        raise ValueError("not a VIEO_SLAM sparse map: sensor type %d" % st)
    if st & SENSOR_ENC:
        e = {}
        e["vscale"], e["rc"] = r.raw("dd")
        e["Sigma"], e["Sigmam"] = r.eig(2, 2), r.eig(6, 6)
        e["dt_cov_noise_fixed"], e["freq_ref"] = r.raw("id")
        m["enc_params"] = e
    if st & SENSOR_IMU:
        p = {}
        p["multiply_g"], p["ref_g"] = r.raw("dd")
        for k in ("Sigma_g", "Sigma_a", "Sigma_bg", "Sigma_ba"):
            p[k] = r.eig(3, 3)
        p["dt_cov_noise_fixed"], p["freq_ref"] = r.raw("id")
        m["imu_params"] = p
        m["gravity"] = r.arr("<f4", 3)
    kfs = []
    for _ in range(r.raw("Q")):
        kid, prev = r.raw("QQ")
        kf = _read_framebase(r)
        kf["id"], kf["prev_id"] = kid, None if prev == ULONG_MAX else prev
        kf["nav"] = _read_nav(r)
        kf["enc_list"] = r.arr("<f8", 3 * r.raw("Q")).reshape(-1, 3)
        kf["imu_list"] = r.arr("<f8", 7 * r.raw("Q")).reshape(-1, 7)
        kf["state"] = r.raw("b")
        kf["matches"] = r.arr("<u8", r.raw("Q"))
        kfs.append(kf)
    mps = []
    for _ in range(r.raw("Q")):
        p = {}
        p["id"], p["ref_kf_id"] = r.raw("QQ")
        p["pos"] = r.arr("<f4", 3)
        obs = []
        for _ in range(r.raw("Q")):
            kid, n = r.raw("QQ")
            obs.append((kid, r.arr("<u8", n)))
        p["observations"] = obs
        mps.append(p)
    for kf in kfs:
        par = r.raw("Q")
        kf["parent_id"] = None if par == ULONG_MAX else par
        kf["loop_ids"] = r.arr("<u8", r.raw("Q")).tolist()
    if r.o != len(data):
        raise ValueError("map file has %d trailing bytes" % (len(data) - r.o))
    m["keyframes"], m["mappoints"] = kfs, mps
    return m


def map_from_replay(R):
    """The sparse map of a finished vieo_slam_amd.replay.Replay run in the file's terms (sensor type 2: IMU)."""
    from . import synth_ba
    from . import synth_scene as sc
    seq = R.seq
    kfs = []
    for k in R.kfs:
        samples = seq.imu_between(R.kfs[k.id - 1].t, k.t) if k.id > 0 else seq.imu[:0]
        imu = np.stack([samples["t"], *samples["a"].T, *samples["w"].T], 1) if len(samples) else np.zeros((0, 7))
        mt = np.where(k.mp_ref >= 0, k.mp_ref, -1).astype(np.int64).astype(np.uint64)  # -1 -> ULONG_MAX
        kfs.append(dict(id=k.id, prev_id=k.id - 1 if k.id > 0 else None, timestamp=k.t, usedistort=False,
                        cameras=[(0, np.array([sc.FX, sc.FY, sc.CX, sc.CY], np.float32))], keys=k.keys, keys_un=k.keys,
                        th_depth=35.0 * sc.BASELINE, descriptors=k.desc, depth=k.depth, uright=k.uright,
                        baseline=sc.BASELINE, n_levels=8, scale_factor=1.2, image_size=(sc.W, sc.H), nav=k.nav,
                        imu_list=imu, state=2, matches=mt, parent_id=k.id - 1 if k.id > 0 else None, loop_ids=[]))
    mps = []
    for m in range(len(R.mp_X)):
        if R.mp_bad[m] or not R.mp_obs[m]:
            continue
        obs = [(kid, [i]) for kid, i in sorted(R.mp_obs[m].items())]
        mps.append(dict(id=m, ref_kf_id=min(R.mp_obs[m]), pos=R.mp_X[m], observations=obs))
    s2 = [x ** 2 for x in synth_ba.IMU_SIGMA]
    imu_params = dict(multiply_g=1.0, ref_g=9.81, Sigma_g=np.eye(3) * s2[0] * synth_ba.IMU_FREQ,
                      Sigma_a=np.eye(3) * s2[1] * synth_ba.IMU_FREQ, Sigma_bg=np.eye(3) * s2[2], Sigma_ba=np.eye(3) * s2[3],
                      dt_cov_noise_fixed=1, freq_ref=0.0)
    return dict(sensor_type=SENSOR_IMU, imu_params=imu_params, gravity=synth_ba.GRAVITY.astype(np.float32),
                keyframes=kfs, mappoints=mps)
