"""ctypes mirror of the one-call frame tracker (include/vieo_hot.h: vieo_tracker_*, vieo_track_frame) and a replay
that drives it: `TrackerReplay` is `replay.Replay` with a frame's tracking done by ONE C-ABI call (the C++ form of
what `replay.ChainedReplay` issues from Python launch by launch).  The map bookkeeping stays in the driver, as it
stays in Tracking / LocalMapping in the reference."""
import ctypes
import time

import numpy as np

from . import replay as rp
from . import synth_ba
from . import synth_scene as sc
from ._lib import check, lib
from .ba_types import CAMERA_DTYPE, IMU_PREINT_DTYPE, LAST_FRAME_POINT_DTYPE, NAVSTATE_DTYPE, VIO_RESULT_DTYPE
from .imu import IMU_NOISE_DTYPE, IMU_SAMPLE_DTYPE
from .map_point import FRUSTUM_POINT_DTYPE
from .orb_extractor import KEYPOINT_DTYPE

TRACKER_PARAMS_DTYPE = np.dtype([
    ("width", "<i4"), ("height", "<i4"), ("n_features", "<i4"), ("n_levels", "<i4"), ("ini_th_fast", "<i4"),
    ("min_th_fast", "<i4"), ("scale_factor", "<f4"), ("fx", "<f4"), ("fy", "<f4"), ("cx", "<f4"), ("cy", "<f4"),
    ("bf", "<f4"), ("baseline", "<f4"), ("th_depth", "<f4"), ("th_last", "<f4"), ("th_local", "<f4"), ("nn_last", "<f4"),
    ("nn_local", "<f4"), ("max_local_points", "<i4"), ("Rcb", "<f8", 9), ("tcb", "<f8", 3), ("gw", "<f8", 3),
    ("inv_sigma_bg2", "<f8"), ("inv_sigma_ba2", "<f8"), ("noise", IMU_NOISE_DTYPE), ("vision_only", "<i4"),
    ("reserved", "<i4")], align=True)
TRACKER_RIG_DTYPE = np.dtype([
    ("n_cams", "<i4"), ("use_lapping", "<i4"), ("lapping", "<i4", 2), ("th_far_pts", "<f4"), ("reserved", "<f4"),
    ("cams", CAMERA_DTYPE, 4), ("Trc", "<f8", (4, 12)), ("Tcr", "<f8", (4, 12))], align=True)
TRACK_INPUT_DTYPE = np.dtype([
    ("left", "<u8"), ("right", "<u8"), ("stride", "<i4"), ("n_imu", "<i4"), ("imu", "<u8"), ("t_ref", "<f8"),
    ("t_cur", "<f8"), ("nav_ref", NAVSTATE_DTYPE), ("nav_last", NAVSTATE_DTYPE), ("nav_prior", "<u8"), ("H_prior", "<u8"),
    ("n_last", "<i4"), ("last_points", "<u8"), ("last_track_depth", "<u8"), ("n_local", "<i4"), ("local_version", "<i4"),
    ("local_points", "<u8"), ("local_desc", "<u8"), ("local_alias", "<u8"), ("images", "<u8", 4), ("next_left", "<u8"),
    ("next_right", "<u8"), ("use_prefetched", "<i4"), ("next_n_imu", "<i4"), ("next_imu", "<u8"), ("next_t_cur", "<f8"), ("next_images", "<u8", 4), ("next_ref_bias", "<u8"),
    ("next_t_ref", "<f8")],
    align=True)
TRACK_OUTPUT_DTYPE = np.dtype([
    ("status", "<i4"), ("n_keys", "<i4"), ("key_cap", "<i4"), ("keys", "<u8"), ("desc", "<u8"), ("uright", "<u8"),
    ("depth", "<u8"), ("point_ref", "<u8"), ("outlier", "<u8"), ("local_track_depth", "<u8"), ("n_matches_last", "<i4"),
    ("n_matches_local", "<i4"), ("widened", "<i4"), ("nav_pred", NAVSTATE_DTYPE), ("imu", IMU_PREINT_DTYPE),
    ("preint_status", "<i4"), ("reserved", "<i4"), ("first", VIO_RESULT_DTYPE), ("second", VIO_RESULT_DTYPE),
    ("ms_gpu", "<f4"), ("ms_host", "<f4"), ("cam_first", "<i4", 5), ("mono_index", "<i4", 4), ("stereo_status", "<i4"),
    ("n_groups", "<i4"), ("n_stereo_matches", "<i4"), ("key_group", "<u8"), ("group_idx", "<u8"), ("group_good", "<u8"),
    ("group_p3d", "<u8")], align=True)

def _bind():
    return lib()  # signatures: _lib._SIGS


def euroc_params(max_local_points=16384, th_last=7.0, th_local=2.0, noise=None):
    """vieo_tracker_params of the replay's rendered EuRoC-like stereo rig (synth_scene)."""
    P = np.zeros(1, TRACKER_PARAMS_DTYPE)
    p = P[0]
    p["width"], p["height"] = rp.W, rp.H
    p["n_features"], p["n_levels"], p["ini_th_fast"], p["min_th_fast"], p["scale_factor"] = (rp.NFEAT, rp.NLEVELS, rp.INI_TH,
                                                                                             rp.MIN_TH, rp.SCALE)
    p["fx"], p["fy"], p["cx"], p["cy"], p["bf"], p["baseline"] = sc.FX, sc.FY, sc.CX, sc.CY, sc.BF, sc.BASELINE
    p["th_depth"], p["th_last"], p["th_local"], p["nn_last"], p["nn_local"] = rp.TH_DEPTH, th_last, th_local, 0.9, 0.8
    p["max_local_points"] = max_local_points
    Tcb = np.linalg.inv(synth_ba.EUROC_TBC)
    p["Rcb"], p["tcb"] = Tcb[:3, :3].reshape(-1), Tcb[:3, 3]
    p["gw"] = synth_ba.GRAVITY
    p["inv_sigma_bg2"], p["inv_sigma_ba2"] = 1.0 / synth_ba.IMU_SIGMA[2] ** 2, 1.0 / synth_ba.IMU_SIGMA[3] ** 2
    if noise is not None:
        p["noise"] = noise
    return P


def rig_params(scene, nfeatures, max_local_points=8192, th_last=7.0, th_local=2.0, noise=None, th_depth=35.0):
    """(vieo_tracker_params, vieo_tracker_rig) of a synth_scene.RigScene (the configuration pipeline_rig.RigFrontEnd runs
    stage by stage): KB8 cameras hand over their lapping area, bf = 0.11 * fx of camera 0."""
    from . import synth_fisheye as sf
    P = np.zeros(1, TRACKER_PARAMS_DTYPE)
    p = P[0]
    c0 = scene.cams[0]
    p["width"], p["height"] = scene.W, scene.H
    p["n_features"], p["n_levels"], p["ini_th_fast"], p["min_th_fast"], p["scale_factor"] = nfeatures, 8, 20, 7, 1.2
    p["fx"], p["fy"], p["cx"], p["cy"] = c0["fx"], c0["fy"], c0["cx"], c0["cy"]
    p["bf"] = 0.11 * float(c0["fx"])
    p["baseline"] = p["bf"] / np.float32(c0["fx"])
    p["th_depth"], p["th_last"], p["th_local"], p["nn_last"], p["nn_local"] = th_depth, th_last, th_local, 0.9, 0.8
    p["max_local_points"] = max_local_points
    p["Rcb"], p["tcb"] = scene.Tcb[:3, :3].reshape(-1), scene.Tcb[:3, 3]
    p["gw"] = synth_ba.GRAVITY
    p["inv_sigma_bg2"], p["inv_sigma_ba2"] = 1.0 / synth_ba.IMU_SIGMA[2] ** 2, 1.0 / synth_ba.IMU_SIGMA[3] ** 2
    if noise is not None:
        p["noise"] = noise
    else:  # EuRoC sigmas, as replay.Sequence builds them
        p["noise"]["sigma_g"] = (np.eye(3) * synth_ba.IMU_SIGMA[0] ** 2 * synth_ba.IMU_FREQ).reshape(-1)
        p["noise"]["sigma_a"] = (np.eye(3) * synth_ba.IMU_SIGMA[1] ** 2 * synth_ba.IMU_FREQ).reshape(-1)
        p["noise"]["freq_ref"], p["noise"]["dt_cov_noise_fixed"] = synth_ba.IMU_FREQ, 1
    R = np.zeros(1, TRACKER_RIG_DTYPE)
    r = R[0]
    nc = len(scene.cams)
    r["n_cams"] = nc
    if int(c0["model"]) == 2:
        r["use_lapping"], r["lapping"] = 1, (0, scene.W - 1)
    r["cams"][:nc] = scene.cams
    Trc, Tcr = sf.rig_extrinsics(scene.Tcr)
    r["Trc"][:nc], r["Tcr"][:nc] = Trc, Tcr
    return P, R


def _view(ptr, dtype, count):
    dtype = np.dtype(dtype)
    if count == 0 or not ptr:
        return np.zeros(0, dtype)
    buf = (ctypes.c_uint8 * (dtype.itemsize * count)).from_address(int(ptr))
    return np.frombuffer(buf, dtype, count)


class Tracker:
    """vieo_tracker: one call per frame.  `track` returns the output record plus numpy views of the per-key arrays
    (valid until the next call).  rig: TRACKER_RIG_DTYPE[1] -> a distorted camera-rig tracker (images = one per camera);
    params["vision_only"] = 1 -> the stereo tracker without IMU (nav_ref = the predicted state)."""

    def __init__(self, params, rig=None):
        L = _bind()
        self.params = np.ascontiguousarray(params, TRACKER_PARAMS_DTYPE).reshape(1)
        self.rig = None if rig is None else np.ascontiguousarray(rig, TRACKER_RIG_DTYPE).reshape(1)
        self.n_img = 2 if rig is None else int(self.rig[0]["n_cams"])
        h = ctypes.c_void_p()
        check(L.vieo_tracker_create_rig(ctypes.byref(h), self.params.ctypes.data,
                                        None if rig is None else self.rig.ctypes.data), "vieo_tracker_create_rig")
        self.h = h
        w, hh = int(self.params[0]["width"]), int(self.params[0]["height"])
        self.planes, self._ptrs = [], []
        for c in range(self.n_img):
            a = ctypes.c_void_p()
            check(L.vieo_tracker_image_buffer(h, c, ctypes.byref(a)))
            self.planes.append(np.ctypeslib.as_array((ctypes.c_uint8 * (w * hh)).from_address(a.value)).reshape(hh, w))
            self._ptrs.append(a.value)
        self.left, self.right = self.planes[0], self.planes[1]
        self.key_cap = L.vieo_tracker_key_capacity(h)
        self.inp = np.zeros(1, TRACK_INPUT_DTYPE)
        self.out = np.zeros(1, TRACK_OUTPUT_DTYPE)

    def close(self):
        if self.h:
            _bind().vieo_tracker_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    STATS_DTYPE = np.dtype([("side_stream_ratio", "<f4"), ("side_stream_selections", "<i4"), ("side_stream_checks", "<i4"),
                            ("replica_repeats", "<i4"), ("ms_gpu_median", "<f4"), ("slow_frames_in_a_row", "<i4"),
                            ("frames_prefetched", "<i4"), ("preints_ahead_used", "<i4")])

    def stats(self):
        """vieo_tracker_get_stats: the second stream's measured overlap ratio, re-selections, replica repeats, ..."""
        st = np.zeros(1, self.STATS_DTYPE)
        check(_bind().vieo_tracker_get_stats(self.h, st.ctypes.data), "vieo_tracker_get_stats")
        return {k: st[0][k].item() for k in st.dtype.names}

    def reprobe(self):
        check(_bind().vieo_tracker_reprobe(self.h), "vieo_tracker_reprobe")
        return self.stats()

    def scale_factors(self):
        s = np.zeros(int(self.params[0]["n_levels"]), np.float32)
        check(_bind().vieo_tracker_scale_factors(self.h, s.ctypes.data))
        return s

    def track(self, left, right, imu, t_ref, t_cur, nav_ref, nav_last, prior, last_points, last_track_depth, local_points,
              local_desc, local_alias, local_version, images=None, next_images=None, use_prefetched=False, next_imu=None):
        """next_images = (left, right) of the frame the NEXT call will track: its extraction + stereo stage run beside this
        frame's searches and optimisations; that call passes use_prefetched=True (its own images are then not read).
        next_imu = (samples of [t_cur, t_next], t_next): the next call's pre-integration, run ahead (used by that call only
        when its reference turns out to be this frame)."""
        i = self.inp[0]
        keep = []
        imgs = [left, right] if images is None else list(images)
        if use_prefetched:
            imgs = [self.planes[c] for c in range(self.n_img)]  # (not read)
        assert len(imgs) == self.n_img
        i["next_left"] = i["next_right"] = 0
        i["next_images"] = 0
        i["use_prefetched"] = int(bool(use_prefetched))
        if next_images is not None:
            nxt = [np.ascontiguousarray(x, np.uint8) for x in next_images]
            assert len(nxt) == self.n_img
            keep += nxt
            if self.rig is not None:
                i["next_images"][:self.n_img] = [x.ctypes.data for x in nxt]
            else:
                i["next_left"], i["next_right"] = nxt[0].ctypes.data, nxt[1].ctypes.data
        i["next_imu"], i["next_n_imu"], i["next_t_cur"] = 0, 0, 0.0
        if next_imu is not None:
            ns = np.ascontiguousarray(next_imu[0], IMU_SAMPLE_DTYPE)
            keep.append(ns)
            i["next_imu"], i["next_n_imu"], i["next_t_cur"] = ns.ctypes.data, len(ns), float(next_imu[1])
        ptrs = []
        for c, img in enumerate(imgs):
            if img is self.planes[c]:
                ptrs.append(self._ptrs[c])
            else:
                img = np.ascontiguousarray(img, np.uint8)
                keep.append(img)
                ptrs.append(img.ctypes.data)
        i["left"], i["right"] = ptrs[0], ptrs[1]
        i["images"] = 0
        if self.rig is not None:
            i["images"][:self.n_img] = ptrs
        i["stride"] = int(self.params[0]["width"])
        imu = np.ascontiguousarray(imu, IMU_SAMPLE_DTYPE)
        i["n_imu"], i["imu"] = len(imu), imu.ctypes.data
        i["t_ref"], i["t_cur"], i["nav_ref"], i["nav_last"] = t_ref, t_cur, nav_ref, nav_last
        if prior is not None:
            pn = np.zeros(1, NAVSTATE_DTYPE)
            pn[0] = prior[0]
            pH = np.ascontiguousarray(prior[1], np.float64)
            keep += [pn, pH]
            i["nav_prior"], i["H_prior"] = pn.ctypes.data, pH.ctypes.data
        else:
            i["nav_prior"] = i["H_prior"] = 0
        lp = np.ascontiguousarray(last_points, LAST_FRAME_POINT_DTYPE)
        ld = np.ascontiguousarray(last_track_depth, np.float32)
        i["n_last"], i["last_points"], i["last_track_depth"] = len(lp), lp.ctypes.data, ld.ctypes.data
        cp = np.ascontiguousarray(local_points, FRUSTUM_POINT_DTYPE)
        cd = np.ascontiguousarray(local_desc, np.uint8)
        ca = np.ascontiguousarray(local_alias, np.int32)
        i["n_local"], i["local_version"] = len(cp), int(local_version)
        i["local_points"], i["local_desc"], i["local_alias"] = cp.ctypes.data, cd.ctypes.data, ca.ctypes.data
        check(_bind().vieo_track_frame(self.h, self.inp.ctypes.data, self.out.ctypes.data), "vieo_track_frame")
        o = self.out[0]
        n, nc = int(o["n_keys"]), len(cp)
        v = dict(keys=_view(o["keys"], KEYPOINT_DTYPE, n), desc=_view(o["desc"], np.uint8, 32 * n).reshape(n, 32),
                 uright=_view(o["uright"], np.float32, n), depth=_view(o["depth"], np.float32, n),
                 point_ref=_view(o["point_ref"], np.int32, n), outlier=_view(o["outlier"], np.uint8, n),
                 local_track_depth=_view(o["local_track_depth"], np.float32, nc))
        if self.rig is not None:
            g, ncam = int(o["n_groups"]), self.n_img
            v.update(key_group=_view(o["key_group"], np.int32, n),
                     group_idx=_view(o["group_idx"], np.int32, g * ncam).reshape(g, ncam),
                     group_good=_view(o["group_good"], np.uint8, g).astype(bool),
                     group_p3d=_view(o["group_p3d"], np.float64, g * 3).reshape(g, 3))
        return o, v


class TrackerReplay(rp.Replay):
    """The sequential replay with every frame's tracking as ONE vieo_track_frame call."""

    def __init__(self, seq, stages, max_local_points=16384, prefetch=False, preint_ahead=False, **kw):
        super().__init__(seq, stages, **kw)
        self.preint_ahead = bool(preint_ahead)  # next_imu without next images (the run-ahead pre-integration alone)
        self.trk = Tracker(euroc_params(max_local_points, self.th_last, self.th_local, seq.noise[0]))
        self._lv = 0
        # frame pipelining: frame k + 1's images go along with frame k's call (vieo_track_input.next_left / next_right)
        self.prefetch, self._prefetched, self._n_run = bool(prefetch), False, 0
        self.stats["ms_chain"] = []
        self.stats["widened"] = 0

    def close(self):
        self.trk.close()

    def run(self, n_frames=None):
        self._n_run = n_frames or self.seq.n_frames
        return super().run(n_frames)

    def _all_local_points(self):
        key = (len(self.kfs), self.stats["lba_applied"])
        if getattr(self, "_lp_key", None) != key:
            out, seen = [], set()
            for k in self.kfs[-self.n_local_kfs:]:
                for m in k.mp_ref[k.mp_ref >= 0]:
                    m = int(m)
                    if m not in seen and not self.mp_bad[m]:
                        seen.add(m)
                        out.append(m)
            self._lp_key, self._lp = key, np.array(out, np.int64)
            self._lv += 1
            cp = np.zeros(len(out), FRUSTUM_POINT_DTYPE)
            cp["Xw"], cp["normal"] = self.mp_X[self._lp], self.mp_normal[self._lp]
            cp["max_distance"], cp["min_distance"] = self.mp_maxd[self._lp], self.mp_mind[self._lp]
            self._lp_pts, self._lp_desc = cp, self.mp_desc[self._lp].copy()
        return self._lp

    def step(self, k):
        from . import frontend
        t0 = time.perf_counter()
        last = self.last
        ref_nav = self.kfs[-1].nav if self.map_updated else last.nav
        prior = None if self.map_updated else last.prior
        t_ref = self.kfs[-1].t if self.map_updated else last.t
        t = self.seq.time(k)
        Li, Ri = self.seq.images(k)
        has = (last.mp_ref >= 0) & ~last.outlier
        has[has] &= ~self.mp_bad[last.mp_ref[has]]
        nl = last.N
        Xw = np.zeros((nl, 3), np.float32)
        Xw[has] = self.mp_X[last.mp_ref[has]]
        pts = frontend.make_last_frame_points(last.keys, np.zeros((nl, 32), np.uint8), Xw, has, True)
        pts["desc"][has] = self.mp_desc[last.mp_ref[has]]
        cand = self._all_local_points()
        where = np.full(len(self.mp_X), -1, np.int32)
        lk = np.nonzero(has)[0]
        where[last.mp_ref[lk[::-1]]] = lk[::-1]
        alias = where[cand] if len(cand) else np.zeros(0, np.int32)
        use_pf = self.prefetch and self._prefetched
        nxt = self.seq.images(k + 1) if (self.prefetch and k + 1 < self._n_run) else None
        self._prefetched = nxt is not None
        nxt_imu = None
        if nxt is not None or (self.preint_ahead and k + 1 < self._n_run):
            t_next = self.seq.time(k + 1)
            nxt_imu = (self.seq.imu_between(t, t_next), t_next)
        o, v = self.trk.track(Li, Ri, self.seq.imu_between(t_ref, t), t_ref, t, ref_nav, last.nav, prior, pts,
                              last.track_depth, self._lp_pts, self._lp_desc, alias, self._lv, next_images=nxt,
                              use_prefetched=use_pf, next_imu=nxt_imu)
        assert int(o["status"]) == 0, "IMU pre-integration failed"
        self.stats["ms_chain"].append((float(o["ms_host"]), float(o["ms_gpu"])))
        self.stats["widened"] += int(o["widened"])
        cap = int(o["key_cap"])
        f = rp._Frame()
        f.k, f.t = k, t
        N = f.N = int(o["n_keys"])
        f.keys, f.desc = v["keys"].copy(), v["desc"].copy()
        f.uright, f.depth = v["uright"].copy(), v["depth"].copy()
        tab = v["point_ref"]
        f.mp_ref = np.full(N, -1, np.int64)
        f.track_depth = np.full(N, np.inf, np.float32)
        a = np.nonzero((tab >= 0) & (tab < cap))[0]
        f.mp_ref[a] = last.mp_ref[tab[a]]
        f.track_depth[a] = last.track_depth[tab[a]]
        b = np.nonzero(tab >= cap)[0]
        f.mp_ref[b] = cand[tab[b] - cap]
        f.track_depth[b] = v["local_track_depth"][tab[b] - cap]
        f.outlier = v["outlier"].astype(bool)
        r1, r2 = o["first"], o["second"]
        nav1 = r1["base"]["nav"]
        f.nav = (r2["base"]["nav"] if int(r2["base"]["status"]) == 0 else nav1).copy()
        f.prior = (f.nav.copy(), r2["H_marg"].copy()) if int(r2["has_marg"]) else None
        self.map_updated = False
        self.stats["n_matches"].append((int(o["n_matches_last"]), int(o["n_matches_local"])))
        self.stats["n_inliers"].append(int(r2["base"]["n_inliers"]))
        return self._finish_frame(k, f, t0)
