"""The sequential replay of replay.py for the other configurations of BASELINE.json -- ONE stream of frames with map growth,
a key frame every `kf_every` frames and its local bundle adjustment applied with the LocalMapping lag:

  RigReplay / RigTrackerReplay        distorted camera rigs, stereo + IMU: the reference's DEFAULT MH05 set-up (2 Radtan
                                      cameras, Examples/RunEuRoC/RunEuRoCVIO.sh:17-18), configs[3] (4 KB8 cameras) and
                                      configs[4] (TUM-VI: 2 KB8 cameras, 1500 features).  Frame::Frame = ExtractORB x
                                      n_cams + ComputeStereoFishEyeMatches (src/Frame.cc:259-316,613-779), the camera loop
                                      in both SearchByProjection overloads, per-camera edges in PoseOptimization and in
                                      LocalBundleAdjustmentNavStatePRV (src/Optimizer.cc:21-769); new map points from the
                                      good stereo groups (Tracking::CreateNewKeyFrame, src/Tracking.cc:2168-2306).
  VisionReplay / VisionTrackerReplay  configs[0]: rectified stereo WITHOUT IMU, 1000 features: TrackWithMotionModel +
                                      TrackLocalMap (src/Tracking.cc:1843-2008) with the constant-velocity prediction and
                                      UpdateLastFrame, vision-only PoseOptimization x 2, LocalBundleAdjustment
                                      (src/Optimizer.cc:1876-2307).

*Replay runs the frame stage by stage on a `stages` object (the C-ABI's host-pointer entries, or the CPU oracle in the
tests: tests/replay_oracle.py); *TrackerReplay runs a frame's tracking as ONE vieo_track_frame call (the map, the key
frames and the local BA are the driver's, shared with the staged class).  Same simplifications as replay.Replay."""
import time
import zlib

import numpy as np

from . import frontend, replay, synth_ba
from . import synth_scene as sc
from .ba_types import (CAMERA_DTYPE, LAST_FRAME_POINT_DTYPE, LBA_IMU_EDGE_DTYPE, LBA_KEYFRAME_DTYPE, LBA_OBS_DTYPE, LBA_PARAMS_DTYPE,
                       LBA_VIO_PARAMS_DTYPE, NAVSTATE_DTYPE, POSE_FRAME_DTYPE, POSE_OBS_DTYPE, VIO_FRAME_DTYPE)
from .map_point import FRUSTUM_FRAME_DTYPE, FRUSTUM_POINT_DTYPE

NLEVELS = 8


# ================================================================ camera rigs
class RigSequence(replay.Sequence):
    """replay.Sequence (trajectory, IMU samples) seen through the distorted cameras of synth_scene.RigScene."""

    def __init__(self, seed, n_frames, rig="radtan", n_cams=2, dt=0.05):
        super().__init__(seed, n_frames, dt)
        self.scene = sc.RigScene(seed, rig, n_cams)
        self.rig_name, self.n_cams = rig, n_cams

    def images(self, k):
        if k not in self._img:
            R, p, _, _, _ = self.state(self.time(k))
            self._img[k] = self.scene.frame(R, p, 1000 * self.seed + 10 * k)[0]
        return self._img[k]


class HipRigStages:
    """the hot-path calls of the rig replay through libvieo_hot.so (host-pointer entries)"""
    name = "hip"

    def __init__(self, nfeatures, n_cams):
        from .pipeline_rig import HipStages
        self.P = HipStages(nfeatures, n_cams)
        for n in ("extract", "fisheye", "project_last_frame", "search", "pose_vio", "in_frustum"):
            setattr(self, n, getattr(self.P, n))

    def scale_factors(self):
        return self.P.ext[0].GetScaleFactors()

    def preintegrate(self, noise, samples, ti, tj, bg, ba):
        from .imu import imu_preintegrate
        out, prv, st = imu_preintegrate(noise, [samples], [ti], [tj], [bg], [ba])
        return out[0], prv[0], int(st[0])

    def lba_vio(self, params, kfs, pts, close, obs, imu):
        from .optimizer import Optimizer
        return Optimizer.LocalBundleAdjustmentNavStatePRV(params, kfs, pts, close, obs, imu)

    def update_normal_depth(self, *a):
        from .map_point import update_normal_and_depth
        return update_normal_and_depth(*a)


class RigReplay(replay.Replay):
    def __init__(self, seq, stages, nfeatures=1200, **kw):
        from .pipeline_rig import RigFrontEnd
        super().__init__(seq, stages, **kw)
        self.scene = seq.scene
        self.nc = len(self.scene.cams)
        self.fe = RigFrontEnd(self.scene, nfeatures, stages=stages, th_last=self.th_last, th_local=self.th_local)
        self.th_depth = 35.0

    # ---- Frame::Frame
    def make_frame(self, k):
        fr = self.fe.make_frame(self.seq.images(k))
        f = replay._Frame()
        f.k, f.t = k, self.seq.time(k)
        f.rf = fr
        f.keys, f.desc, f.uright, f.depth, f.N = fr.keys, fr.desc, fr.uright, fr.depth, fr.N
        f.key_cam, f.cam_first = fr.key_cam, fr.cam_first
        f.key_group = fr.fe["key_group"]
        f.mp_ref = np.full(f.N, -1, np.int64)
        f.track_depth = np.full(f.N, np.inf, np.float32)
        return f

    # ---- Tracking::CreateNewKeyFrame: a point per good stereo group none of whose keys holds one; all the keys of the
    #      group then hold it (and observe it: one observation per (key frame, camera))
    def insert_keyframe(self, f, nav, imu_edge):
        kf = f
        kf.id = len(self.kfs)
        kf.nav = nav.copy()
        _, kf.Rwc, kf.twc = replay._Tcw_of(kf.nav, self.Tbc)
        kf.imu_edge = imu_edge
        self.kfs.append(kf)
        for i in np.nonzero(kf.mp_ref >= 0)[0]:  # AddObservation
            self._add_obs(int(kf.mp_ref[i]), kf.id, int(i))
        fe = kf.rf.fe
        groups = []
        for g in np.nonzero(fe["group_good"])[0]:
            ks = fe["group_idx"][g]
            ks = [int(kf.cam_first[c] + ks[c]) for c in range(self.nc) if ks[c] >= 0]
            if ks and all(kf.mp_ref[i] < 0 for i in ks):
                groups.append((float(fe["group_p3d"][g][2]), int(g), ks))
        groups.sort(key=lambda t: t[0])  # nearest first (stable: group order)
        n_close = sum(1 for z, _, _ in groups if z <= self.th_depth)
        groups = groups[:max(n_close, min(100, len(groups)))]
        if groups:
            n0 = len(self.mp_X)
            P3 = np.array([fe["group_p3d"][g] for _, g, _ in groups])
            Xw = (P3 @ kf.Rwc.T + kf.twc)
            first = np.array([ks[0] for _, _, ks in groups])
            self.mp_X = np.concatenate([self.mp_X, Xw.astype(np.float32)])
            self.mp_desc = np.concatenate([self.mp_desc, kf.desc[first]])
            self.mp_bad = np.concatenate([self.mp_bad, np.zeros(len(groups), bool)])
            d = Xw - kf.twc
            dist = np.linalg.norm(d, axis=1)
            self.mp_normal = np.concatenate([self.mp_normal, (d / dist[:, None]).astype(np.float32)])
            maxd = (dist * self.scale[kf.keys["octave"][first]]).astype(np.float32)
            self.mp_maxd = np.concatenate([self.mp_maxd, maxd])
            self.mp_mind = np.concatenate([self.mp_mind, (maxd / self.scale[NLEVELS - 1]).astype(np.float32)])
            for j, (_, _, ks) in enumerate(groups):
                self.mp_obs.append({})
                for i in ks:
                    kf.mp_ref[i] = n0 + j
                    self._add_obs(n0 + j, kf.id, i)
        return kf

    def _add_obs(self, m, kid, i):
        self.mp_obs[m][kid] = tuple(sorted(set(self.mp_obs[m].get(kid, ()) + (i,))))

    def _update_normal_depth(self, ids):
        ids = [m for m in ids if not self.mp_bad[m] and self.mp_obs[m]]
        if not ids:
            return
        kids = [sorted(self.mp_obs[m]) for m in ids]
        first = np.zeros(len(ids) + 1, np.int32)
        first[1:] = np.cumsum([len(k) for k in kids])
        obs_centre = np.fromiter((k for ks in kids for k in ks), np.int32, int(first[-1]))
        centres = np.array([k.twc for k in self.kfs], np.float32)
        ref = np.array([ks[0] for ks in kids], np.int32)
        ref_scale = np.array([self.scale[self.kfs[ks[0]].keys["octave"][self.mp_obs[m][ks[0]][0]]] for m, ks in zip(ids, kids)],
                             np.float32)
        nrm, mx, mn = self.S.update_normal_depth(self.mp_X[ids], first, obs_centre, centres, ref, ref_scale, self.scale[NLEVELS - 1])
        self.mp_normal[ids], self.mp_maxd[ids], self.mp_mind[ids] = nrm, mx, mn

    # ---- Optimizer::LocalBundleAdjustmentNavStatePRV with per-camera edges (camera of an observation in bits 24..27)
    def local_ba(self, apply=True):
        local = self.kfs[-self.n_local:]
        first = local[0].id
        prev = self.kfs[first - 1] if first > 0 else None
        local_ids = {k.id for k in local}
        pts, seen = [], set()
        for k in local:
            for m in k.mp_ref[k.mp_ref >= 0]:
                m = int(m)
                if m not in seen and not self.mp_bad[m]:
                    seen.add(m)
                    pts.append(m)
        fixed_ids = [prev.id] if prev is not None else []
        for m in pts:
            for kid in sorted(self.mp_obs[m]):
                if kid not in local_ids and kid not in fixed_ids:
                    fixed_ids.append(kid)
        order = [k.id for k in local] + fixed_ids
        index = {kid: i for i, kid in enumerate(order)}
        kfs = np.zeros(len(order), LBA_KEYFRAME_DTYPE)
        for i, kid in enumerate(order):
            kfs[i]["nav"] = self.kfs[kid].nav
            kfs[i]["fixed"] = int(i >= len(local) or kid == 0)
        rows = []
        for j, m in enumerate(pts):
            for kid in sorted(self.mp_obs[m]):
                if kid in index:
                    k = self.kfs[kid]
                    for i in self.mp_obs[m][kid]:
                        rows.append((index[kid] | (int(k.key_cam[i]) << 24), j, k.keys["x"][i], k.keys["y"][i], -1.0,
                                     self.inv_sigma2[k.keys["octave"][i]], kid, i))
        obs = np.zeros(len(rows), LBA_OBS_DTYPE)
        for r, row in enumerate(rows):
            obs[r] = row[:6]
        edges = []
        for k in local:
            if k.id > 0 and (k.id - 1) in index and k.imu_edge is not None:
                e = np.zeros(1, LBA_IMU_EDGE_DTYPE)[0]
                e["kf_i"], e["kf_j"] = index[k.id - 1], index[k.id]
                e["dt_kf"] = k.t - self.kfs[k.id - 1].t
                e["imu"] = k.imu_edge
                edges.append(e)
        imu = np.array(edges, LBA_IMU_EDGE_DTYPE) if edges else np.zeros(0, LBA_IMU_EDGE_DTYPE)
        P = np.zeros(1, LBA_VIO_PARAMS_DTYPE)
        b = P[0]["base"]
        c0 = self.scene.cams[0]
        b["Rcb"], b["tcb"] = self.Tcb[:3, :3].reshape(-1), self.Tcb[:3, 3]
        b["fx"], b["fy"], b["cx"], b["cy"], b["bf"] = c0["fx"], c0["fy"], c0["cx"], c0["cy"], self.fe.bf
        b["its0"], b["its1"] = 4, 6
        b["n_cams"], b["cams"] = self.nc, self.scene.cams.ctypes.data
        P[0]["gw"] = synth_ba.GRAVITY
        P[0]["inv_sigma_bg2"], P[0]["inv_sigma_ba2"] = 1.0 / synth_ba.IMU_SIGMA[2] ** 2, 1.0 / synth_ba.IMU_SIGMA[3] ** 2
        P[0]["lambda_init"] = 1.0
        P[0]["qRbe"][0] = 1.0
        X = self.mp_X[pts]
        t0 = time.perf_counter()
        navs, Xo, erase, res = self.S.lba_vio(P, kfs, X, np.zeros(len(pts), np.uint8), obs, imu)
        self.stats["ms_lba"].append(1e3 * (time.perf_counter() - t0))
        self.stats["lba"] += 1
        self.stats.setdefault("lba_shapes", []).append((len(order), len(fixed_ids) + int(order[0] == 0), len(pts), len(obs)))
        job = dict(local=local, kfs=kfs, pts=pts, rows=rows, navs=navs, Xo=Xo, erase=erase, res=res)
        if apply:
            self._lba_apply(job)
        return job

    def _lba_apply(self, job):
        local, kfs, pts, rows, navs, Xo, erase, res = (job[k] for k in ("local", "kfs", "pts", "rows", "navs", "Xo", "erase", "res"))
        self.stats["lba_applied"] += 1
        if int(res["status"]) != 0:
            return res
        for r in np.nonzero(erase)[0]:  # ErasePairObs: this key's observation of the point
            _, j, _, _, _, _, kid, i = rows[r]
            m = pts[j]
            left = tuple(x for x in self.mp_obs[m].get(kid, ()) if x != i)
            if left:
                self.mp_obs[m][kid] = left
            else:
                self.mp_obs[m].pop(kid, None)
            self.kfs[kid].mp_ref[i] = -1
            if not self.mp_obs[m]:
                self.mp_bad[m] = True
        for i, k in enumerate(local):
            if not kfs[i]["fixed"]:
                k.nav = navs[i].copy()
                _, k.Rwc, k.twc = replay._Tcw_of(k.nav, self.Tbc)
        self.mp_X[pts] = Xo
        self._update_normal_depth(pts)
        return res

    # ---- the glue of a frame
    def _last_points(self, last):
        has = (last.mp_ref >= 0) & ~last.outlier
        has[has] &= ~self.mp_bad[last.mp_ref[has]]
        pts = np.zeros(last.N, LAST_FRAME_POINT_DTYPE)
        pts["Xw"][has] = self.mp_X[last.mp_ref[has]]
        pts["octave"], pts["angle"] = last.keys["octave"], last.keys["angle"]
        pts["flags"] = has.astype(np.int32) * 3
        pts["desc"][has] = self.mp_desc[last.mp_ref[has]]
        return pts, has

    def _obs(self, f):
        idx = np.nonzero(f.mp_ref >= 0)[0]
        obs = np.zeros(len(idx), POSE_OBS_DTYPE)
        obs["Xw"] = self.mp_X[f.mp_ref[idx]]
        obs["u"], obs["v"], obs["ur"] = f.keys["x"][idx], f.keys["y"][idx], -1.0
        obs["inv_sigma2"] = self.inv_sigma2[f.keys["octave"][idx]]
        obs["flags"] = (f.key_cam[idx] << 8) | (f.track_depth[idx] < np.float32(max(10.0, self.th_depth))).astype(np.int32)
        return obs, idx

    def _vio_frame(self, nav, ref_nav, im, prior, dt_frames, marg):
        F = np.zeros(1, VIO_FRAME_DTYPE)
        f = F[0]
        b = f["base"]
        c0 = self.scene.cams[0]
        b["nav"] = nav
        b["Rcb"], b["tcb"] = self.Tcb[:3, :3].reshape(-1), self.Tcb[:3, 3]
        b["fx"], b["fy"], b["cx"], b["cy"], b["bf"] = c0["fx"], c0["fy"], c0["cx"], c0["cy"], self.fe.bf
        b["n_cams"], b["cams"] = self.nc, self.scene.cams.ctypes.data
        f["nav_last"], f["imu"], f["gw"] = ref_nav, im, synth_ba.GRAVITY
        f["inv_sigma_bg2"], f["inv_sigma_ba2"] = 1.0 / synth_ba.IMU_SIGMA[2] ** 2, 1.0 / synth_ba.IMU_SIGMA[3] ** 2
        f["dt_frames"], f["th_depth"], f["compute_marg"] = dt_frames, self.th_depth, int(marg)
        if prior is not None:
            f["nav_prior"], f["H_prior"], f["last_has_prior"] = prior[0], prior[1], 1
        return F

    def _frustum_frame(self, Tcw):
        F = np.zeros(1, FRUSTUM_FRAME_DTYPE)
        f, R = F[0], self.fe.rig[0]
        f["Rcrw"], f["tcrw"], f["Ow"] = Tcw[:, :3].reshape(-1), Tcw[:, 3], -Tcw[:, :3].T @ Tcw[:, 3]
        f["n_cams"], f["use_distort"], f["cams"] = self.nc, 1, self.scene.cams.ctypes.data
        for c in range(self.nc):
            f["Tcr"][c], f["trc"][c], f["bounds"][c] = R["Tcr"][c], R["trc"][c], R["bounds"][c]
        f["bf"], f["n_levels"], f["viewing_cos_limit"] = self.fe.bf, NLEVELS, 0.5
        f["log_scale_factor"] = np.float32(np.log(np.float32(1.2)))
        return F

    def _frustum_points(self, ids):
        P = np.zeros(len(ids), FRUSTUM_POINT_DTYPE)
        P["Xw"], P["normal"] = self.mp_X[ids], self.mp_normal[ids]
        P["max_distance"], P["min_distance"] = self.mp_maxd[ids], self.mp_mind[ids]
        return P

    def _sbp_cam(self, nav, Tcw_last, th):
        Tcw, _, _ = replay._Tcw_of(nav, self.Tbc)
        return frontend.make_sbp_camera(Tcw, Tcw_last, self.fe.K0, self.fe.bounds[0], self.fe.bf, self.fe.bf / self.fe.K0[0], th,
                                        self.scale), Tcw

    def initialise(self):
        super().initialise()
        self.last.outlier = np.zeros(self.last.N, bool)

    # ---- one frame, member by member (Tracking::TrackWithIMU + TrackLocalMapWithIMU with the camera loops)
    def step(self, k):
        t0 = time.perf_counter()
        S, last, nc = self.S, self.last, self.nc
        f = self.make_frame(k)
        ref_nav = self.kfs[-1].nav if self.map_updated else last.nav
        prior = None if self.map_updated else last.prior
        t_ref = self.kfs[-1].t if self.map_updated else last.t
        im, prv, st = S.preintegrate(self.seq.noise, self.seq.imu_between(t_ref, f.t), t_ref, f.t, ref_nav["bg"], ref_nav["ba"])
        assert st == 0, "IMU pre-integration failed"
        nav_pred = self.predict(ref_nav, im)
        Tcw_last, _, _ = replay._Tcw_of(last.nav, self.Tbc)
        pts, has = self._last_points(last)
        cam, _ = self._sbp_cam(nav_pred, Tcw_last, self.th_last)
        bounds = self.fe.bounds
        q1 = S.project_last_frame(pts, cam, self.fe.rig)
        n1, a1 = S.search(0, q1, f.keys, f.uright, f.desc, None, bounds, f.cam_first, 0.9)
        if n1 < 20:
            cam[0]["th"] = 2 * self.th_last
            q1 = S.project_last_frame(pts, cam, self.fe.rig)
            n1, a1 = S.search(0, q1, f.keys, f.uright, f.desc, None, bounds, f.cam_first, 0.9)
        ok = a1 >= 0
        f.mp_ref[ok] = last.mp_ref[a1[ok] // nc]  # query (i, camj) belongs to last-frame key i
        f.track_depth[ok] = last.track_depth[a1[ok] // nc]
        obs1, idx1 = self._obs(f)
        F1 = self._vio_frame(nav_pred, ref_nav, im, prior, f.t - t_ref, False)
        F1[0]["base"]["n_obs"] = len(obs1)
        r1, o1 = S.pose_vio(F1, obs1)
        f.mp_ref[idx1[o1 != 0]] = -1
        nav1 = r1["base"]["nav"] if int(r1["base"]["status"]) == 0 else nav_pred
        Tcw1, _, _ = replay._Tcw_of(nav1, self.Tbc)
        cand = self._local_points(f)
        n2 = 0
        if len(cand):
            info = S.in_frustum(self._frustum_frame(Tcw1), self._frustum_points(cand))
            q2, owner = frontend.queries_from_track_info(info, self.mp_desc[cand], self.th_local, self.scale)
            if len(q2):
                taken = (f.mp_ref >= 0).astype(np.uint8)
                n2, a2 = S.search(1, q2, f.keys, f.uright, f.desc, taken, bounds, f.cam_first, 0.8)
                ok = a2 >= 0
                f.mp_ref[ok] = cand[owner[a2[ok]]]
                f.track_depth[ok] = info["track_depth"][owner[a2[ok]]]
        obs2, idx2 = self._obs(f)
        F2 = self._vio_frame(nav1, ref_nav, im, prior, f.t - t_ref, True)
        F2[0]["base"]["n_obs"] = len(obs2)
        r2, o2 = S.pose_vio(F2, obs2)
        f.outlier = np.zeros(f.N, bool)
        f.outlier[idx2[o2 != 0]] = True
        f.nav = (r2["base"]["nav"] if int(r2["base"]["status"]) == 0 else nav1).copy()
        f.prior = (f.nav.copy(), r2["H_marg"].copy()) if int(r2["has_marg"]) else None
        self.map_updated = False
        self.stats["n_matches"].append((int(n1), int(n2)))
        self.stats["n_inliers"].append(int(r2["base"]["n_inliers"]))
        # (LM iteration counts of the two optimisations: near convergence the sign of a gain ratio is rounding noise, and a
        # trial more or less moves the result by its last step -- the float divergence tools/rig_drift.py traces)
        self.stats.setdefault("lm_iterations", []).append((int(r1["base"]["lm_iterations"]), int(r2["base"]["lm_iterations"])))
        # (which key holds which point, which observations were kept: the counts above can agree while the sets differ)
        self.stats.setdefault("assignment_crc", []).append((zlib.crc32(np.ascontiguousarray(f.mp_ref).tobytes()), zlib.crc32(np.ascontiguousarray(f.outlier).tobytes())))
        if self.stats.get("keep_marg"):  # (diagnosis only: tools/rig_drift.py)
            self.stats.setdefault("H_marg", []).append(r2["H_marg"].copy() if int(r2["has_marg"]) else None)
        return self._finish_frame(k, f, t0)


class RigTrackerReplay(RigReplay):
    """every frame's tracking as ONE vieo_track_frame call on a rig tracker (vieo_tracker_create_rig)"""

    def __init__(self, seq, stages, nfeatures=1200, max_local_points=16384, prefetch=False, **kw):
        from .tracker import Tracker, rig_params
        super().__init__(seq, stages, nfeatures, **kw)
        prm, rg = rig_params(self.scene, nfeatures, max_local_points=max_local_points, th_last=self.th_last, th_local=self.th_local,
                             noise=seq.noise[0], th_depth=self.th_depth)
        self.trk = Tracker(prm, rg)
        self._lv = 0
        # frame pipelining (vieo_track_input.next_images / next_imu): frame k + 1 goes along with frame k's call
        self.prefetch, self._prefetched, self._n_run = bool(prefetch), False, 0
        self.stats["ms_chain"] = []
        self.stats["widened"] = 0

    def close(self):
        self.trk.close()

    def run(self, n_frames=None):
        self._n_run = n_frames or self.seq.n_frames
        return super().run(n_frames)

    def _pipelining(self, k, t):
        """(use_prefetched, next_images, next_imu) of frame k's call"""
        use = self.prefetch and self._prefetched
        nxt = self.seq.images(k + 1) if (self.prefetch and k + 1 < self._n_run) else None
        self._prefetched = nxt is not None
        nxt_imu = None
        if nxt is not None and len(self.seq.imu):
            t_next = self.seq.time(k + 1)
            nxt_imu = (self.seq.imu_between(t, t_next), t_next)
        return use, nxt, nxt_imu

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
            self._lp_pts, self._lp_desc = self._frustum_points(self._lp), self.mp_desc[self._lp].copy()
        return self._lp

    def step(self, k):
        t0 = time.perf_counter()
        last = self.last
        ref_nav = self.kfs[-1].nav if self.map_updated else last.nav
        prior = None if self.map_updated else last.prior
        t_ref = self.kfs[-1].t if self.map_updated else last.t
        t = self.seq.time(k)
        pts, has = self._last_points(last)
        # the keys of one stereo group hold the same MapPoint: they name the first of them as their table entry
        lk = np.nonzero(has)[0]
        where = np.full(len(self.mp_X), -1, np.int32)
        where[last.mp_ref[lk[::-1]]] = lk[::-1]  # (the lowest key index wins)
        pts["reserved"][lk, 0] = where[last.mp_ref[lk]] + 1
        cand = self._all_local_points()
        alias = where[cand] if len(cand) else np.zeros(0, np.int32)
        use_pf, nxt, nxt_imu = self._pipelining(k, t)
        o, v = self.trk.track(None, None, self.seq.imu_between(t_ref, t), t_ref, t, ref_nav, last.nav, prior, pts, last.track_depth,
                              self._lp_pts, self._lp_desc, alias, self._lv, images=self.seq.images(k), next_images=nxt,
                              use_prefetched=use_pf, next_imu=nxt_imu)
        assert int(o["status"]) == 0 and int(o["stereo_status"]) == 0, "tracking call failed"
        self.stats["ms_chain"].append((float(o["ms_host"]), float(o["ms_gpu"])))
        self.stats["widened"] += int(o["widened"])
        cap, nc = int(o["key_cap"]), self.nc
        f = replay._Frame()
        f.k, f.t = k, t
        N = f.N = int(o["n_keys"])
        f.keys, f.desc = v["keys"].copy(), v["desc"].copy()
        f.uright, f.depth = v["uright"].copy(), v["depth"].copy()
        f.cam_first = np.array(o["cam_first"][:nc + 1], np.int32)
        f.key_cam = (np.searchsorted(f.cam_first, np.arange(N), side="right") - 1).astype(np.int32)
        f.key_group = v["key_group"].copy()

        class _RF:  # what insert_keyframe reads of the staged frame
            pass
        f.rf = _RF()
        f.rf.fe = dict(group_good=v["group_good"].copy(), group_idx=v["group_idx"].copy(), group_p3d=v["group_p3d"].copy(),
                       key_group=f.key_group)
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
        f.nav = (r2["base"]["nav"] if int(r2["base"]["status"]) == 0 else r1["base"]["nav"]).copy()
        f.prior = (f.nav.copy(), r2["H_marg"].copy()) if int(r2["has_marg"]) else None
        self.map_updated = False
        self.stats["n_matches"].append((int(o["n_matches_last"]), int(o["n_matches_local"])))
        self.stats["n_inliers"].append(int(r2["base"]["n_inliers"]))
        # (LM iteration counts of the two optimisations: near convergence the sign of a gain ratio is rounding noise, and a
        # trial more or less moves the result by its last step -- the float divergence tools/rig_drift.py traces)
        self.stats.setdefault("lm_iterations", []).append((int(r1["base"]["lm_iterations"]), int(r2["base"]["lm_iterations"])))
        # (which key holds which point, which observations were kept: the counts above can agree while the sets differ)
        self.stats.setdefault("assignment_crc", []).append((zlib.crc32(np.ascontiguousarray(f.mp_ref).tobytes()), zlib.crc32(np.ascontiguousarray(f.outlier).tobytes())))
        if self.stats.get("keep_marg"):  # (diagnosis only: tools/rig_drift.py)
            self.stats.setdefault("H_marg", []).append(r2["H_marg"].copy() if int(r2["has_marg"]) else None)
        return self._finish_frame(k, f, t0)


# ================================================================ rectified stereo without IMU
NFEAT_VISION = 1000


class HipVisionStages(replay.HipStages):
    name = "hip"

    def __init__(self, resident=False):
        from .matching import ORBmatcher
        from .orb_extractor import ORBextractor
        self.extL = ORBextractor(NFEAT_VISION, replay.SCALE, NLEVELS, replay.INI_TH, replay.MIN_TH)
        self.extR = ORBextractor(NFEAT_VISION, replay.SCALE, NLEVELS, replay.INI_TH, replay.MIN_TH)
        self.M = ORBmatcher
        self.resident, self.resident_calls = bool(resident), 0

    def pose(self, F, obs):
        from .optimizer import Optimizer
        return Optimizer.PoseOptimization(F, obs)

    def lba(self, params, kfs, pts, obs):
        from .optimizer import Optimizer
        return Optimizer.LocalBundleAdjustment(params, kfs, pts, obs)


def _T_of(nav):
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = synth_ba.quat_to_R(nav["q"]), nav["p"]
    return T


def _nav_of(T, like):
    n = like.copy()
    n["p"], n["q"] = T[:3, 3], synth_ba._R_to_quat(T[:3, :3])
    return n


class VisionReplay(replay.Replay):
    """configs[0].  The body pose plays the NavState's part (p, q; v and the biases are carried along untouched): the
    vision-only PoseOptimization takes the same vieo_pose_frame."""

    def __init__(self, seq, stages, th_local=1.0, **kw):
        super().__init__(seq, stages, th_local=th_local, **kw)
        self.velocity = None  # T_b(k-1)<-b(k): the constant-velocity model in the body frame

    def make_frame(self, k):
        f = super().make_frame(k)
        return f

    def _anchor(self, f):
        """Tracking::UpdateLastFrame (src/Tracking.cc:1790-1841): the last frame's pose follows its reference key frame"""
        kf = self.kfs[f.ref_kf]
        return _T_of(kf.nav) @ f.T_rel

    def _set_ref(self, f):
        f.ref_kf = len(self.kfs) - 1
        f.T_rel = np.linalg.inv(_T_of(self.kfs[f.ref_kf].nav)) @ _T_of(f.nav)

    def initialise(self):
        super().initialise()
        self._set_ref(self.last)
        self.velocity = np.eye(4)

    def _pose_frame(self, nav, n_obs):
        F = np.zeros(1, POSE_FRAME_DTYPE)
        b = F[0]
        b["nav"] = nav
        b["Rcb"], b["tcb"] = self.Tcb[:3, :3].reshape(-1), self.Tcb[:3, 3]
        b["fx"], b["fy"], b["cx"], b["cy"], b["bf"] = sc.FX, sc.FY, sc.CX, sc.CY, sc.BF
        b["n_obs"] = n_obs
        return F

    def _obs(self, f):
        obs, idx = super()._obs(f)
        obs["flags"] = 0  # (the vision-only optimisation has no close-point gate)
        return obs, idx

    def _predict(self, last):
        Twb_last = self._anchor(last)
        nav_last = _nav_of(Twb_last, last.nav)
        return nav_last, _nav_of(Twb_last @ self.velocity, last.nav)

    def step(self, k):
        t0 = time.perf_counter()
        S, last = self.S, self.last
        f = self.make_frame(k)
        nav_last, nav_pred = self._predict(last)
        Tcw, _, _ = replay._Tcw_of(nav_pred, self.Tbc)
        Tcw_last, _, _ = replay._Tcw_of(nav_last, self.Tbc)
        has = (last.mp_ref >= 0) & ~last.outlier
        has[has] &= ~self.mp_bad[last.mp_ref[has]]
        Xw = np.zeros((last.N, 3), np.float32)
        Xw[has] = self.mp_X[last.mp_ref[has]]
        pts = frontend.make_last_frame_points(last.keys, np.zeros((last.N, 32), np.uint8), Xw, has, True)
        pts["desc"][has] = self.mp_desc[last.mp_ref[has]]
        cam = frontend.make_sbp_camera(Tcw, Tcw_last, replay.K, replay.BOUNDS, sc.BF, sc.BASELINE, self.th_last, self.scale)
        n1, a1 = S.search(0, S.project_last_frame(pts, cam), f.keys, f.uright, f.desc, None, 0.9)
        if n1 < 20:
            cam[0]["th"] = 2 * self.th_last
            n1, a1 = S.search(0, S.project_last_frame(pts, cam), f.keys, f.uright, f.desc, None, 0.9)
        assert n1 >= 20, "TrackWithMotionModel lost the frame"
        ok = a1 >= 0
        f.mp_ref[ok] = last.mp_ref[a1[ok]]
        f.track_depth[ok] = last.track_depth[a1[ok]]
        obs1, idx1 = self._obs(f)
        r1, o1 = S.pose(self._pose_frame(nav_pred, len(obs1)), obs1)
        f.mp_ref[idx1[o1 != 0]] = -1
        nav1 = r1["nav"] if int(r1["status"]) == 0 else nav_pred
        Tcw1, _, _ = replay._Tcw_of(nav1, self.Tbc)
        cand = self._local_points(f)
        n2 = 0
        if len(cand):
            FF = np.zeros(1, FRUSTUM_FRAME_DTYPE)
            ff = FF[0]
            ff["Rcrw"], ff["tcrw"], ff["Ow"] = Tcw1[:, :3].reshape(-1), Tcw1[:, 3], -Tcw1[:, :3].T @ Tcw1[:, 3]
            ff["n_cams"], ff["use_distort"], ff["cams"] = 1, 0, self._pinhole().ctypes.data
            ff["Tcr"][0] = np.eye(4)[:3].reshape(-1)
            ff["bounds"][0] = replay.BOUNDS
            ff["bf"], ff["n_levels"], ff["viewing_cos_limit"] = sc.BF, NLEVELS, 0.5
            ff["log_scale_factor"] = np.float32(np.log(np.float32(replay.SCALE)))
            P = np.zeros(len(cand), FRUSTUM_POINT_DTYPE)
            P["Xw"], P["normal"] = self.mp_X[cand], self.mp_normal[cand]
            P["max_distance"], P["min_distance"] = self.mp_maxd[cand], self.mp_mind[cand]
            info = S.in_frustum(FF, P)
            q2, owner = frontend.queries_from_track_info(info, self.mp_desc[cand], self.th_local, self.scale)
            if len(q2):
                taken = (f.mp_ref >= 0).astype(np.uint8)
                n2, a2 = S.search(1, q2, f.keys, f.uright, f.desc, taken, 0.8)
                ok = a2 >= 0
                f.mp_ref[ok] = cand[owner[a2[ok]]]
                f.track_depth[ok] = info["track_depth"][owner[a2[ok]]]
        obs2, idx2 = self._obs(f)
        r2, o2 = S.pose(self._pose_frame(nav1, len(obs2)), obs2)
        f.outlier = np.zeros(f.N, bool)
        f.outlier[idx2[o2 != 0]] = True
        f.nav = (r2["nav"] if int(r2["status"]) == 0 else nav1).copy()
        f.prior = None
        self.stats["n_matches"].append((int(n1), int(n2)))
        self.stats["n_inliers"].append(int(r2["n_inliers"]))
        return self._after_tracking(k, f, nav_last, t0)

    def _after_tracking(self, k, f, nav_last, t0):
        # mVelocity = Tcw_cur * Twc_last (src/Tracking.cc:1178-1189), here between the body poses
        self.velocity = np.linalg.inv(_T_of(nav_last)) @ _T_of(f.nav)
        self.map_updated = False
        self._set_ref(f)
        return self._finish_frame(k, f, t0)

    # ---- key frames and the vision-only local BA
    def _finish_frame(self, k, f, t0):
        if k % self.kf_every == 0:
            f.mp_ref[f.outlier] = -1
            kf = self.insert_keyframe(f, f.nav, None)
            self.stats["ms_frames"].append(1e3 * (time.perf_counter() - t0))
            if self.lba_lag <= 0:
                self.local_ba()
                f.nav = kf.nav.copy()
            else:
                self.before_frame(k + 1)
                job = self.local_ba(apply=False)
                self._pending = (k + self.lba_lag, job)
            f.outlier = np.zeros(f.N, bool)
            self._set_ref(f)  # the key frame is its own reference: T_rel = identity
        else:
            self.stats["ms_frames"].append(1e3 * (time.perf_counter() - t0))
        self.last = f
        self.traj.append(f.nav.copy())
        self.stats["frames"] += 1
        return f.nav

    def local_ba(self, apply=True):
        local = self.kfs[-self.n_local:]
        local_ids = {k.id for k in local}
        pts, seen = [], set()
        for k in local:
            for m in k.mp_ref[k.mp_ref >= 0]:
                m = int(m)
                if m not in seen and not self.mp_bad[m]:
                    seen.add(m)
                    pts.append(m)
        fixed_ids = []
        for m in pts:
            for kid in sorted(self.mp_obs[m]):
                if kid not in local_ids and kid not in fixed_ids:
                    fixed_ids.append(kid)
        order = [k.id for k in local] + fixed_ids
        index = {kid: i for i, kid in enumerate(order)}
        kfs = np.zeros(len(order), LBA_KEYFRAME_DTYPE)
        for i, kid in enumerate(order):
            kfs[i]["nav"] = self.kfs[kid].nav
            kfs[i]["fixed"] = int(i >= len(local) or kid == 0)
        rows = []
        for j, m in enumerate(pts):
            for kid in sorted(self.mp_obs[m]):
                if kid in index:
                    kk = self.kfs[kid]
                    i = self.mp_obs[m][kid]
                    rows.append((index[kid], j, kk.keys["x"][i], kk.keys["y"][i], kk.uright[i], self.inv_sigma2[kk.keys["octave"][i]],
                                 kid, i))
        obs = np.zeros(len(rows), LBA_OBS_DTYPE)
        for r, row in enumerate(rows):
            obs[r] = row[:6]
        P = np.zeros(1, LBA_PARAMS_DTYPE)
        b = P[0]
        b["Rcb"], b["tcb"] = self.Tcb[:3, :3].reshape(-1), self.Tcb[:3, 3]
        b["fx"], b["fy"], b["cx"], b["cy"], b["bf"] = sc.FX, sc.FY, sc.CX, sc.CY, sc.BF
        b["its0"], b["its1"] = 5, 10  # src/Optimizer.cc:2179-2258
        t0 = time.perf_counter()
        navs, Xo, erase, res = self.S.lba(P, kfs, self.mp_X[pts], obs)
        self.stats["ms_lba"].append(1e3 * (time.perf_counter() - t0))
        self.stats["lba"] += 1
        self.stats.setdefault("lba_shapes", []).append((len(order), len(fixed_ids) + int(order[0] == 0), len(pts), len(obs)))
        job = dict(local=local, kfs=kfs, pts=pts, rows=rows, navs=navs, Xo=Xo, erase=erase, res=res)
        if apply:
            self._lba_apply(job)
        return job


class VisionTrackerReplay(VisionReplay):
    """every frame's tracking as ONE vieo_track_frame call on a vision-only tracker (params.vision_only = 1)"""

    def __init__(self, seq, stages, max_local_points=16384, prefetch=False, **kw):
        from .tracker import Tracker, euroc_params
        super().__init__(seq, stages, **kw)
        prm = euroc_params(max_local_points, self.th_last, self.th_local, seq.noise[0])
        prm[0]["n_features"], prm[0]["vision_only"] = NFEAT_VISION, 1
        self.trk = Tracker(prm)
        self._lv = 0
        self.prefetch, self._prefetched, self._n_run = bool(prefetch), False, 0
        self.stats["ms_chain"] = []

    def run(self, n_frames=None):
        self._n_run = n_frames or self.seq.n_frames
        return super().run(n_frames)

    def close(self):
        self.trk.close()

    _all_local_points = RigTrackerReplay._all_local_points

    def _frustum_points(self, ids):
        P = np.zeros(len(ids), FRUSTUM_POINT_DTYPE)
        P["Xw"], P["normal"] = self.mp_X[ids], self.mp_normal[ids]
        P["max_distance"], P["min_distance"] = self.mp_maxd[ids], self.mp_mind[ids]
        return P

    def step(self, k):
        t0 = time.perf_counter()
        last = self.last
        nav_last, nav_pred = self._predict(last)
        Li, Ri = self.seq.images(k)
        has = (last.mp_ref >= 0) & ~last.outlier
        has[has] &= ~self.mp_bad[last.mp_ref[has]]
        Xw = np.zeros((last.N, 3), np.float32)
        Xw[has] = self.mp_X[last.mp_ref[has]]
        pts = frontend.make_last_frame_points(last.keys, np.zeros((last.N, 32), np.uint8), Xw, has, True)
        pts["desc"][has] = self.mp_desc[last.mp_ref[has]]
        cand = self._all_local_points()
        where = np.full(len(self.mp_X), -1, np.int32)
        lk = np.nonzero(has)[0]
        where[last.mp_ref[lk[::-1]]] = lk[::-1]
        alias = where[cand] if len(cand) else np.zeros(0, np.int32)
        t = self.seq.time(k)
        use_pf = self.prefetch and self._prefetched
        nxt = self.seq.images(k + 1) if (self.prefetch and k + 1 < self._n_run) else None
        self._prefetched = nxt is not None
        o, v = self.trk.track(Li, Ri, np.zeros(0, self.seq.imu.dtype), last.t, t, nav_pred, nav_last, None, pts, last.track_depth,
                              self._lp_pts, self._lp_desc, alias, self._lv, next_images=nxt, use_prefetched=use_pf)
        assert int(o["status"]) == 0, "TrackWithMotionModel lost the frame"
        self.stats["ms_chain"].append((float(o["ms_host"]), float(o["ms_gpu"])))
        cap = int(o["key_cap"])
        f = replay._Frame()
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
        r1, r2 = o["first"]["base"], o["second"]["base"]
        f.nav = (r2["nav"] if int(r2["status"]) == 0 else r1["nav"]).copy()
        f.prior = None
        self.stats["n_matches"].append((int(o["n_matches_last"]), int(o["n_matches_local"])))
        self.stats["n_inliers"].append(int(r2["n_inliers"]))
        return self._after_tracking(k, f, nav_last, t0)
