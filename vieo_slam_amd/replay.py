"""Sequential stereo-inertial replay: ONE stream of frames in which frame t's pose, map points and marginal prior feed
frame t+1 and a local bundle adjustment every few frames writes key-frame states and points back -- the call pattern of
the reference's harness and threads run single-threaded (Examples/Stereo/stereo_euroc.cc:235-307 -> System::TrackStereo
-> Tracking::Track: src/Tracking.cc:261-378 TrackWithIMU, :385-451 PredictNavStateByIMU, :453-488 TrackLocalMapWithIMU,
:2308-2370 SearchLocalPoints; LocalMapping::Run -> Optimizer::LocalBundleAdjustmentNavStatePRV, LocalMapping.cc:113-139).

The driver owns the glue the reference's Frame / KeyFrame / MapPoint / Map objects provide (numpy, float32 where the
reference stores float32) and calls every hot-path function through a `stages` object: `HipStages` = the C-ABI of
libvieo_hot.so; the tests pass the CPU oracle with the same interface and compare the two trajectories (BASELINE
configs[2]: "ATE within 1e-4 of ref").  Simplifications against the full system, identical on both sides: a new key frame
every `kf_every` frames (with IMU the reference inserts one every <= 0.5 s when LocalMapping is idle, Tracking.cc:2085-
2101), new map points only from stereo depth (CreateNewKeyFrame, Tracking.cc:2180-2250; no triangulation against
neighbours, no fusing, no culling), the local map = the points of the last `n_local_kfs` key frames, no relocalisation /
loop closing, IMU initialised (gravity and the starting bias known)."""
import time

import numpy as np

from . import frontend, synth, synth_ba
from . import synth_scene as sc
from .ba_types import (IMU_PREINT_DTYPE, LAST_FRAME_POINT_DTYPE, LBA_IMU_EDGE_DTYPE, LBA_KEYFRAME_DTYPE, LBA_OBS_DTYPE,
                       LBA_VIO_PARAMS_DTYPE, NAVSTATE_DTYPE, POSE_OBS_DTYPE, VIO_FRAME_DTYPE)
from .imu import IMU_NOISE_DTYPE, IMU_SAMPLE_DTYPE
from .map_point import FRUSTUM_FRAME_DTYPE, FRUSTUM_POINT_DTYPE

W, H = sc.W, sc.H
K = (sc.FX, sc.FY, sc.CX, sc.CY)
BOUNDS = np.array([0, W, 0, H], np.float32)
NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH = 1200, 1.2, 8, 20, 7
TH_DEPTH = 35.0 * sc.BASELINE  # ThDepth: 35 (EuRoC_VIO.yaml) x baseline


# ---------------------------------------------------------------- the synthetic sequence
class Sequence:
    """A smooth analytic trajectory above the textured plane: body position = sums of sinusoids, attitude = the
    look-down rotation times Exp(theta(t)); IMU samples at 200 Hz from the analytic angular rate / acceleration plus
    bias and noise (EuRoC sigmas); stereo frames rendered through the rectified pinhole model at 20 Hz."""

    def __init__(self, seed, n_frames, dt=0.05):
        self.seed, self.n_frames, self.dt = seed, n_frames, dt
        rng = np.random.default_rng(seed + 99)
        self.scene = sc.Scene(seed)
        Rwb0, pwb0 = sc.look_down_pose(np.random.default_rng(seed + 5))
        pwb0 = pwb0 - np.array([0, 0, 1.8])  # 2.2 .. 3.2 m above the plane: most stereo points are close ones
        self.R0, self.c = Rwb0, pwb0
        self.A = np.array([0.9, 0.7, 0.25]) * rng.uniform(0.8, 1.2, 3)
        self.wp = np.array([0.55, 0.41, 0.33]) * rng.uniform(0.9, 1.1, 3)
        self.ph = rng.uniform(0, 2 * np.pi, 3)
        self.B = np.array([0.06, 0.05, 0.12]) * rng.uniform(0.8, 1.2, 3)
        self.wr = np.array([0.5, 0.37, 0.29]) * rng.uniform(0.9, 1.1, 3)
        self.pr = rng.uniform(0, 2 * np.pi, 3)
        self.bg = rng.normal(0, 0.004, 3)
        self.ba = rng.normal(0, 0.02, 3)
        self.t0 = 100.0
        h = 1.0 / synth_ba.IMU_FREQ
        n_imu = int(round(n_frames * dt / h)) + 40
        ts = self.t0 - 10 * h + np.arange(n_imu) * h
        S = np.zeros(n_imu, IMU_SAMPLE_DTYPE)
        sg = synth_ba.IMU_SIGMA[0] * np.sqrt(synth_ba.IMU_FREQ)
        sa = synth_ba.IMU_SIGMA[1] * np.sqrt(synth_ba.IMU_FREQ)
        for k, t in enumerate(ts):
            R, p, v, a, om = self.state(t)
            S[k]["t"] = t
            S[k]["w"] = om + self.bg + rng.normal(0, sg, 3)
            S[k]["a"] = R.T @ (a - synth_ba.GRAVITY) + self.ba + rng.normal(0, sa, 3)
        self.imu = S
        self.noise = np.zeros(1, IMU_NOISE_DTYPE)
        self.noise[0]["sigma_g"] = (np.eye(3) * synth_ba.IMU_SIGMA[0] ** 2 * synth_ba.IMU_FREQ).reshape(-1)
        self.noise[0]["sigma_a"] = (np.eye(3) * synth_ba.IMU_SIGMA[1] ** 2 * synth_ba.IMU_FREQ).reshape(-1)
        self.noise[0]["freq_ref"], self.noise[0]["dt_cov_noise_fixed"] = synth_ba.IMU_FREQ, 1
        self._img = {}

    def time(self, k):
        return self.t0 + k * self.dt

    def state(self, t):
        """(Rwb, pwb, vwb, world acceleration, body angular rate) at time t"""
        s = t - self.t0
        arg = self.wp * s + self.ph
        p = self.c + self.A * (np.sin(arg) - np.sin(self.ph))
        v = self.A * self.wp * np.cos(arg)
        a = -self.A * self.wp ** 2 * np.sin(arg)
        ar = self.wr * s + self.pr
        th = self.B * (np.sin(ar) - np.sin(self.pr))
        thd = self.B * self.wr * np.cos(ar)
        R = self.R0 @ synth_ba.so3_exp(th)
        om = synth_ba.so3_Jr(th) @ thd  # body rate of R0 Exp(theta(t))
        return R, p, v, a, om

    def truth(self, k):
        R, p, v, _, _ = self.state(self.time(k))
        return dict(p=p, q=synth_ba._R_to_quat(R), v=v)

    def images(self, k):
        if k not in self._img:
            R, p, _, _, _ = self.state(self.time(k))
            L, Rr, _, _, _ = self.scene.stereo(R, p, 1000 * self.seed + 10 * k)
            self._img[k] = (L, Rr)
        return self._img[k]

    def imu_between(self, ti, tj):
        """the samples the reference hands to PreIntegration for [ti, tj]: from the last one at or before ti to the first
        one at or after tj (the ends are interpolated there, OdomPreIntegrator.h:226-330)"""
        t = self.imu["t"]
        a = max(int(np.searchsorted(t, ti, side="right")) - 1, 0)
        b = min(int(np.searchsorted(t, tj, side="left")), len(t) - 1)
        return self.imu[a:b + 1]


# ---------------------------------------------------------------- the hot-path calls (C-ABI)
class HipStages:
    """Every call of the replay that belongs to the hot path, through libvieo_hot.so (host-pointer entry points).
    resident=True: the frame's stereo match and its two projection searches read the keys / descriptors / pyramid where
    the extractions left them (include/vieo_hot.h "the resident frame": the drop-in shims' fast form) whenever the
    handles still hold the frame they are given -- same results, less traffic."""
    name = "hip"

    def __init__(self, resident=False):
        from .matching import ORBmatcher
        from .orb_extractor import ORBextractor
        self.extL = ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)
        self.extR = ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)
        self.M = ORBmatcher
        self.resident = bool(resident)
        self.resident_calls = 0

    def scale_factors(self):
        return self.extL.GetScaleFactors()

    def extract(self, cam, image):
        return (self.extL if cam == 0 else self.extR)(image)

    def stereo(self, kl, dl, kr, dr):
        from .matching import compute_stereo_matches, compute_stereo_matches_resident
        if self.resident and self.extL.holds(kl) and self.extR.holds(kr):
            self.resident_calls += 1
            return compute_stereo_matches_resident(self.extL, self.extR, sc.BASELINE, sc.BF)
        return compute_stereo_matches(self.extL, self.extR, kl, dl, kr, dr, sc.BASELINE, sc.BF)

    def search_last_frame(self, pts, cam, keys, ur, desc, nn):
        """SearchByProjection(Frame&, const Frame&, ...): projection + search -> (queries or None, nmatches, assign)"""
        if self.resident and self.extL.holds(keys):
            self.resident_calls += 1
            n, a = self.M(nn, True).search_last_frame_resident(self.extL, pts, cam)
            return n, a
        q = self.project_last_frame(pts, cam)
        return self.search(0, q, keys, ur, desc, None, nn)

    def preintegrate(self, noise, samples, ti, tj, bg, ba):
        from .imu import imu_preintegrate
        out, prv, st = imu_preintegrate(noise, [samples], [ti], [tj], [bg], [ba])
        return out[0], prv[0], int(st[0])

    def project_last_frame(self, pts, cam):
        return self.M.project_last_frame(pts, cam)

    def search(self, mode, q, keys, ur, desc, taken, nn):
        if self.resident and self.extL.holds(keys):
            self.resident_calls += 1
            return self.M(nn, True).search_resident(mode, self.extL, q, taken, BOUNDS)
        return self.M(nn, True)._search(mode, q, keys, ur, desc, taken, BOUNDS)

    def pose_vio(self, F, obs):
        from .optimizer import Optimizer
        return Optimizer.PoseOptimizationVIO(F, obs)

    def in_frustum(self, F, P):
        from .map_point import is_in_frustum
        return is_in_frustum(F, P)

    def lba_vio(self, params, kfs, pts, close, obs, imu):
        from .optimizer import Optimizer
        return Optimizer.LocalBundleAdjustmentNavStatePRV(params, kfs, pts, close, obs, imu)

    def update_normal_depth(self, points, first, obs_centre, centres, ref_centre, ref_scale, scale_last):
        from .map_point import update_normal_and_depth
        return update_normal_and_depth(points, first, obs_centre, centres, ref_centre, ref_scale, scale_last)


# ---------------------------------------------------------------- the driver
class _Frame:
    pass


def _Tcw_of(nav, Tbc):
    Rwb = synth_ba.quat_to_R(nav["q"])
    Rwc, twc = Rwb @ Tbc[:3, :3], nav["p"] + Rwb @ Tbc[:3, 3]
    return frontend.pose_to_Tcw(Rwc, twc), Rwc, twc


class Replay:
    def __init__(self, seq, stages, kf_every=10, n_local=10, n_local_kfs=10, th_last=7.0, th_local=2.0, verbose=False,
                 lba_lag=0):
        # LocalMapping runs beside Tracking in the reference (src/LocalMapping.cc:113-139): the local BA of a key frame
        # made at frame k works on the map as it was then, and its write-back reaches the tracker some frames later.  A
        # fixed lag makes that reproducible: lba_lag = 0 applies the result before frame k + 1 (the inline form), lba_lag = L
        # before frame k + L (the C++ replay overlaps the solve with the L - 1 frames in between; here it is solved at
        # once and held back, which is the same map for the same frames).
        self.lba_lag, self._pending = int(lba_lag), None
        self.seq, self.S = seq, stages
        self.kf_every, self.n_local, self.n_local_kfs = kf_every, n_local, n_local_kfs
        self.th_last, self.th_local = th_last, th_local
        self.verbose = verbose
        self.Tbc = synth_ba.EUROC_TBC
        self.Tcb = np.linalg.inv(self.Tbc)
        self.scale = np.asarray(stages.scale_factors(), np.float32)
        self.inv_sigma2 = (np.float32(1.0) / (self.scale * self.scale)).astype(np.float32)
        # the map
        self.mp_X = np.zeros((0, 3), np.float32)
        self.mp_desc = np.zeros((0, 32), np.uint8)
        self.mp_bad = np.zeros(0, bool)
        self.mp_obs = []      # per point: dict key-frame id -> key index
        self.mp_normal = np.zeros((0, 3), np.float32)
        self.mp_maxd = np.zeros(0, np.float32)
        self.mp_mind = np.zeros(0, np.float32)
        self.kfs = []         # key frames: _Frame with id / nav / keys / mp_ref / imu edge from the previous one
        self.traj = []        # optimised NavState per frame
        self.stats = dict(frames=0, lba=0, lba_applied=0, ms_frames=[], ms_lba=[], n_matches=[], n_inliers=[])
        self.last = None
        self.map_updated = False

    # ---- Frame::Frame: extraction of both images + ComputeStereoMatches
    def make_frame(self, k):
        L, R = self.seq.images(k)
        f = _Frame()
        f.k, f.t = k, self.seq.time(k)
        _, f.keys, f.desc = self.S.extract(0, L)
        _, kr, dr = self.S.extract(1, R)
        f.uright, f.depth = self.S.stereo(f.keys, f.desc, kr, dr)
        f.N = len(f.keys)
        f.mp_ref = np.full(f.N, -1, np.int64)
        f.track_depth = np.full(f.N, np.inf, np.float32)
        return f

    # ---- map points
    def _add_points(self, kf, idx, Xw):
        n0 = len(self.mp_X)
        self.mp_X = np.concatenate([self.mp_X, Xw.astype(np.float32)])
        self.mp_desc = np.concatenate([self.mp_desc, kf.desc[idx]])
        self.mp_bad = np.concatenate([self.mp_bad, np.zeros(len(idx), bool)])
        for j, i in enumerate(idx):
            self.mp_obs.append({kf.id: int(i)})
        d = Xw.astype(np.float64) - kf.twc
        dist = np.linalg.norm(d, axis=1)
        self.mp_normal = np.concatenate([self.mp_normal, (d / dist[:, None]).astype(np.float32)])
        maxd = (dist * self.scale[kf.keys["octave"][idx]]).astype(np.float32)
        self.mp_maxd = np.concatenate([self.mp_maxd, maxd])
        self.mp_mind = np.concatenate([self.mp_mind, (maxd / self.scale[NLEVELS - 1]).astype(np.float32)])
        kf.mp_ref[idx] = n0 + np.arange(len(idx))

    def _update_normal_depth(self, ids):
        """MapPoint::UpdateNormalAndDepth (MapPoint.cc:424-480) for the given points, one batched call: observations in
        key-frame order, the reference key frame = the oldest observer"""
        ids = [m for m in ids if not self.mp_bad[m] and self.mp_obs[m]]
        if not ids:
            return
        kids = [sorted(self.mp_obs[m]) for m in ids]
        first = np.zeros(len(ids) + 1, np.int32)
        first[1:] = np.cumsum([len(k) for k in kids])
        obs_centre = np.fromiter((k for ks in kids for k in ks), np.int32, int(first[-1]))
        centres = np.array([k.twc for k in self.kfs], np.float32)
        ref = np.array([ks[0] for ks in kids], np.int32)
        ref_scale = np.array([self.scale[self.kfs[ks[0]].keys["octave"][self.mp_obs[m][ks[0]]]] for m, ks in zip(ids, kids)],
                             np.float32)
        nrm, mx, mn = self.S.update_normal_depth(self.mp_X[ids], first, obs_centre, centres, ref, ref_scale,
                                                 self.scale[NLEVELS - 1])
        self.mp_normal[ids], self.mp_maxd[ids], self.mp_mind[ids] = nrm, mx, mn

    # ---- Tracking::CreateNewKeyFrame + LocalMapping::ProcessNewKeyFrame
    def insert_keyframe(self, f, nav, imu_edge):
        kf = f
        kf.id = len(self.kfs)
        kf.nav = nav.copy()
        _, kf.Rwc, kf.twc = _Tcw_of(kf.nav, self.Tbc)
        kf.imu_edge = imu_edge  # (IMU_PREINT record with Sigma PRV, dt between the key frames) from the previous one
        self.kfs.append(kf)
        has = np.nonzero(kf.mp_ref >= 0)[0]
        for i in has:  # AddObservation
            self.mp_obs[int(kf.mp_ref[i])][kf.id] = int(i)
        # new points from stereo: keys with depth and without a point, nearest first; all close ones, at least 100
        cand = np.nonzero((kf.depth > 0) & (kf.mp_ref < 0))[0]
        cand = cand[np.argsort(kf.depth[cand], kind="stable")]
        close = kf.depth[cand] <= TH_DEPTH
        n_take = max(int(close.sum()), min(100, len(cand)))
        cand = cand[:n_take]
        if len(cand):
            z = kf.depth[cand].astype(np.float64)
            Xc = np.stack([(kf.keys["x"][cand] - sc.CX) * z / sc.FX, (kf.keys["y"][cand] - sc.CY) * z / sc.FY, z], 1)
            Xw = Xc @ kf.Rwc.T + kf.twc
            self._add_points(kf, cand, Xw)
        return kf

    # ---- Optimizer::LocalBundleAdjustmentNavStatePRV on the last n_local key frames
    def local_ba(self, apply=True):
        local = self.kfs[-self.n_local:]
        first = local[0].id
        prev = self.kfs[first - 1] if first > 0 else None
        local_ids = {k.id for k in local}
        pts = []
        seen = set()
        for k in local:  # lLocalMapPoints: key frames oldest first, keys in order
            for m in k.mp_ref[k.mp_ref >= 0]:
                m = int(m)
                if m not in seen and not self.mp_bad[m]:
                    seen.add(m)
                    pts.append(m)
        fixed_ids = []
        if prev is not None:
            fixed_ids.append(prev.id)
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
                    i = self.mp_obs[m][kid]
                    rows.append((index[kid], j, k.keys["x"][i], k.keys["y"][i], k.uright[i],
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
        b["Rcb"], b["tcb"] = self.Tcb[:3, :3].reshape(-1), self.Tcb[:3, 3]
        b["fx"], b["fy"], b["cx"], b["cy"], b["bf"] = sc.FX, sc.FY, sc.CX, sc.CY, sc.BF
        b["its0"], b["its1"] = 4, 6
        P[0]["gw"] = synth_ba.GRAVITY
        P[0]["inv_sigma_bg2"] = 1.0 / synth_ba.IMU_SIGMA[2] ** 2
        P[0]["inv_sigma_ba2"] = 1.0 / synth_ba.IMU_SIGMA[3] ** 2
        P[0]["lambda_init"] = 1.0
        P[0]["qRbe"][0] = 1.0
        X = self.mp_X[pts]
        close = np.zeros(len(pts), np.uint8)
        t0 = time.perf_counter()
        navs, Xo, erase, res = self.S.lba_vio(P, kfs, X, close, obs, imu)
        self.stats["ms_lba"].append(1e3 * (time.perf_counter() - t0))
        self.stats["lba"] += 1
        job = dict(local=local, kfs=kfs, pts=pts, rows=rows, navs=navs, Xo=Xo, erase=erase, res=res)
        if apply:
            self._lba_apply(job)
        return job

    def _lba_apply(self, job):
        """the write-back of Optimizer::LocalBundleAdjustmentNavStatePRV (Optimizer.cc:704-768)"""
        local, kfs, pts, rows, navs, Xo, erase, res = (job[k] for k in ("local", "kfs", "pts", "rows", "navs", "Xo", "erase",
                                                                         "res"))
        self.stats["lba_applied"] += 1
        if int(res["status"]) != 0:
            return res
        for r in np.nonzero(erase)[0]:  # ErasePairObs
            _, j, _, _, _, _, kid, i = rows[r]
            m = pts[j]
            self.mp_obs[m].pop(kid, None)
            self.kfs[kid].mp_ref[i] = -1
            if not self.mp_obs[m]:
                self.mp_bad[m] = True
        for i, k in enumerate(local):
            if not kfs[i]["fixed"]:
                k.nav = navs[i].copy()
                _, k.Rwc, k.twc = _Tcw_of(k.nav, self.Tbc)
        self.mp_X[pts] = Xo
        self._update_normal_depth(pts)
        return res

    # ---- Tracking::PredictNavStateByIMU (Tracking.cc:385-451)
    @staticmethod
    def predict(nav_ref, im):
        ns = nav_ref.copy()
        dt = float(im["dt"])
        gw = synth_ba.GRAVITY
        Rwb = synth_ba.quat_to_R(ns["q"])
        dbg, dba = ns["dbg"].copy(), ns["dba"].copy()
        M = lambda n: im[n].reshape(3, 3)  # noqa: E731
        p = ns["p"] + ns["v"] * dt + gw * (dt * dt / 2) + Rwb @ (im["pij"] + M("Jgp") @ dbg + M("Jap") @ dba)
        v = ns["v"] + gw * dt + Rwb @ (im["vij"] + M("Jgv") @ dbg + M("Jav") @ dba)
        R = Rwb @ M("Rij") @ synth_ba.so3_exp(M("JgR") @ dbg)
        ns["p"], ns["v"], ns["q"] = p, v, synth_ba._R_to_quat(R)
        ns["bg"] = ns["bg"] + dbg
        ns["ba"] = ns["ba"] + dba
        ns["dbg"], ns["dba"] = 0, 0
        return ns

    def _vio_frame(self, nav_pred, ref_nav, im, prior, dt_frames, marg):
        F = np.zeros(1, VIO_FRAME_DTYPE)
        f = F[0]
        b = f["base"]
        b["nav"] = nav_pred
        b["Rcb"], b["tcb"] = self.Tcb[:3, :3].reshape(-1), self.Tcb[:3, 3]
        b["fx"], b["fy"], b["cx"], b["cy"], b["bf"] = sc.FX, sc.FY, sc.CX, sc.CY, sc.BF
        f["nav_last"] = ref_nav
        f["imu"] = im
        f["gw"] = synth_ba.GRAVITY
        f["inv_sigma_bg2"] = 1.0 / synth_ba.IMU_SIGMA[2] ** 2
        f["inv_sigma_ba2"] = 1.0 / synth_ba.IMU_SIGMA[3] ** 2
        f["dt_frames"], f["th_depth"] = dt_frames, TH_DEPTH
        f["compute_marg"] = int(marg)
        if prior is not None:
            f["nav_prior"], f["H_prior"], f["last_has_prior"] = prior[0], prior[1], 1
        return F

    def _obs(self, f):
        idx = np.nonzero(f.mp_ref >= 0)[0]
        obs = np.zeros(len(idx), POSE_OBS_DTYPE)
        obs["Xw"] = self.mp_X[f.mp_ref[idx]]
        obs["u"], obs["v"], obs["ur"] = f.keys["x"][idx], f.keys["y"][idx], f.uright[idx]
        obs["inv_sigma2"] = self.inv_sigma2[f.keys["octave"][idx]]
        obs["flags"] = (f.track_depth[idx] < max(10.0, TH_DEPTH)).astype(np.int32)
        return obs, idx

    def _local_points(self, f):
        """UpdateLocalMap: the points of the last n_local_kfs key frames, not yet in the frame"""
        inframe = set(int(m) for m in f.mp_ref[f.mp_ref >= 0])
        out, seen = [], set()
        for k in self.kfs[-self.n_local_kfs:]:
            for m in k.mp_ref[k.mp_ref >= 0]:
                m = int(m)
                if m not in seen and m not in inframe and not self.mp_bad[m]:
                    seen.add(m)
                    out.append(m)
        return np.array(out, np.int64)

    # ---- the first frame: StereoInitialization (Tracking.cc:1500-1580) with the true state
    def initialise(self):
        f = self.make_frame(0)
        tr = self.seq.truth(0)
        nav = np.zeros(1, NAVSTATE_DTYPE)[0]
        nav["p"], nav["q"], nav["v"] = tr["p"], tr["q"], tr["v"]
        nav["bg"], nav["ba"] = self.seq.bg, self.seq.ba  # IMU initialisation done: biases known at the start
        self.insert_keyframe(f, nav, None)
        f.nav = nav.copy()
        f.prior = None
        f.outlier = np.zeros(f.N, bool)
        self.last, self.map_updated = f, True
        self.traj.append(nav.copy())
        self.last_kf_frame_imu = None

    # ---- one frame: Tracking::Track for the stereo-inertial steady state
    def step(self, k):
        t0 = time.perf_counter()
        S, last = self.S, self.last
        f = self.make_frame(k)
        ref_nav = self.kfs[-1].nav if self.map_updated else last.nav  # Tracking.cc:392-409
        prior = None if self.map_updated else last.prior
        t_ref = self.kfs[-1].t if self.map_updated else last.t
        im, prv, st = S.preintegrate(self.seq.noise, self.seq.imu_between(t_ref, f.t), t_ref, f.t, ref_nav["bg"],
                                     ref_nav["ba"])
        assert st == 0, "IMU pre-integration failed"
        nav_pred = self.predict(ref_nav, im)
        # ---- TrackWithIMU: SearchByProjection(last frame) + PoseOptimization
        Tcw, _, _ = _Tcw_of(nav_pred, self.Tbc)
        Tcw_last, _, _ = _Tcw_of(last.nav, self.Tbc)
        has = (last.mp_ref >= 0) & ~last.outlier
        has[has] &= ~self.mp_bad[last.mp_ref[has]]
        Xw = np.zeros((last.N, 3), np.float32)
        Xw[has] = self.mp_X[last.mp_ref[has]]
        pts = frontend.make_last_frame_points(last.keys, np.zeros((last.N, 32), np.uint8), Xw, has, True)
        pts["desc"][has] = self.mp_desc[last.mp_ref[has]]
        cam = frontend.make_sbp_camera(Tcw, Tcw_last, K, BOUNDS, sc.BF, sc.BASELINE, self.th_last, self.scale)
        if hasattr(S, "search_last_frame"):  # (one call: the projection and the search)
            n1, a1 = S.search_last_frame(pts, cam, f.keys, f.uright, f.desc, 0.9)
        else:
            n1, a1 = S.search(0, S.project_last_frame(pts, cam), f.keys, f.uright, f.desc, None, 0.9)
        if n1 < 20:  # the wider window of Tracking.cc:301-309
            cam[0]["th"] = 2 * self.th_last
            if hasattr(S, "search_last_frame"):
                n1, a1 = S.search_last_frame(pts, cam, f.keys, f.uright, f.desc, 0.9)
            else:
                n1, a1 = S.search(0, S.project_last_frame(pts, cam), f.keys, f.uright, f.desc, None, 0.9)
        ok = a1 >= 0
        f.mp_ref[ok] = last.mp_ref[a1[ok]]
        f.track_depth[ok] = last.track_depth[a1[ok]]
        obs1, idx1 = self._obs(f)
        F1 = self._vio_frame(nav_pred, ref_nav, im, prior, f.t - t_ref, False)
        F1[0]["base"]["n_obs"] = len(obs1)
        r1, o1 = S.pose_vio(F1, obs1)
        f.mp_ref[idx1[o1 != 0]] = -1  # Discard outliers
        # ---- TrackLocalMapWithIMU: SearchLocalPoints + PoseOptimization(bComputeMarg)
        nav1 = r1["base"]["nav"] if int(r1["base"]["status"]) == 0 else nav_pred
        Tcw1, _, _ = _Tcw_of(nav1, self.Tbc)
        cand = self._local_points(f)
        n2 = 0
        if len(cand):
            FF = np.zeros(1, FRUSTUM_FRAME_DTYPE)
            ff = FF[0]
            ff["Rcrw"], ff["tcrw"], ff["Ow"] = Tcw1[:, :3].reshape(-1), Tcw1[:, 3], -Tcw1[:, :3].T @ Tcw1[:, 3]
            ff["n_cams"], ff["use_distort"], ff["cams"] = 1, 0, self._pinhole().ctypes.data
            ff["Tcr"][0] = np.eye(4)[:3].reshape(-1)
            ff["bounds"][0] = BOUNDS
            ff["bf"], ff["n_levels"], ff["viewing_cos_limit"] = sc.BF, NLEVELS, 0.5
            ff["log_scale_factor"] = np.float32(np.log(np.float32(SCALE)))
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
        return self._finish_frame(k, f, t0)

    # ---- NeedNewKeyFrame / CreateNewKeyFrame / LocalMapping (single-threaded: runs before the next frame)
    def _finish_frame(self, k, f, t0):
        S = self.S
        n1, n2 = self.stats["n_matches"][-1]
        if k % self.kf_every == 0:
            kf_prev = self.kfs[-1]
            im_kf, prv_kf, st = S.preintegrate(self.seq.noise, self.seq.imu_between(kf_prev.t, f.t), kf_prev.t, f.t,
                                               kf_prev.nav["bg"], kf_prev.nav["ba"])
            assert st == 0
            edge = im_kf.copy()
            edge["Sigma"] = prv_kf.reshape(-1)  # mSigmaijPRV for the local BA
            f.mp_ref[f.outlier] = -1
            kf = self.insert_keyframe(f, f.nav, edge)
            self.stats["ms_frames"].append(1e3 * (time.perf_counter() - t0))
            if self.lba_lag <= 0:
                res = self.local_ba()["res"]
                f.nav = kf.nav.copy()  # mLastFrame follows its reference key frame (UpdateLastFrame)
                self.map_updated = True
            else:
                self.before_frame(k + 1)  # (a job still pending from the previous key frame is applied first)
                job = self.local_ba(apply=False)
                self._pending, res = (k + self.lba_lag, job), job["res"]
            f.outlier = np.zeros(f.N, bool)
            if self.verbose:
                print("  kf %d: lba status %d, %d trials, chi2 %.1f -> %.1f" % (
                    kf.id, int(res["status"]), int(res["lm_trials"]), res["chi2_initial"], res["chi2_final"]))
        else:
            self.stats["ms_frames"].append(1e3 * (time.perf_counter() - t0))
        self.last = f
        self.traj.append(f.nav.copy())
        self.stats["frames"] += 1
        if self.verbose:
            dt, dr = synth_ba.pose_error(f.nav, self.seq.truth(k))
            print("frame %d: matches %d + %d, inliers %d, err %.2e m %.2e rad" % (k, n1, n2, self.stats["n_inliers"][-1],
                                                                             dt, dr))
        return f.nav

    def _pinhole(self):
        if not hasattr(self, "_cam"):
            from .ba_types import CAMERA_DTYPE
            self._cam = np.zeros(1, CAMERA_DTYPE)
            c = self._cam[0]
            c["fx"], c["fy"], c["cx"], c["cy"] = sc.FX, sc.FY, sc.CX, sc.CY
        return self._cam

    def before_frame(self, k):
        """LocalMapping's pending write-back reaches the tracker before frame k (lba_lag)"""
        if self._pending is not None and k >= self._pending[0]:
            job, self._pending = self._pending[1], None
            self._lba_apply(job)
            self.map_updated = True

    def run(self, n_frames=None):
        n = n_frames or self.seq.n_frames
        self.initialise()
        for k in range(1, n):
            self.before_frame(k)
            self.step(k)
        return np.array(self.traj, NAVSTATE_DTYPE)


def first_decision_flip(stats_a, stats_b):
    """First tracked frame (1-based index into the trajectories) whose INTEGER decisions differ between two runs of the same
    replay -- matches found by the two searches, inliers kept by the second optimisation -- or None.  Up to that frame the
    two runs worked on the same inputs and must agree to the parity tolerance; behind it they track different maps (one
    observation sitting on its chi2 gate or one window candidate on its ratio test is enough), and what can be asked is
    that both keep tracking."""
    ma, mb = stats_a["n_matches"], stats_b["n_matches"]
    ia, ib = stats_a["n_inliers"], stats_b["n_inliers"]
    for k in range(min(len(ma), len(mb))):
        if tuple(ma[k]) != tuple(mb[k]) or ia[k] != ib[k]:
            return k + 1
    return None


# ---------------------------------------------------------------- one frame as ONE chain of launches
class _Arena:
    """A pinned host block and its device twin with the same layout: fields are carved out once, the block travels
    in one asynchronous copy."""

    def __init__(self, nbytes):
        import ctypes
        from ._lib import DeviceBuffer, check, lib
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        check(lib().vieo_host_alloc_pinned(ctypes.byref(p), self.nbytes), "pinned")
        self.h_ptr = p.value
        self.h = np.ctypeslib.as_array((ctypes.c_uint8 * self.nbytes).from_address(self.h_ptr))
        self.dev = DeviceBuffer(self.nbytes)
        self.d_ptr = self.dev.ptr
        self.off = 0

    def field(self, dtype, count):
        dtype = np.dtype(dtype)
        self.off = (self.off + 255) & ~255
        nb = dtype.itemsize * int(count)
        assert self.off + nb <= self.nbytes, "arena too small"
        view = self.h[self.off:self.off + nb].view(dtype)
        dptr = self.d_ptr + self.off
        self.off += nb
        return view, dptr

    def free(self):
        from ._lib import lib
        if self.h_ptr:
            lib().vieo_host_free_pinned(self.h_ptr)
            self.h_ptr = None
            self.dev.free()


class ChainedReplay(Replay):
    """The same replay with a frame's tracking as ONE chain of launches on one stream: the frame's inputs go up in one
    copy from pinned memory, extraction (both images) -> ComputeStereoMatches -> SearchByProjection(last frame) ->
    PoseOptimization -> isInFrustum + query construction from the optimised pose (vieo_track_local_queries_device) ->
    SearchByProjection(local map) -> PoseOptimization(bComputeMarg) run back to back on the device-pointer entry
    points with the bookkeeping between them in the vieo_track_* glue kernels, and the results come back in one copy;
    the host synchronises once per frame.  The local-map candidates are all points of the local key frames (the ones a
    key of the frame already holds are taken out on the device, as Tracking::SearchLocalPoints does).  Two rare
    branches are not worth a device form and re-run the frame through the stage-by-stage path of the base class: fewer
    than 20 matches in the first search (the wider window of Tracking.cc:301-309) and a failed first optimisation."""

    CCAP = 16384  # local-map candidates per frame

    def __init__(self, seq, stages, **kw):
        super().__init__(seq, stages, **kw)
        import ctypes
        from ._lib import check, lib
        from .ba_types import PROJ_QUERY_DTYPE, SBP_CAMERA_DTYPE, VIO_RESULT_DTYPE
        from .orb_extractor import KEYPOINT_DTYPE, ORBextractor
        self.L = lib()
        self.ext = ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)
        self.stream = self.L.vieo_orb_stream(self.ext._h)
        self.cap = cap = self.ext.max_keypoints()
        ccap = self.CCAP
        self.pcap = cap + ccap
        A = self.ain = _Arena(2 * W * H + 64 * cap + 3672 * 2 + 12 * self.pcap + 4 * self.pcap + (32 + 32 + 4) * ccap + (1 << 16))
        self.h_img, self.d_img = A.field(np.uint8, 2 * W * H)
        self.h_pts, self.d_pts = A.field(LAST_FRAME_POINT_DTYPE, cap)
        self.h_npts, self.d_npts = A.field(np.int32, 4)
        self.h_cam, self.d_cam = A.field(SBP_CAMERA_DTYPE, 1)
        self.h_f1, self.d_f1 = A.field(VIO_FRAME_DTYPE, 1)
        self.h_f2, self.d_f2 = A.field(VIO_FRAME_DTYPE, 1)
        self.h_xyz, self.d_xyz = A.field(np.float32, 3 * self.pcap)
        self.h_dep, self.d_dep = A.field(np.float32, self.pcap)
        self.h_cpt, self.d_cpt = A.field(FRUSTUM_POINT_DTYPE, ccap)
        self.h_cdesc, self.d_cdesc = A.field(np.uint8, 32 * ccap)
        self.h_alias, self.d_alias = A.field(np.int32, ccap)
        self.h_const, self.d_const = A.field(np.float32, 32)  # inv_sigma2[16], scale[16]
        self.h_const[:NLEVELS] = self.inv_sigma2
        self.h_const[16:16 + NLEVELS] = self.scale
        self.d_isig, self.d_scale = self.d_const, self.d_const + 64
        O = self.aout = _Arena((28 + 32) * 2 * cap + 16 * cap + 2 * VIO_RESULT_DTYPE.itemsize + 3672 + 4 * self.pcap + (1 << 14))
        self.o_cnt, self.d_cnt = O.field(np.int32, 4)
        self.o_kp, self.d_kp = O.field(KEYPOINT_DTYPE, 2 * cap)
        self.o_desc, self.d_desc = O.field(np.uint8, 2 * cap * 32)
        self.o_ur, self.d_ur = O.field(np.float32, cap)
        self.o_dp, self.d_dp = O.field(np.float32, cap)
        self.o_mpref, self.d_mpref = O.field(np.int32, cap)
        self.o_obskey, self.d_obskey = O.field(np.int32, cap)
        self.o_outl, self.d_outl = O.field(np.uint8, cap)
        self.o_r1, self.d_r1 = O.field(VIO_RESULT_DTYPE, 1)
        self.o_r2, self.d_r2 = O.field(VIO_RESULT_DTYPE, 1)
        self.o_nm, self.d_nm = O.field(np.int32, 4)
        self.o_f2, self.d_f2out = O.field(VIO_FRAME_DTYPE, 1)
        self.o_cdep, self.d_cdep = O.field(np.float32, ccap)
        self.o_nq, self.d_nq = O.field(np.int32, 4)
        from ._lib import DeviceBuffer
        self.d_q1, self.d_q2 = DeviceBuffer(64 * cap), DeviceBuffer(64 * ccap)
        self.d_assign, self.d_taken, self.d_held = DeviceBuffer(4 * cap), DeviceBuffer(cap), DeviceBuffer(self.pcap)
        self.d_obs = DeviceBuffer(32 * cap)
        self.bounds = (ctypes.c_float * 4)(*BOUNDS.tolist())
        self.ff = np.zeros(1, FRUSTUM_FRAME_DTYPE)
        ff = self.ff[0]
        ff["n_cams"], ff["use_distort"], ff["cams"] = 1, 0, self._pinhole().ctypes.data
        ff["Tcr"][0] = np.eye(4)[:3].reshape(-1)
        ff["bounds"][0] = BOUNDS
        ff["bf"], ff["n_levels"], ff["viewing_cos_limit"] = sc.BF, NLEVELS, 0.5
        ff["log_scale_factor"] = np.float32(np.log(np.float32(SCALE)))
        self.stats["fallbacks"] = 0

    def close(self):
        for a in (getattr(self, "ain", None), getattr(self, "aout", None)):
            if a is not None:
                a.free()
        self.ain = self.aout = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _all_local_points(self):
        # the local key frames and their points change only when a key frame is inserted / a local BA has run
        key = (len(self.kfs), self.stats["lba_applied"])
        if getattr(self, "_lp_key", None) == key:
            return self._lp
        self._lp_key, self._lp = key, self._all_local_points_now()
        return self._lp

    def _all_local_points_now(self):
        out, seen = [], set()
        for k in self.kfs[-self.n_local_kfs:]:
            for m in k.mp_ref[k.mp_ref >= 0]:
                m = int(m)
                if m not in seen and not self.mp_bad[m]:
                    seen.add(m)
                    out.append(m)
        return np.array(out, np.int64)

    def step(self, k):
        from ._lib import check
        t0 = time.perf_counter()
        L, last, cap, st = self.L, self.last, self.cap, self.stream
        ref_nav = self.kfs[-1].nav if self.map_updated else last.nav
        prior = None if self.map_updated else last.prior
        t_ref = self.kfs[-1].t if self.map_updated else last.t
        t = self.seq.time(k)
        im, prv, stt = self.S.preintegrate(self.seq.noise, self.seq.imu_between(t_ref, t), t_ref, t, ref_nav["bg"],
                                           ref_nav["ba"])
        assert stt == 0, "IMU pre-integration failed"
        nav_pred = self.predict(ref_nav, im)
        Tcw, _, _ = _Tcw_of(nav_pred, self.Tbc)
        Tcw_last, _, _ = _Tcw_of(last.nav, self.Tbc)
        # ---- the frame's inputs, written straight into the pinned block
        Li, Ri = self.seq.images(k)
        self.h_img[:W * H] = Li.reshape(-1)
        self.h_img[W * H:] = Ri.reshape(-1)
        has = (last.mp_ref >= 0) & ~last.outlier
        has[has] &= ~self.mp_bad[last.mp_ref[has]]
        nl = last.N
        Xw = np.zeros((nl, 3), np.float32)
        Xw[has] = self.mp_X[last.mp_ref[has]]
        pts = frontend.make_last_frame_points(last.keys, np.zeros((nl, 32), np.uint8), Xw, has, True)
        pts["desc"][has] = self.mp_desc[last.mp_ref[has]]
        self.h_pts[:nl] = pts
        self.h_npts[0] = nl
        self.h_cam[0] = frontend.make_sbp_camera(Tcw, Tcw_last, K, BOUNDS, sc.BF, sc.BASELINE, self.th_last, self.scale)[0]
        self.h_f1[0] = self._vio_frame(nav_pred, ref_nav, im, prior, t - t_ref, False)[0]
        self.h_f2[0] = self._vio_frame(nav_pred, ref_nav, im, prior, t - t_ref, True)[0]
        xyz = self.h_xyz.reshape(-1, 3)
        xyz[:nl] = Xw
        self.h_dep[:nl] = last.track_depth
        cand = self._all_local_points()
        nc = len(cand)
        assert nc <= self.CCAP, "more local-map candidates than the chained replay holds"
        if nc:
            xyz[cap:cap + nc] = self.mp_X[cand]
            cp = self.h_cpt[:nc]
            cp["Xw"], cp["normal"] = self.mp_X[cand], self.mp_normal[cand]
            cp["max_distance"], cp["min_distance"] = self.mp_maxd[cand], self.mp_mind[cand]
            self.h_cdesc[:32 * nc] = self.mp_desc[cand].reshape(-1)
            where = np.full(len(self.mp_X), -1, np.int32)  # map point -> entry of the last frame's table
            lk = np.nonzero(has)[0]
            where[last.mp_ref[lk[::-1]]] = lk[::-1]
            self.h_alias[:nc] = where[cand]
        # ---- one copy up, the chain, one copy back
        t1 = time.perf_counter()
        check(L.vieo_memcpy_h2d_async(self.ain.d_ptr, self.ain.h_ptr, self.ain.off, st))
        try:  # (rectified stereo frames without encoder: the per-call modes 1, 1 of the _ex entries)
            self.ext.extract_batch_device(self.d_img, 2, W, H, W, W * H, self.d_kp, self.d_desc, cap, self.d_cnt)
            check(L.vieo_stereo_match_rectified_batch_device(self.ext._h, 1, self.d_kp, self.d_desc, self.d_cnt, cap,
                                                             sc.BASELINE, sc.BF, self.d_ur, self.d_dp), "stereo")
            check(L.vieo_sbp_project_last_frame_batch_device(self.d_pts, self.d_npts, cap, 1, self.d_cam, self.d_q1.ptr, st))
            check(L.vieo_search_by_projection_batch_device(0, self.d_q1.ptr, self.d_npts, cap, 1, self.d_kp, self.d_ur,
                                                           self.d_desc, None, self.d_cnt, cap, 0, 2, self.bounds, 0.9, 1,
                                                           self.d_assign.ptr, self.d_nm, st), "sbp1")
            check(L.vieo_track_merge_assign_batch_device(self.d_assign.ptr, self.d_mpref, self.d_cnt, cap, 1, 0, 2, 0, 1, st))
            close = float(max(10.0, TH_DEPTH))
            check(L.vieo_track_build_obs_depth_batch_device(self.d_mpref, self.d_xyz, self.d_dep, close, self.pcap, self.d_kp,
                                                            self.d_ur, self.d_cnt, cap, 1, 0, 2, self.d_isig, self.d_obs.ptr,
                                                            self.d_obskey, self.d_f1, 1, st))
            check(L.vieo_pose_optimization_vio_batch_device_ex(self.d_f1, 1, self.d_obs.ptr, self.d_outl, self.d_r1, 1, 1, st), "pose1")
            check(L.vieo_track_after_pose_batch_device(self.d_mpref, self.d_obskey, self.d_outl, self.d_f1, self.d_r1, 1, cap,
                                                       1, self.d_f2, self.d_taken.ptr, st))
            check(L.vieo_track_mark_held_batch_device(self.d_mpref, self.d_cnt, cap, 1, 0, 2, self.d_held.ptr, self.pcap, st))
            check(L.vieo_track_local_queries_device(self.ff.ctypes.data, self.d_f1, self.d_r1, self.d_cpt, self.d_cdesc,
                                                    self.d_alias, self.d_held.ptr, self.pcap, nc, self.th_local, 0.0, self.d_scale,
                                                    self.d_q2.ptr, self.d_dep + 4 * cap, self.d_nq, st), "local queries")
            check(L.vieo_search_by_projection_batch_device(1, self.d_q2.ptr, self.d_nq, self.CCAP, 1, self.d_kp, self.d_ur,
                                                           self.d_desc, self.d_taken.ptr, self.d_cnt, cap, 0, 2, self.bounds,
                                                           0.8, 1, self.d_assign.ptr, self.d_nm + 4, st), "sbp2")
            check(L.vieo_track_merge_assign_batch_device(self.d_assign.ptr, self.d_mpref, self.d_cnt, cap, 1, 0, 2, cap, 0, st))
            check(L.vieo_track_build_obs_depth_batch_device(self.d_mpref, self.d_xyz, self.d_dep, close, self.pcap, self.d_kp,
                                                            self.d_ur, self.d_cnt, cap, 1, 0, 2, self.d_isig, self.d_obs.ptr,
                                                            self.d_obskey, self.d_f2, 1, st))
            check(L.vieo_pose_optimization_vio_batch_device_ex(self.d_f2, 1, self.d_obs.ptr, self.d_outl, self.d_r2, 1, 1, st), "pose2")
        finally:
            pass
        check(L.vieo_memcpy_d2h_async(self.aout.h_ptr, self.aout.d_ptr, self.aout.off, st))
        check(L.vieo_memcpy_d2h_async(self.o_f2.ctypes.data, self.d_f2, VIO_FRAME_DTYPE.itemsize, st))
        check(L.vieo_memcpy_d2h_async(self.o_cdep.ctypes.data, self.d_dep + 4 * cap, 4 * max(nc, 1), st))
        t2 = time.perf_counter()
        check(L.vieo_stream_synchronize(st))
        t3 = time.perf_counter()
        self.stats.setdefault("ms_chain", []).append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
        # ---- the frame as the stage-by-stage path leaves it
        r1, r2 = self.o_r1[0], self.o_r2[0]
        n1, n2 = int(self.o_nm[0]), int(self.o_nm[1])
        if n1 < 20 or int(r1["base"]["status"]) != 0:
            self.stats["fallbacks"] += 1
            return super().step(k)
        f = _Frame()
        f.k, f.t = k, t
        N = f.N = int(min(self.o_cnt[0], cap))
        f.keys, f.desc = self.o_kp[:N].copy(), self.o_desc[:32 * N].reshape(N, 32).copy()
        f.uright, f.depth = self.o_ur[:N].copy(), self.o_dp[:N].copy()
        tab = self.o_mpref[:N]
        f.mp_ref = np.full(N, -1, np.int64)
        f.track_depth = np.full(N, np.inf, np.float32)
        a = np.nonzero((tab >= 0) & (tab < cap))[0]
        f.mp_ref[a] = last.mp_ref[tab[a]]
        f.track_depth[a] = last.track_depth[tab[a]]
        b = np.nonzero(tab >= cap)[0]
        f.mp_ref[b] = cand[tab[b] - cap]
        f.track_depth[b] = self.o_cdep[tab[b] - cap]
        nobs = int(self.o_f2[0]["base"]["n_obs"])
        f.outlier = np.zeros(N, bool)
        f.outlier[self.o_obskey[:nobs][self.o_outl[:nobs] != 0]] = True
        nav1 = r1["base"]["nav"]
        f.nav = (r2["base"]["nav"] if int(r2["base"]["status"]) == 0 else nav1).copy()
        f.prior = (f.nav.copy(), r2["H_marg"].copy()) if int(r2["has_marg"]) else None
        self.map_updated = False
        self.stats["n_matches"].append((n1, n2))
        self.stats["n_inliers"].append(int(r2["base"]["n_inliers"]))
        return self._finish_frame(k, f, t0)

def ate_between(traj_a, traj_b):
    """RMSE of the position differences of two trajectories of the same frames (no alignment: same world frame)"""
    d = np.asarray(traj_a["p"]) - np.asarray(traj_b["p"])
    return float(np.sqrt((d * d).sum(1).mean()))
