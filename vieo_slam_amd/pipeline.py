"""Device-resident replay of the per-frame front end for a BATCH of independent stereo-inertial
frames, in the reference's call order (Tracking::TrackWithIMU + TrackLocalMapWithIMU):

  ORBextractor x2 -> ComputeStereoMatches -> SearchByProjection(last frame) -> PoseOptimization(VIO)
  -> SearchByProjection(local map) -> PoseOptimization(VIO, bComputeMarg)

All stages are the C-ABI batch entry points chained on the extractor's HIP stream; nothing returns
to the host between them.  Inputs that the hot path does not produce itself (last frame's map
points, local-map queries from Frame::isInFrustum, IMU pre-integration, predicted state) are
prepared once on the host and stay resident in HBM.

Two workloads.  "r2" (rounds 1-2): 8 views of one textured plane, the second search's queries precomputed on the host
from the last frame's own points.  "r3" (SURVEY.md 8d as written): 64 distinct cases over 8 textures with non-planar
depth (floating sheets) and low-contrast patches (cells that need minThFAST), and the head of SearchLocalPoints INSIDE
the step: Frame::isInFrustum + query construction for 4-5 k local-map candidates per frame on the device from the first
optimisation's pose in HBM (vieo_track_local_queries_batch_device), the frame's own matches dropped through the held
table -- the stage Tracking.cc:2308-2370 runs between the two optimisations."""
import ctypes

import numpy as np

from . import frontend, synth, synth_ba
from . import synth_scene as sc
from ._lib import DeviceBuffer, check, lib
from .ba_types import (LAST_FRAME_POINT_DTYPE, POSE_OBS_DTYPE, PROJ_QUERY_DTYPE, SBP_CAMERA_DTYPE,
                       VIO_FRAME_DTYPE, VIO_RESULT_DTYPE)
from .orb_extractor import KEYPOINT_DTYPE, ORBextractor

W, H = sc.W, sc.H
K = (sc.FX, sc.FY, sc.CX, sc.CY)
BOUNDS = np.array([0, W, 0, H], np.float32)
NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH = 1200, 1.2, 8, 20, 7  # EuRoC_VIO.yaml:138-151


def make_cases(n_base, seed0=1, verbose=False, workload="r2"):
    if workload == "r2":
        scene = sc.Scene(seed0)
        return [sc.make_tracking_case(seed0 + i, scene=scene) for i in range(n_base)]
    # r3: 8 textures with dull patches x as many sheet layouts as needed; every case its own layout and pose
    n_tex = min(8, n_base)
    texs = [sc.Scene(seed0 + 100 * t, relief=(seed0 + 100 * t, 12), low_contrast=True) for t in range(n_tex)]
    cases = []
    for i in range(n_base):
        scene = texs[i % n_tex] if i < n_tex else texs[i % n_tex].with_relief((seed0 + 7 * i, 12))
        c = sc.make_tracking_case(seed0 + i, scene=scene)
        c["scene"] = scene
        cases.append(c)
    return cases

N_EXTRA = 3072  # synthetic local-map candidates per frame on top of the last frame's own points (r3)


class FramePipeline:
    STAGES = ("extract", "stereo", "sbp_last", "pose1", "sbp_local", "pose2", "total")

    def __init__(self, cases, batch, seed=0, noise=True, workload="r2"):
        self.workload = workload
        self.B = B = batch
        self.ext = ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)
        self.stream = lib().vieo_orb_stream(self.ext._h)
        self.cap = cap = self.ext.max_keypoints()
        ext0 = [ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH) for _ in range(2)]
        scf = self.ext.GetScaleFactors()
        self.inv_sigma2 = (np.float32(1.0) / (scf * scf)).astype(np.float32)
        rng = np.random.default_rng(seed)
        imgs = np.zeros((B, 2, H, W), np.uint8)
        pts = np.zeros((B, cap), LAST_FRAME_POINT_DTYPE)
        npts = np.zeros(B, np.int32)
        cams = np.zeros(B, SBP_CAMERA_DTYPE)
        q2 = np.zeros((B, cap), PROJ_QUERY_DTYPE)
        xyz = np.zeros((B, 2 * cap, 3), np.float32)
        f1 = np.zeros(B, VIO_FRAME_DTYPE)
        self.truth = []
        from .matching import compute_stereo_matches
        for b in range(B):
            case = cases[b % len(cases)]

            def jitter(im):
                if not noise or b < len(cases):
                    return im
                return np.clip(im.astype(np.int16) + rng.integers(-2, 3, im.shape), 0, 255).astype(np.uint8)
            L0, R0 = [jitter(x) for x in case["images0"]]
            imgs[b, 0], imgs[b, 1] = [jitter(x) for x in case["images1"]]
            # ---- frame t0 (the "last frame"): its keys with stereo depth become map points
            _, k0, d0 = ext0[0](L0)
            _, k0r, d0r = ext0[1](R0)
            ur0, dp0 = compute_stereo_matches(ext0[0], ext0[1], k0, d0, k0r, d0r, sc.BASELINE, sc.BF)
            Ri, pi, Rwc0, twc0 = case["pose0"]
            Xw, valid = frontend.unproject_stereo(k0, dp0, K, Rwc0, twc0)
            n0 = len(k0)
            pts[b, :n0] = frontend.make_last_frame_points(k0, d0, Xw, valid, True)
            npts[b] = n0
            xyz[b, :n0] = Xw
            xyz[b, cap:cap + n0] = Xw
            # ---- predicted state of frame t1 = truth + small error (PredictNavStateByIMU)
            F = case["vio"].copy()
            F[0]["base"]["nav"]["p"] += rng.normal(0, 0.01, 3)
            F[0]["base"]["nav"]["q"] = synth_ba.quat_mul(
                F[0]["base"]["nav"]["q"], synth_ba.quat_from_rotvec(rng.normal(0, 0.003, 3)))
            f1[b] = F[0]
            Rwb = synth_ba.quat_to_R(F[0]["base"]["nav"]["q"])
            Tbc = synth_ba.EUROC_TBC
            Rwc = Rwb @ Tbc[:3, :3]
            twc = F[0]["base"]["nav"]["p"] + Rwb @ Tbc[:3, 3]
            Tcw = frontend.pose_to_Tcw(Rwc, twc)
            cams[b] = frontend.make_sbp_camera(Tcw, frontend.pose_to_Tcw(Rwc0, twc0), K, BOUNDS,
                                               sc.BF, sc.BASELINE, 7.0, scf)[0]
            # ---- local-map queries (what Frame::isInFrustum hands to the second search):
            # the same world points seen again, window 4.0 * scale[level], levels [L-1, L]
            Xc = Xw.astype(np.float64) @ Tcw[:, :3].T + Tcw[:, 3]
            z = np.where(Xc[:, 2] > 0.1, Xc[:, 2], 1.0)
            u = (sc.FX * Xc[:, 0] / z + sc.CX).astype(np.float32)
            v = (sc.FY * Xc[:, 1] / z + sc.CY).astype(np.float32)
            inimg = valid & (Xc[:, 2] > 0.1) & (u >= 0) & (u < W) & (v >= 0) & (v < H)
            q = q2[b, :n0]
            q["u"], q["v"] = u, v
            q["ur"] = u - np.float32(sc.BF) / z.astype(np.float32)
            q["radius"] = np.float32(4.0) * scf[k0["octave"]]
            q["level_min"], q["level_max"] = k0["octave"] - 1, k0["octave"]
            q["angle"] = k0["angle"]
            q["flags"] = inimg.astype(np.int32) * 3
            q["desc"] = d0
            self.truth.append(case["truth"])
        if workload == "r3":
            self._local_candidates(cases, pts, npts, xyz_last=xyz, rng=rng)
        f2 = f1.copy()
        f2["compute_marg"] = 1
        self.imgs_host = imgs
        self.n_img = 2 * B
        D = DeviceBuffer
        self.d_img = D(imgs.nbytes)
        self.d_img.upload(imgs)
        self.d_kp, self.d_desc, self.d_cnt = D(2 * B * cap * 28), D(2 * B * cap * 32), D(2 * B * 8)
        self.d_ur, self.d_dp = D(B * cap * 4), D(B * cap * 4)
        self.d_pts, self.d_npts, self.d_cams = D(pts.nbytes), D(npts.nbytes), D(cams.nbytes)
        self.d_pts.upload(pts), self.d_npts.upload(npts), self.d_cams.upload(cams)
        if workload == "r3":
            self.d_q1, self.d_q2 = D(B * cap * 64), D(B * self.ccap * 64)
        else:
            self.d_q1, self.d_q2 = D(B * cap * 64), D(q2.nbytes)
            self.d_q2.upload(q2)
        self.d_assign, self.d_nm = D(B * cap * 4), D(B * 4)
        self.d_mpref, self.d_taken = D(B * cap * 4), D(B * cap)
        if workload == "r3":
            xyz = self.xyz3
        self.d_xyz, self.d_isig = D(xyz.nbytes), D(self.inv_sigma2.nbytes)
        self.d_xyz.upload(xyz), self.d_isig.upload(self.inv_sigma2)
        self.d_obs, self.d_obskey, self.d_outl = D(B * cap * 32), D(B * cap * 4), D(B * cap)
        self.d_f1, self.d_f2 = D(f1.nbytes), D(f2.nbytes)
        self.d_f1.upload(f1), self.d_f2.upload(f2)
        self.d_r1, self.d_r2 = D(B * VIO_RESULT_DTYPE.itemsize), D(B * VIO_RESULT_DTYPE.itemsize)
        self.bounds = (ctypes.c_float * 4)(*BOUNDS.tolist())
        self.f1_host, self.pts_host, self.q2_host = f1, pts, q2
        check(lib().vieo_device_synchronize())

    # ---- r3: the local-map candidates of every frame (what UpdateLocalMap hands to SearchLocalPoints)
    def _local_candidates(self, cases, pts, npts, xyz_last, rng):
        from .ba_types import CAMERA_DTYPE
        from .map_point import FRUSTUM_FRAME_DTYPE, FRUSTUM_POINT_DTYPE
        B, cap = self.B, self.cap
        self.ccap = ccap = cap + N_EXTRA
        self.pcap = cap + ccap
        scf = self.ext.GetScaleFactors()
        cpt = np.zeros((B, ccap), FRUSTUM_POINT_DTYPE)
        cdesc = np.zeros((B, ccap, 32), np.uint8)
        alias = np.full((B, ccap), -1, np.int32)
        ncand = np.zeros(B, np.int32)
        self.xyz3 = np.zeros((B, self.pcap, 3), np.float32)
        for b in range(B):
            case = cases[b % len(cases)]
            n0 = int(npts[b])
            Xw = pts[b, :n0]["Xw"].astype(np.float64)
            valid = (pts[b, :n0]["flags"] & 1) > 0
            Ow0 = case["pose0"][3]
            d = Xw - Ow0
            dist = np.linalg.norm(d, axis=1)
            dist = np.where(valid, dist, 1.0)
            c = cpt[b, :n0]
            c["Xw"], c["normal"] = Xw, (d / dist[:, None])
            maxd = (dist * scf[pts[b, :n0]["octave"]]).astype(np.float32)
            c["max_distance"] = np.where(valid, maxd, 0)  # a key without a point: outside every distance range
            c["min_distance"] = np.where(valid, maxd / scf[NLEVELS - 1], 0)
            cdesc[b, :n0] = pts[b, :n0]["desc"]
            alias[b, :n0] = np.arange(n0)
            # points of the local key frames the frame does not hold: true surface points (they project where the
            # scene is) seen from about here, with descriptors of their own
            Xe = case["scene"].surface_points(rng, N_EXTRA)
            Ow1 = case["pose1"][3]
            de = Xe - Ow1
            diste = np.linalg.norm(de, axis=1)
            e = cpt[b, n0:n0 + N_EXTRA]
            e["Xw"], e["normal"] = Xe, de / diste[:, None]
            md = (diste * scf[rng.integers(0, NLEVELS, N_EXTRA)]).astype(np.float32)
            e["max_distance"], e["min_distance"] = md, md / scf[NLEVELS - 1]
            cdesc[b, n0:n0 + N_EXTRA] = rng.integers(0, 256, (N_EXTRA, 32), dtype=np.uint8)
            ncand[b] = n0 + N_EXTRA
            self.xyz3[b, :n0] = xyz_last[b, :n0]
            self.xyz3[b, cap:cap + n0 + N_EXTRA] = cpt[b, :n0 + N_EXTRA]["Xw"]
        D = DeviceBuffer
        self.d_cpt, self.d_cdesc, self.d_alias, self.d_ncand = D(cpt.nbytes), D(cdesc.nbytes), D(alias.nbytes), D(ncand.nbytes)
        self.d_cpt.upload(cpt), self.d_cdesc.upload(cdesc), self.d_alias.upload(alias), self.d_ncand.upload(ncand)
        self.d_held, self.d_cdep, self.d_nq = D(B * self.pcap), D(B * ccap * 4), D(B * 4)
        self.d_scale = D(scf.nbytes)
        self.d_scale.upload(np.ascontiguousarray(scf, np.float32))
        self._cam = np.zeros(1, CAMERA_DTYPE)
        self._cam[0]["fx"], self._cam[0]["fy"], self._cam[0]["cx"], self._cam[0]["cy"] = sc.FX, sc.FY, sc.CX, sc.CY
        self.ff = np.zeros(1, FRUSTUM_FRAME_DTYPE)
        ff = self.ff[0]
        ff["n_cams"], ff["use_distort"], ff["cams"] = 1, 0, self._cam.ctypes.data
        ff["Tcr"][0] = np.eye(4)[:3].reshape(-1)
        ff["bounds"][0] = BOUNDS
        ff["bf"], ff["n_levels"], ff["viewing_cos_limit"] = sc.BF, NLEVELS, 0.5
        ff["log_scale_factor"] = np.float32(np.log(np.float32(SCALE)))
        self.ncand_host = ncand

    # ---- HIP-event stamps between stage groups (ring of 64 steps), on the pipeline's stream
    def enable_timing(self, on=True):
        self._ev = None
        self._ev_steps = 0
        if on:
            self._ev = []
            for _ in range(64):
                row = []
                for _ in range(len(self.STAGES)):
                    e = ctypes.c_void_p()
                    check(lib().vieo_event_create(ctypes.byref(e)))
                    row.append(e)
                self._ev.append(row)

    def _stamp(self, k):
        if getattr(self, "_ev", None):
            check(lib().vieo_event_record(self._ev[self._ev_steps % 64][k], self.stream))

    def stage_ms_all(self):
        out = []
        n = min(self._ev_steps, 64)
        for s in range(self._ev_steps - n, self._ev_steps):
            row = self._ev[s % 64]
            ms = {}
            for k, name in enumerate(self.STAGES[:-1]):
                v = ctypes.c_float()
                check(lib().vieo_event_elapsed_ms(row[k], row[k + 1], ctypes.byref(v)))
                ms[name] = v.value
            v = ctypes.c_float()
            check(lib().vieo_event_elapsed_ms(row[0], row[-1], ctypes.byref(v)))
            ms["total"] = v.value
            out.append(ms)
        return out

    def step(self):
        # every frame here is a rectified stereo frame without encoder (Frame::usedistort_ false): the camera-rig and
        # encoder kernel instances are skipped by the per-call modes of the _ex entries (VIEO_POSE_CAMS_RECTIFIED = 1,
        # VIEO_POSE_ENC_NONE = 1)
        self._step()

    def _step(self):
        L, B, cap, st = lib(), self.B, self.cap, self.stream
        self._stamp(0)
        self.ext.extract_batch_device(self.d_img.ptr, self.n_img, W, H, W, W * H, self.d_kp.ptr,
                                      self.d_desc.ptr, cap, self.d_cnt.ptr)
        self._stamp(1)
        check(L.vieo_stereo_match_rectified_batch_device(self.ext._h, B, self.d_kp.ptr, self.d_desc.ptr,
                                                         self.d_cnt.ptr, cap, sc.BASELINE, sc.BF,
                                                         self.d_ur.ptr, self.d_dp.ptr), "stereo")
        self._stamp(2)
        check(L.vieo_sbp_project_last_frame_batch_device(self.d_pts.ptr, self.d_npts.ptr, cap, B,
                                                         self.d_cams.ptr, self.d_q1.ptr, st), "project")
        check(L.vieo_search_by_projection_batch_device(0, self.d_q1.ptr, self.d_npts.ptr, cap, B,
                                                       self.d_kp.ptr, self.d_ur.ptr, self.d_desc.ptr,
                                                       None, self.d_cnt.ptr, cap, 0, 2, self.bounds, 0.9,
                                                       1, self.d_assign.ptr, self.d_nm.ptr, st), "sbp1")
        self._stamp(3)
        check(L.vieo_track_merge_assign_batch_device(self.d_assign.ptr, self.d_mpref.ptr, self.d_cnt.ptr,
                                                     cap, B, 0, 2, 0, 1, st))
        check(L.vieo_track_build_obs_batch_device(self.d_mpref.ptr, self.d_xyz.ptr,
                                                  self.pcap if self.workload == "r3" else 2 * cap, self.d_kp.ptr,
                                                  self.d_ur.ptr, self.d_cnt.ptr, cap, B, 0, 2,
                                                  self.d_isig.ptr, self.d_obs.ptr, self.d_obskey.ptr,
                                                  self.d_f1.ptr, 1, st))
        check(L.vieo_pose_optimization_vio_batch_device_ex(self.d_f1.ptr, B, self.d_obs.ptr, self.d_outl.ptr,
                                                           self.d_r1.ptr, 1, 1, st), "pose1")
        self._stamp(4)
        check(L.vieo_track_after_pose_batch_device(self.d_mpref.ptr, self.d_obskey.ptr, self.d_outl.ptr,
                                                   self.d_f1.ptr, self.d_r1.ptr, 1, cap, B, self.d_f2.ptr,
                                                   self.d_taken.ptr, st))
        if self.workload == "r3":  # SearchLocalPoints' head on the device: held points, isInFrustum, window queries
            check(L.vieo_track_mark_held_batch_device(self.d_mpref.ptr, self.d_cnt.ptr, cap, B, 0, 2, self.d_held.ptr,
                                                      self.pcap, st))
            check(L.vieo_track_local_queries_batch_device(self.ff.ctypes.data, self.d_f1.ptr, self.d_r1.ptr, B, self.d_cpt.ptr,
                                                          self.d_cdesc.ptr, self.d_alias.ptr, self.d_ncand.ptr, self.ccap,
                                                          self.d_held.ptr, self.pcap, 2.0, 0.0, self.d_scale.ptr,
                                                          self.d_q2.ptr, self.d_cdep.ptr, self.ccap, self.d_nq.ptr, st),
                  "local queries")
            nq, qcap, ptab = self.d_nq.ptr, self.ccap, self.pcap
        else:
            nq, qcap, ptab = self.d_npts.ptr, cap, 2 * cap
        check(L.vieo_search_by_projection_batch_device(1, self.d_q2.ptr, nq, qcap, B,
                                                       self.d_kp.ptr, self.d_ur.ptr, self.d_desc.ptr,
                                                       self.d_taken.ptr, self.d_cnt.ptr, cap, 0, 2,
                                                       self.bounds, 0.8, 1, self.d_assign.ptr,
                                                       self.d_nm.ptr, st), "sbp2")
        self._stamp(5)
        check(L.vieo_track_merge_assign_batch_device(self.d_assign.ptr, self.d_mpref.ptr, self.d_cnt.ptr,
                                                     cap, B, 0, 2, cap, 0, st))
        check(L.vieo_track_build_obs_batch_device(self.d_mpref.ptr, self.d_xyz.ptr, ptab, self.d_kp.ptr,
                                                  self.d_ur.ptr, self.d_cnt.ptr, cap, B, 0, 2,
                                                  self.d_isig.ptr, self.d_obs.ptr, self.d_obskey.ptr,
                                                  self.d_f2.ptr, 1, st))
        check(L.vieo_pose_optimization_vio_batch_device_ex(self.d_f2.ptr, B, self.d_obs.ptr, self.d_outl.ptr,
                                                           self.d_r2.ptr, 1, 1, st), "pose2")
        self._stamp(6)
        if getattr(self, "_ev", None):
            self._ev_steps += 1

    def sync(self):
        self.ext.sync()

    def results(self):
        self.sync()
        B, cap = self.B, self.cap
        return dict(r1=self.d_r1.download(VIO_RESULT_DTYPE, (B,)),
                    r2=self.d_r2.download(VIO_RESULT_DTYPE, (B,)),
                    counts=self.d_cnt.download(np.int32, (2 * B, 2)),
                    f2=self.d_f2.download(VIO_FRAME_DTYPE, (B,)),
                    mp_ref=self.d_mpref.download(np.int32, (B, cap)),
                    uright=self.d_ur.download(np.float32, (B, cap)),
                    kps=self.d_kp.download(KEYPOINT_DTYPE, (2 * B, cap)),
                    desc=self.d_desc.download(np.uint8, (2 * B, cap, 32)))
