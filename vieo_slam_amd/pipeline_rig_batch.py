"""Device-resident batched tracking step of distorted camera-rig frames (BASELINE configs[3] shape: many rig frames /
sequences per GPU), the rig counterpart of pipeline.FramePipeline: B rig frames already in HBM run

  ExtractORB x (B * n_cams) -> ComputeStereoFishEyeMatches (device stage) -> SearchByProjection(last frame, camera loop,
  compacted queries) -> PoseOptimization(VIO, rig) -> isInFrustum + queries -> SearchByProjection(local map)
  -> PoseOptimization(VIO, rig, marg)

as one chain of batch launches on one stream with no host round trip (the same C-ABI entries the one-call rig tracker
chains for a single frame).  The last frame's map points and the local map of every frame are prepared on the host from a
stage-by-stage frame 0 (pipeline_rig.RigFrontEnd), untimed."""
import ctypes

import numpy as np

from . import frontend, synth_ba
from . import synth_fisheye as sf
from ._lib import DeviceBuffer, check, lib
from .ba_types import (FISHEYE_PARAMS_DTYPE, LAST_FRAME_POINT_DTYPE, SBP_CAMERA_DTYPE, VIO_FRAME_DTYPE, VIO_RESULT_DTYPE)
from .map_point import FRUSTUM_POINT_DTYPE
from .matching import FisheyeStereoDevice
from .orb_extractor import KEYPOINT_DTYPE, ORBextractor
from .pipeline_rig import NLEVELS, RigFrontEnd


class RigFramePipeline:
    STAGES = ("extract", "stereo", "sbp_last", "pose1", "sbp_local", "pose2", "total")

    def __init__(self, scene, cases, nfeatures, batch, seed=0, noise=True, th_depth=35.0):
        L = lib()
        self.scene, self.B, self.nc = scene, batch, len(scene.cams)
        B, nc = self.B, self.nc
        self.fe = fe = RigFrontEnd(scene, nfeatures)
        self.ext = ORBextractor(nfeatures, 1.2, NLEVELS, 20, 7)
        self.stream = L.vieo_orb_stream(self.ext._h)
        self.cap = cap = self.ext.max_keypoints()
        self.kc = kc = nc * cap
        W, H = scene.W, scene.H
        self.lap = fe.lapping
        rng = np.random.default_rng(seed)
        # ---- per base case: frame 0 stage by stage -> map points, last-frame points, local map
        prep = []
        for case in cases:
            fr0 = fe.make_frame(case["images0"])
            mps = fe.make_map_points(fr0, case["pose0"][2], case["pose0"][3])
            pts = fe.last_frame_points(fr0, mps)
            has = mps["key_mp"] >= 0
            pts["reserved"][has, 0] = mps["first_key"][mps["key_mp"][has]] + 1
            z = fr0.fe["group_p3d"][np.nonzero(fr0.fe["group_good"])[0]][:, 2].astype(np.float32)
            ld = np.full(fr0.N, np.inf, np.float32)
            ld[has] = z[mps["key_mp"][has]]
            _, P = fe._frustum(np.eye(3, 4), mps, case["pose0"])
            prep.append(dict(fr0=fr0, mps=mps, pts=pts, last_depth=ld, z=z, P=np.ascontiguousarray(P, FRUSTUM_POINT_DTYPE)))
        self.prep = prep
        self.ccap = ccap = max(len(p["P"]) for p in prep) + 16
        self.pcap = pcap = kc + ccap
        imgs = np.zeros((B, nc, H, W), np.uint8)
        pts = np.zeros((B, kc), LAST_FRAME_POINT_DTYPE)
        n_last, n_q1 = np.zeros(B, np.int32), np.zeros(B, np.int32)  # points of the last frame, (point, camera) query slots
        xyz = np.zeros((B, pcap, 3), np.float32)
        dep = np.full((B, pcap), np.inf, np.float32)
        cpt = np.zeros((B, ccap), FRUSTUM_POINT_DTYPE)
        cdesc = np.zeros((B, ccap, 32), np.uint8)
        alias = np.full((B, ccap), -1, np.int32)
        ncand = np.zeros(B, np.int32)
        f1 = np.zeros(B, VIO_FRAME_DTYPE)
        cams = np.zeros(B, SBP_CAMERA_DTYPE)
        self.d_camarr = DeviceBuffer(scene.cams.nbytes)
        self.d_camarr.upload(scene.cams)
        self.truth, self.case_of = [], []
        for b in range(B):
            ci = b % len(cases)
            case, pr = cases[ci], prep[ci]
            self.case_of.append(ci)
            for c in range(nc):
                im = case["images1"][c]
                if noise and b >= len(cases):
                    im = np.clip(im.astype(np.int16) + rng.integers(-2, 3, im.shape), 0, 255).astype(np.uint8)
                imgs[b, c] = im
            n0 = pr["fr0"].N
            pts[b, :n0] = pr["pts"]
            n_last[b], n_q1[b] = n0, n0 * nc
            has = pr["mps"]["key_mp"] >= 0
            xyz[b, :n0][has] = pr["mps"]["Xw"][pr["mps"]["key_mp"][has]]
            dep[b, :n0] = pr["last_depth"]
            ncl = len(pr["P"])
            cpt[b, :ncl], cdesc[b, :ncl] = pr["P"], pr["mps"]["desc"]
            alias[b, :ncl] = pr["mps"]["first_key"]
            ncand[b] = ncl
            xyz[b, kc:kc + ncl] = pr["mps"]["Xw"]
            F = case["vio"].copy()
            bb = F[0]["base"]
            bb["nav"]["p"] += rng.normal(0, 0.01, 3)
            bb["nav"]["q"] = synth_ba.quat_mul(bb["nav"]["q"], synth_ba.quat_from_rotvec(rng.normal(0, 0.003, 3)))
            bb["n_cams"], bb["cams"] = nc, self.d_camarr.ptr
            F[0]["th_depth"] = th_depth
            f1[b] = F[0]
            Ri, pi, Rwc0, twc0 = case["pose0"]
            cams[b] = fe._sbp_cam(bb["nav"], frontend.pose_to_Tcw(Rwc0, twc0), fe.th_last)[0][0]
            self.truth.append(case["truth"])
        f2 = f1.copy()
        f2["compute_marg"] = 1
        self.f1_host, self.imgs_host = f1, imgs
        rigs = np.repeat(fe.rig, B)
        # ---- the device-resident batch
        D = DeviceBuffer
        self.n_img = B * nc
        self.d_img = D(imgs.nbytes)
        self.d_img.upload(imgs)
        self.d_kp, self.d_desc, self.d_cnt = D(self.n_img * cap * KEYPOINT_DTYPE.itemsize), D(self.n_img * cap * 32), D(self.n_img * 8)
        sig2 = fe.sigma2
        fp = np.zeros(1, FISHEYE_PARAMS_DTYPE)
        fp[0]["n_cams"], fp[0]["n_levels"], fp[0]["bf"], fp[0]["th_far_pts"] = nc, NLEVELS, fe.bf, 0.0
        fp[0]["cams"], fp[0]["Trc"], fp[0]["Tcr"], fp[0]["level_sigma2"] = (scene.cams.ctypes.data, fe._Trc.ctypes.data,
                                                                            fe._Tcr.ctypes.data, sig2.ctypes.data)
        self.fish = FisheyeStereoDevice(fp, cap, max_frames=B)
        gcap = self.gcap = self.fish.gcap
        self.d_kcat, self.d_dcat = D(B * kc * KEYPOINT_DTYPE.itemsize), D(B * kc * 32)
        self.d_first, self.d_fcnt = D(B * (nc + 1) * 4), D(B * 8)
        self.d_depth, self.d_ur, self.d_kg = D(B * kc * 4), D(B * kc * 4), D(B * kc * 4)
        self.d_gidx, self.d_good, self.d_p3d, self.d_hdr = D(B * gcap * nc * 4), D(B * gcap), D(B * gcap * 24), D(B * 32)
        self.d_pts, self.d_cams, self.d_rigs = D(pts.nbytes), D(cams.nbytes), D(rigs.nbytes)
        self.d_n_last, self.d_n_q1 = D(n_last.nbytes), D(n_q1.nbytes)
        self.d_pts.upload(pts), self.d_cams.upload(cams), self.d_rigs.upload(rigs)
        self.d_n_last.upload(n_last), self.d_n_q1.upload(n_q1)
        self.d_q1, self.d_q1c, self.d_qsrc, self.d_nq1 = D(B * kc * nc * 64), D(B * kc * nc * 64), D(B * kc * nc * 4), D(B * 4)
        self.d_q2, self.d_nq2 = D(B * ccap * nc * 64), D(B * 4)
        self.d_assign, self.d_nm = D(B * kc * 4), D(B * 4)
        self.d_mpref, self.d_taken, self.d_held = D(B * kc * 4), D(B * kc), D(B * pcap)
        self.d_xyz, self.d_dep = D(xyz.nbytes), D(dep.nbytes)
        self.d_xyz.upload(xyz), self.d_dep.upload(dep)
        self.d_cpt, self.d_cdesc, self.d_alias, self.d_ncand = D(cpt.nbytes), D(cdesc.nbytes), D(alias.nbytes), D(ncand.nbytes)
        self.d_cpt.upload(cpt), self.d_cdesc.upload(cdesc), self.d_alias.upload(alias), self.d_ncand.upload(ncand)
        consts = np.concatenate([np.pad(fe.inv_sigma2, (0, 16 - NLEVELS)), np.pad(fe.scale, (0, 16 - NLEVELS))]).astype(np.float32)
        self.d_consts = D(consts.nbytes)
        self.d_consts.upload(consts)
        self.d_obs, self.d_obskey, self.d_outl = D(B * kc * 32), D(B * kc * 4), D(B * kc)
        self.d_f1, self.d_f2 = D(f1.nbytes), D(f2.nbytes)
        self.d_f1.upload(f1), self.d_f2.upload(f2)
        self.d_r1, self.d_r2 = D(B * VIO_RESULT_DTYPE.itemsize), D(B * VIO_RESULT_DTYPE.itemsize)
        self.bounds = np.ascontiguousarray(fe.bounds, np.float32)
        self.ff = fe._frustum(np.eye(3, 4), prep[0]["mps"], cases[0]["pose0"])[0]
        self.close = float(max(10.0, th_depth))
        check(L.vieo_device_synchronize())
        self._ev = None

    def enable_timing(self, on=True):
        self._ev, self._ev_steps = None, 0
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
        if self._ev:
            check(lib().vieo_event_record(self._ev[self._ev_steps % 64][k], self.stream))

    def stage_ms_all(self):
        out = []
        n = min(self._ev_steps, 64)
        for s in range(self._ev_steps - n, self._ev_steps):
            row, ms = self._ev[s % 64], {}
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
        L, B, nc, cap, kc, st = lib(), self.B, self.nc, self.cap, self.kc, self.stream
        W, H = self.scene.W, self.scene.H
        self._stamp(0)
        self.ext.extract_batch_device(self.d_img.ptr, self.n_img, W, H, W, W * H, self.d_kp.ptr, self.d_desc.ptr, cap,
                                      self.d_cnt.ptr, lapping=self.lap)
        self._stamp(1)
        check(L.vieo_stereo_fisheye_match_batch_device(self.fish.h, self.d_kp.ptr, self.d_desc.ptr, self.d_cnt.ptr, B,
                                                       self.d_kcat.ptr, self.d_dcat.ptr, self.d_first.ptr, self.d_fcnt.ptr,
                                                       self.d_depth.ptr, self.d_ur.ptr, self.d_kg.ptr, self.d_gidx.ptr,
                                                       self.d_good.ptr, self.d_p3d.ptr, self.d_hdr.ptr, st), "fisheye")
        self._stamp(2)
        check(L.vieo_sbp_project_last_frame_rig_batch_device(self.d_pts.ptr, self.d_n_last.ptr, kc, B, self.d_cams.ptr,
                                                             self.d_rigs.ptr, nc, self.d_q1.ptr, st), "project")
        check(L.vieo_track_compact_queries_batch_device(self.d_q1.ptr, self.d_n_q1.ptr, kc * nc, B, self.d_q1c.ptr, self.d_qsrc.ptr,
                                                        self.d_nq1.ptr, st), "compact")
        check(L.vieo_search_by_projection_rig_batch_device(0, self.d_q1c.ptr, self.d_nq1.ptr, kc * nc, B, self.d_kcat.ptr,
                                                           self.d_ur.ptr, self.d_dcat.ptr, None, self.d_first.ptr, kc,
                                                           self.bounds.ctypes.data, nc, 0.9, 1, self.d_assign.ptr, self.d_nm.ptr,
                                                           st), "sbp1")
        self._stamp(3)
        check(L.vieo_track_merge_assign_rig_batch_device(self.d_assign.ptr, self.d_mpref.ptr, self.d_fcnt.ptr, kc, B, 0, 1, 0, 1, nc,
                                                         self.d_pts.ptr, self.d_qsrc.ptr, kc * nc, st))
        check(L.vieo_track_build_obs_rig_batch_device(self.d_mpref.ptr, self.d_xyz.ptr, self.d_dep.ptr, self.close, self.pcap,
                                                      self.d_kcat.ptr, self.d_ur.ptr, self.d_fcnt.ptr, self.d_first.ptr, nc, kc, B,
                                                      self.d_consts.ptr, self.d_obs.ptr, self.d_obskey.ptr, self.d_f1.ptr, 1, st))
        check(L.vieo_pose_optimization_vio_batch_device_ex(self.d_f1.ptr, B, self.d_obs.ptr, self.d_outl.ptr, self.d_r1.ptr,
                                                           2, 1, st), "pose1")
        self._stamp(4)
        check(L.vieo_track_after_pose_batch_device(self.d_mpref.ptr, self.d_obskey.ptr, self.d_outl.ptr, self.d_f1.ptr,
                                                   self.d_r1.ptr, 1, kc, B, self.d_f2.ptr, self.d_taken.ptr, st))
        check(L.vieo_track_mark_held_batch_device(self.d_mpref.ptr, self.d_fcnt.ptr, kc, B, 0, 1, self.d_held.ptr, self.pcap, st))
        check(L.vieo_track_local_queries_batch_device(self.ff.ctypes.data, self.d_f1.ptr, self.d_r1.ptr, B, self.d_cpt.ptr,
                                                      self.d_cdesc.ptr, self.d_alias.ptr, self.d_ncand.ptr, self.ccap,
                                                      self.d_held.ptr, self.pcap, self.fe.th_local, 0.0, self.d_consts.ptr + 64,
                                                      self.d_q2.ptr, self.d_dep.ptr + 4 * kc, self.pcap, self.d_nq2.ptr, st),
              "local queries")
        check(L.vieo_search_by_projection_rig_batch_device(1, self.d_q2.ptr, self.d_nq2.ptr, self.ccap * nc, B, self.d_kcat.ptr,
                                                           self.d_ur.ptr, self.d_dcat.ptr, self.d_taken.ptr, self.d_first.ptr, kc,
                                                           self.bounds.ctypes.data, nc, self.fe.nn_local, 1, self.d_assign.ptr,
                                                           self.d_nm.ptr, st), "sbp2")
        self._stamp(5)
        check(L.vieo_track_merge_assign_rig_batch_device(self.d_assign.ptr, self.d_mpref.ptr, self.d_fcnt.ptr, kc, B, 0, 1, kc, 0, nc,
                                                         None, None, 0, st))
        check(L.vieo_track_build_obs_rig_batch_device(self.d_mpref.ptr, self.d_xyz.ptr, self.d_dep.ptr, self.close, self.pcap,
                                                      self.d_kcat.ptr, self.d_ur.ptr, self.d_fcnt.ptr, self.d_first.ptr, nc, kc, B,
                                                      self.d_consts.ptr, self.d_obs.ptr, self.d_obskey.ptr, self.d_f2.ptr, 1, st))
        check(L.vieo_pose_optimization_vio_batch_device_ex(self.d_f2.ptr, B, self.d_obs.ptr, self.d_outl.ptr, self.d_r2.ptr,
                                                           2, 1, st), "pose2")
        self._stamp(6)
        if self._ev:
            self._ev_steps += 1

    def sync(self):
        self.ext.sync()

    def results(self):
        self.sync()
        B, kc = self.B, self.kc
        return dict(r1=self.d_r1.download(VIO_RESULT_DTYPE, (B,)), r2=self.d_r2.download(VIO_RESULT_DTYPE, (B,)),
                    mp_ref=self.d_mpref.download(np.int32, (B, kc)), first=self.d_first.download(np.int32, (B, self.nc + 1)),
                    hdr=self.d_hdr.download(np.int32, (B, 8)), keys=self.d_kcat.download(KEYPOINT_DTYPE, (B, kc)),
                    depth=self.d_depth.download(np.float32, (B, kc)), nm=self.d_nm.download(np.int32, (B,)))

    def close_all(self):
        self.fish.close()
