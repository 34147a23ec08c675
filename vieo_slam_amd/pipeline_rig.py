"""Front end of ONE frame of a distorted camera rig (BASELINE configs[3] / [4]: Radtan EuRoC stereo, KB8 TUM-VI
stereo, up to 4 cameras), in the reference's call order (Frame::Frame -> Tracking::TrackWithIMU ->
TrackLocalMapWithIMU; src/Frame.cc:225-316, src/Tracking.cc:261-378,453-488):

  ORBextractor x n_cams -> ComputeStereoFishEyeMatches -> SearchByProjection(last frame, camera loop)
  -> PoseOptimization(VIO, rig) -> Frame::isInFrustum -> SearchByProjection(local map) -> PoseOptimization(VIO, marg)

Every stage is one call through the C-ABI (vieo_hot.h); the glue between them -- what Tracking.cc does with the
Frame / MapPoint objects -- is plain numpy here.  `track()` records the inputs and outputs of every stage so that a
test can re-evaluate each stage with the CPU oracle on the very same inputs (tests/test_pipeline_rig.py)."""
import numpy as np

from . import frontend, synth_ba
from . import synth_fisheye as sf
from .ba_types import FISHEYE_PARAMS_DTYPE, LAST_FRAME_POINT_DTYPE, POSE_OBS_DTYPE
from .map_point import FRUSTUM_FRAME_DTYPE, FRUSTUM_POINT_DTYPE
from .orb_extractor import KEYPOINT_DTYPE

NLEVELS, SCALE, INI_TH, MIN_TH = 8, 1.2, 20, 7


class HipStages:
    """The stage calls of the chain on the HIP path (C-ABI through the ctypes mirrors)."""

    def __init__(self, nfeatures, n_cams):
        from .matching import ORBmatcher
        from .orb_extractor import ORBextractor
        self.ext = [ORBextractor(nfeatures, SCALE, NLEVELS, INI_TH, MIN_TH) for _ in range(n_cams)]
        self.M = ORBmatcher

    def extract(self, c, image, lapping):
        return self.ext[c](image, None, lapping)

    def fisheye(self, params, keys, descs, mono):
        from .matching import compute_stereo_fisheye_matches
        return compute_stereo_fisheye_matches(params, keys, descs, mono)

    def project_last_frame(self, pts, cam, rig):
        return self.M.project_last_frame(pts, cam, rig)

    def search(self, mode, q, keys, ur, desc, taken, bounds, cam_first, nn, ori=True):
        return self.M(nn, ori)._search(mode, q, keys, ur, desc, taken, bounds, cam_first=cam_first)

    def pose_vio(self, F, obs):
        from .optimizer import Optimizer
        return Optimizer.PoseOptimizationVIO(F, obs)

    def in_frustum(self, F, P):
        from .map_point import is_in_frustum
        return is_in_frustum(F, P)


class RigFrame:
    """What Frame::Frame leaves behind for a rig frame: per-camera keys, the camera-major key list, depths."""

    def __init__(self, keys, descs, mono, fe):
        self.cam_keys, self.cam_descs, self.mono, self.fe = keys, descs, mono, fe
        self.cam_first = np.concatenate([[0], np.cumsum([len(k) for k in keys])]).astype(np.int32)
        self.keys = np.concatenate(keys) if len(keys) else np.zeros(0, KEYPOINT_DTYPE)
        self.desc = np.ascontiguousarray(np.concatenate(descs))
        self.N = len(self.keys)
        self.uright = np.full(self.N, -1.0, np.float32)  # Frame.cc:759
        self.depth = fe["depth"]
        self.key_cam = (np.searchsorted(self.cam_first, np.arange(self.N), side="right") - 1).astype(np.int32)


class RigFrontEnd:
    def __init__(self, scene, nfeatures, stages=None, th_last=7.0, th_local=2.0, nn_local=0.8):
        self.scene, self.nf = scene, nfeatures
        self.nc = len(scene.cams)
        self.S = stages or HipStages(nfeatures, self.nc)
        self.th_last, self.th_local, self.nn_local = th_last, th_local, nn_local
        W, H = scene.W, scene.H
        self.bounds = np.tile(np.array([0, W, 0, H], np.float32), (self.nc, 1))
        self.rig = sf.make_sbp_rig(scene.cams, scene.Tcr, self.bounds, True)
        self.scale = (np.float32(SCALE) ** np.arange(NLEVELS, dtype=np.float32)).astype(np.float32)
        sig2 = (self.scale * self.scale).astype(np.float32)
        self.sigma2 = np.ascontiguousarray(sig2)
        self.inv_sigma2 = (np.float32(1.0) / sig2).astype(np.float32)
        # KB8 cameras pass their lapping area (Frame.cc:269-273; TUM_VI_512_VIO.yaml:88-91: the whole image)
        self.lapping = [0, W - 1] if scene.cams[0]["model"] == 2 else None
        c0 = scene.cams[0]
        self.bf = 0.11 * float(c0["fx"])
        self.K0 = (float(c0["fx"]), float(c0["fy"]), float(c0["cx"]), float(c0["cy"]))
        # vieo_fisheye_params (host pointers kept alive on self)
        self._Trc, self._Tcr = sf.rig_extrinsics(scene.Tcr)
        self.fparams = np.zeros(1, FISHEYE_PARAMS_DTYPE)
        P = self.fparams[0]
        P["n_cams"], P["n_levels"], P["bf"], P["th_far_pts"] = self.nc, NLEVELS, self.bf, 0.0
        P["cams"], P["Trc"], P["Tcr"], P["level_sigma2"] = (scene.cams.ctypes.data, self._Trc.ctypes.data,
                                                            self._Tcr.ctypes.data, self.sigma2.ctypes.data)
        self.trace = []

    def _rec(self, name, inputs, outputs):
        self.trace.append((name, inputs, outputs))
        return outputs

    # ---- Frame::Frame (Frame.cc:225-316): extraction per camera + ComputeStereoFishEyeMatches
    def make_frame(self, images):
        keys, descs, mono = [], [], []
        for c, im in enumerate(images):
            m, k, d = self._rec("extract", dict(c=c, image=im, lapping=self.lapping), self.S.extract(c, im, self.lapping))
            keys.append(k), descs.append(d), mono.append(m)
        mono = np.array(mono, np.int32)
        fe = self._rec("fisheye", dict(keys=keys, descs=descs, mono=mono), self.S.fisheye(self.fparams, keys, descs, mono))
        return RigFrame(keys, descs, mono, fe)

    # ---- map points of the last frame: every good stereo group (UnprojectStereo of the rig, world frame)
    @staticmethod
    def make_map_points(fr, Rwc, twc):
        good = np.nonzero(fr.fe["group_good"])[0]
        Xw = (fr.fe["group_p3d"][good] @ Rwc.T + twc).astype(np.float32)
        key_mp = np.full(fr.N, -1, np.int32)
        gmap = np.full(len(fr.fe["group_good"]), -1, np.int32)
        gmap[good] = np.arange(len(good))
        has = fr.fe["key_group"] >= 0
        key_mp[has] = gmap[fr.fe["key_group"][has]]
        first_key = np.full(len(good), -1, np.int64)
        for n in np.nonzero(key_mp >= 0)[0][::-1]:
            first_key[key_mp[n]] = n  # lowest key index of the group
        return dict(Xw=Xw, desc=fr.desc[first_key], octave=fr.keys["octave"][first_key], key_mp=key_mp,
                    first_key=first_key)

    def last_frame_points(self, fr, mps):
        pts = np.zeros(fr.N, LAST_FRAME_POINT_DTYPE)
        has = mps["key_mp"] >= 0
        pts["Xw"][has] = mps["Xw"][mps["key_mp"][has]]
        pts["octave"], pts["angle"] = fr.keys["octave"], fr.keys["angle"]
        pts["flags"] = has.astype(np.int32) * 3
        pts["desc"][has] = mps["desc"][mps["key_mp"][has]]
        return pts

    def _sbp_cam(self, nav, Tcw_last, th):
        Rwb = synth_ba.quat_to_R(nav["q"])
        Tbc = self.scene.Tbc
        Rwc, twc = Rwb @ Tbc[:3, :3], nav["p"] + Rwb @ Tbc[:3, 3]
        Tcw = frontend.pose_to_Tcw(Rwc, twc)
        return frontend.make_sbp_camera(Tcw, Tcw_last, self.K0, self.bounds[0], self.bf, self.bf / self.K0[0], th,
                                        self.scale), Tcw

    def _obs(self, fr, mp_ref, mps, track_depth=None, th_depth=35.0):
        """track_depth: mTrackDepth per map point -> the close bit of the observation (bClose of the visual-inertial
        PoseOptimization, include/Optimizer.h:561: track_depth_ < max(10, mThDepth)); None: no close bits."""
        idx = np.nonzero(mp_ref >= 0)[0]
        obs = np.zeros(len(idx), POSE_OBS_DTYPE)
        obs["Xw"] = mps["Xw"][mp_ref[idx]]
        obs["u"], obs["v"], obs["ur"] = fr.keys["x"][idx], fr.keys["y"][idx], -1.0
        obs["inv_sigma2"] = self.inv_sigma2[fr.keys["octave"][idx]]
        obs["flags"] = fr.key_cam[idx] << 8
        if track_depth is not None:
            obs["flags"] |= (track_depth[mp_ref[idx]] < np.float32(max(10.0, th_depth))).astype(np.int32)
        return obs, idx

    def _frustum(self, Tcw, mps, pose0):
        F = np.zeros(1, FRUSTUM_FRAME_DTYPE)
        f, R = F[0], self.rig[0]
        f["Rcrw"], f["tcrw"], f["Ow"] = Tcw[:, :3].reshape(-1), Tcw[:, 3], -Tcw[:, :3].T @ Tcw[:, 3]
        f["n_cams"], f["use_distort"], f["cams"] = self.nc, 1, self.scene.cams.ctypes.data
        for c in range(self.nc):
            f["Tcr"][c], f["trc"][c], f["bounds"][c] = R["Tcr"][c], R["trc"][c], R["bounds"][c]
        f["bf"], f["n_levels"], f["viewing_cos_limit"] = self.bf, NLEVELS, 0.5
        f["log_scale_factor"] = np.float32(np.log(np.float32(SCALE)))
        P = np.zeros(len(mps["Xw"]), FRUSTUM_POINT_DTYPE)
        P["Xw"] = mps["Xw"]
        d = mps["Xw"].astype(np.float64) - pose0[3]  # from the camera centre that created the point
        dist = np.linalg.norm(d, axis=1)
        P["normal"] = (d / dist[:, None]).astype(np.float32)
        P["max_distance"] = (dist * self.scale[mps["octave"]]).astype(np.float32)
        P["min_distance"] = P["max_distance"] / self.scale[NLEVELS - 1]
        return F, P

    # ---- Tracking::TrackWithIMU + TrackLocalMapWithIMU for one frame
    def track(self, case, rng=None, pred=None, track_depth=None):
        """case: synth_scene.make_rig_tracking_case.  returns dict(r1, r2, mp_ref, frame, map points).
        pred = (NavState, IMU pre-integration record): the predicted state and the measurement of the optimiser problems
        (default: truth + a small error, the case's own pre-integration); track_depth: mTrackDepth of the last frame's map
        points -> close bits on the observations (default: none, also none for the local-map candidates)."""
        self.trace = []
        rng = rng or np.random.default_rng(0)
        fr0 = self.make_frame(case["images0"])
        Ri, pi, Rwc0, twc0 = case["pose0"]
        mps = self.make_map_points(fr0, Rwc0, twc0)
        fr1 = self.make_frame(case["images1"])
        # predicted state of the current frame = truth + a small error (PredictNavStateByIMU)
        F1 = case["vio"].copy()
        b = F1[0]["base"]
        if pred is None:
            b["nav"]["p"] += rng.normal(0, 0.01, 3)
            b["nav"]["q"] = synth_ba.quat_mul(b["nav"]["q"], synth_ba.quat_from_rotvec(rng.normal(0, 0.003, 3)))
        else:
            b["nav"], F1[0]["imu"] = pred
        th_depth = float(F1[0]["th_depth"])
        td = None if track_depth is None else np.array(track_depth, np.float32)
        b["n_cams"], b["cams"] = self.nc, self.scene.cams.ctypes.data
        cam, Tcw = self._sbp_cam(b["nav"], frontend.pose_to_Tcw(Rwc0, twc0), self.th_last)
        pts = self.last_frame_points(fr0, mps)
        q1 = self._rec("project_last_frame", dict(pts=pts, cam=cam), self.S.project_last_frame(pts, cam, self.rig))
        n1, a1 = self._rec("search", dict(mode=0, q=q1, fr=fr1, taken=None, nn=0.9),
                           self.S.search(0, q1, fr1.keys, fr1.uright, fr1.desc, None, self.bounds, fr1.cam_first, 0.9))
        mp_ref = np.full(fr1.N, -1, np.int32)
        ok = a1 >= 0
        mp_ref[ok] = mps["key_mp"][a1[ok] // self.nc]  # query (i, camj) belongs to last-frame key i
        obs1, idx1 = self._obs(fr1, mp_ref, mps, td, th_depth)
        F1[0]["base"]["n_obs"] = len(obs1)
        r1, o1 = self._rec("pose_vio", dict(F=F1, obs=obs1), self.S.pose_vio(F1, obs1))
        mp_ref[idx1[o1 != 0]] = -1  # "Discard outliers" (Tracking.cc:1903-1921)
        # ---- TrackLocalMap: the map points not yet in the frame, through isInFrustum
        nav1 = r1["base"]["nav"]
        _, Tcw1 = self._sbp_cam(nav1, frontend.pose_to_Tcw(Rwc0, twc0), self.th_last)
        FF, P = self._frustum(Tcw1, mps, case["pose0"])
        in_frame = np.zeros(len(mps["Xw"]), bool)
        in_frame[mp_ref[mp_ref >= 0]] = True
        cand = np.nonzero(~in_frame)[0]
        info = self._rec("in_frustum", dict(F=FF, P=P[cand]), self.S.in_frustum(FF, P[cand]))
        q2, owner = frontend.queries_from_track_info(info, mps["desc"][cand], self.th_local, self.scale)
        taken = (mp_ref >= 0).astype(np.uint8)
        n2, a2 = self._rec("search", dict(mode=1, q=q2, fr=fr1, taken=taken, nn=self.nn_local),
                           self.S.search(1, q2, fr1.keys, fr1.uright, fr1.desc, taken, self.bounds, fr1.cam_first,
                                         self.nn_local))
        ok = a2 >= 0
        mp_ref[ok] = cand[owner[a2[ok]]]
        if td is not None:
            td[cand] = info["track_depth"]  # isInFrustum sets mTrackDepth of every candidate it sees
        obs2, idx2 = self._obs(fr1, mp_ref, mps, td, th_depth)
        F2 = F1.copy()
        F2[0]["base"]["nav"] = nav1
        F2[0]["base"]["n_obs"] = len(obs2)
        F2[0]["compute_marg"] = 1
        r2, o2 = self._rec("pose_vio", dict(F=F2, obs=obs2), self.S.pose_vio(F2, obs2))
        return dict(r1=r1, r2=r2, o2=o2, mp_ref=mp_ref, fr0=fr0, fr1=fr1, mps=mps, n1=n1, n2=n2, obs2=obs2, idx2=idx2,
                    frustum_points=P, cand=cand)
