"""Checker (test infrastructure): ONE frame of a device-resident batch of vieo_slam_amd.pipeline.FramePipeline (workload
r3: isInFrustum + queries + both searches + both optimisations on the device) re-evaluated stage by stage on the CPU
oracle from the same inputs.  Used by tests/test_pipeline.py and by bench.py's `parity_sample` (after its timed region)."""
import numpy as np

from vieo_slam_amd import frontend, synth_ba
from vieo_slam_amd import synth_scene as sc
from vieo_slam_amd.ba_types import POSE_OBS_DTYPE, SBP_CAMERA_DTYPE
from vieo_slam_amd.map_point import FRUSTUM_POINT_DTYPE

BOUNDS = np.array([0, 752, 0, 480], np.float32)


def obs_from(mp_ref, xyz, keys, ur, inv_sigma2):
    idx = np.nonzero(mp_ref >= 0)[0]
    obs = np.zeros(len(idx), POSE_OBS_DTYPE)
    obs["Xw"] = xyz[mp_ref[idx]]
    obs["u"], obs["v"], obs["ur"] = keys["x"][idx], keys["y"][idx], ur[idx]
    obs["inv_sigma2"] = inv_sigma2[keys["octave"][idx]]
    return obs, idx


class R3FrameChecker:
    """P: the pipeline (after a step), R = P.results().  check(b) -> dict of what differs for frame b."""

    def __init__(self, P, R, oracle):
        self.P, self.R, self.o = P, R, oracle
        B = P.B
        self.xyz = P.d_xyz.download(np.float32, (B, P.pcap, 3))
        self.cpt = P.d_cpt.download(FRUSTUM_POINT_DTYPE, (B, P.ccap))
        self.cdesc = P.d_cdesc.download(np.uint8, (B, P.ccap, 32))
        self.cams = P.d_cams.download(SBP_CAMERA_DTYPE, (B,))
        self.scf = np.asarray(P.ext.GetScaleFactors(), np.float32)
        self.oL, self.oR = oracle.extractor(1200), oracle.extractor(1200)

    def check(self, b):
        P, R, o = self.P, self.R, self.o
        cap = P.cap
        _, k1, d1 = self.oL(P.imgs_host[b, 0])
        _, kr, dr = self.oR(P.imgs_host[b, 1])
        n = len(k1)
        out = dict(frame=int(b), n_keys=n, retried=int((k1["response"] < 20).sum()))
        out["keys_equal"] = bool(R["counts"][2 * b, 0] == n and
                                 np.array_equal(R["kps"][2 * b, :n].view(np.uint8), k1.view(np.uint8)))
        ur, _ = o.stereo_match(self.oL, self.oR, k1, d1, kr, dr, sc.BASELINE, sc.BF)
        out["uright_equal"] = bool(np.array_equal(ur.view(np.uint32), R["uright"][b, :n].view(np.uint32)))
        n0 = int(np.count_nonzero(np.any(P.pts_host[b]["desc"] != 0, axis=1)))
        q1 = o.sbp_project_last_frame(P.pts_host[b][:n0], np.array([self.cams[b]]))
        _, a1 = o.search_by_projection(0, q1, k1, ur, d1, None, BOUNDS)
        mp = np.where(a1 >= 0, a1, -1)
        obs1, idx1 = obs_from(mp, self.xyz[b], k1, ur, P.inv_sigma2)
        F1 = np.array([P.f1_host[b]])
        F1[0]["base"]["n_obs"] = len(obs1)
        r1, o1 = o.pose_optimization_vio(F1, obs1)
        mp[idx1[o1 != 0]] = -1
        taken = (mp >= 0).astype(np.uint8)
        # SearchLocalPoints at the first optimisation's pose
        nav = r1["base"]["nav"]
        Rwb = synth_ba.quat_to_R(nav["q"])
        Rcb, tcb = F1[0]["base"]["Rcb"].reshape(3, 3), F1[0]["base"]["tcb"]
        Rcw = Rcb @ Rwb.T
        tcw = tcb - Rcw @ nav["p"]
        FF = P.ff.copy()
        FF[0]["Rcrw"], FF[0]["tcrw"], FF[0]["Ow"] = Rcw.reshape(-1), tcw, -Rcw.T @ tcw
        nc = int(P.ncand_host[b])
        info = o.is_in_frustum(FF, self.cpt[b, :nc])
        held = np.zeros(P.pcap, bool)
        held[mp[mp >= 0]] = True
        al = np.full(nc, -1)
        al[:n0] = np.arange(n0)
        info["n"][(al >= 0) & held[np.maximum(al, 0)]] = 0
        q2, owner = frontend.queries_from_track_info(info, self.cdesc[b, :nc], 2.0, self.scf)
        out["n_local_queries"] = len(q2)
        # the device keeps one slot per candidate; same order, the empty slots carry flags = 0
        full = np.zeros(nc, q2.dtype)
        full[owner] = q2
        _, a2 = o.search_by_projection(1, full, k1, ur, d1, taken, BOUNDS, nn_ratio=0.8)
        mp = np.where(a2 >= 0, cap + a2, mp)
        out["matches_equal"] = bool(np.array_equal(R["mp_ref"][b, :n], mp))
        obs2, idx2 = obs_from(mp, self.xyz[b], k1, ur, P.inv_sigma2)
        F2 = F1.copy()
        F2[0]["base"]["nav"] = r1["base"]["nav"]
        F2[0]["base"]["n_obs"] = len(obs2)
        F2[0]["compute_marg"] = 1
        r2, o2 = o.pose_optimization_vio(F2, obs2)
        se3, inl = 0.0, True
        for ref, got in ((r1, R["r1"][b]), (r2, R["r2"][b])):
            dt, dr = synth_ba.pose_error(ref["base"]["nav"], got["base"]["nav"])
            se3 = max(se3, float(dt), float(dr))
            inl = inl and int(ref["base"]["n_inliers"]) == int(got["base"]["n_inliers"])
        out["max_se3_error"], out["inliers_equal"] = se3, bool(inl)
        gdt, gdr = synth_ba.pose_error(R["r2"][b]["base"]["nav"], P.truth[b])
        out["error_vs_truth"] = (float(gdt), float(gdr))
        return out
