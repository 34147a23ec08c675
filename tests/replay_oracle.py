"""The CPU oracle behind the stage interface of vieo_slam_amd/replay.py (test infrastructure: the sequential replay
runs once on these stages and once on HipStages, and the two trajectories are compared)."""
import numpy as np

from vieo_slam_amd import synth_scene as sc
from vieo_slam_amd.replay import BOUNDS, INI_TH, MIN_TH, NFEAT, NLEVELS, SCALE


class OracleStages:
    name = "oracle"

    def __init__(self, oracle):
        self.o = oracle
        self.extL = oracle.extractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)
        self.extR = oracle.extractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)

    def scale_factors(self):
        return np.array(self.extL.scale_factors(), np.float32)

    def extract(self, cam, image):
        return (self.extL if cam == 0 else self.extR)(image)

    def stereo(self, kl, dl, kr, dr):
        return self.o.stereo_match(self.extL, self.extR, kl, dl, kr, dr, sc.BASELINE, sc.BF)

    def preintegrate(self, noise, samples, ti, tj, bg, ba):
        out, prv, st = self.o.imu_preintegrate(noise, [samples], [ti], [tj], [bg], [ba])
        return out[0], prv[0], int(st[0])

    def project_last_frame(self, pts, cam):
        return self.o.sbp_project_last_frame(pts, cam)

    def search(self, mode, q, keys, ur, desc, taken, nn):
        return self.o.search_by_projection(mode, q, keys, ur, desc, taken, BOUNDS, nn_ratio=nn)

    def pose_vio(self, F, obs):
        return self.o.pose_optimization_vio(F, obs)

    def in_frustum(self, F, P):
        return self.o.is_in_frustum(F, P)

    def lba_vio(self, params, kfs, pts, close, obs, imu):
        return self.o.local_ba_vio(params, kfs, pts, close, obs, imu)

    def update_normal_depth(self, points, first, obs_centre, centres, ref_centre, ref_scale, scale_last):
        return self.o.update_normal_and_depth(points, first, obs_centre, centres, ref_centre, ref_scale, scale_last)


class OracleRigStages:
    """the rig replay's stage interface (vieo_slam_amd/replay_modes.py: RigReplay) on the CPU oracle"""
    name = "oracle"

    def __init__(self, oracle, nfeatures, n_cams):
        self.o = oracle
        self.ext = [oracle.extractor(nfeatures, SCALE, NLEVELS, INI_TH, MIN_TH) for _ in range(n_cams)]

    def scale_factors(self):
        return np.array(self.ext[0].scale_factors(), np.float32)

    def extract(self, c, image, lapping):
        return self.ext[c](image, lapping)

    def fisheye(self, params, keys, descs, mono):
        return self.o.stereo_fisheye(params, keys, descs, mono)

    def project_last_frame(self, pts, cam, rig):
        return self.o.sbp_project_last_frame(pts, cam, rig)

    def search(self, mode, q, keys, ur, desc, taken, bounds, cam_first, nn, ori=True):
        return self.o.search_by_projection(mode, q, keys, ur, desc, taken, bounds, nn_ratio=nn, check_ori=ori, cam_first=cam_first)

    def pose_vio(self, F, obs):
        return self.o.pose_optimization_vio(F, obs)

    def in_frustum(self, F, P):
        return self.o.is_in_frustum(F, P)

    def preintegrate(self, noise, samples, ti, tj, bg, ba):
        out, prv, st = self.o.imu_preintegrate(noise, [samples], [ti], [tj], [bg], [ba])
        return out[0], prv[0], int(st[0])

    def lba_vio(self, params, kfs, pts, close, obs, imu):
        return self.o.local_ba_vio(params, kfs, pts, close, obs, imu)

    def update_normal_depth(self, *a):
        return self.o.update_normal_and_depth(*a)


class OracleVisionStages(OracleStages):
    """the vision-only replay's stage interface (replay_modes.VisionReplay: 1000 features, no IMU) on the CPU oracle"""

    def __init__(self, oracle):
        from vieo_slam_amd.replay_modes import NFEAT_VISION
        self.o = oracle
        self.extL = oracle.extractor(NFEAT_VISION, SCALE, NLEVELS, INI_TH, MIN_TH)
        self.extR = oracle.extractor(NFEAT_VISION, SCALE, NLEVELS, INI_TH, MIN_TH)

    def pose(self, F, obs):
        return self.o.pose_optimization(F, obs)

    def lba(self, params, kfs, pts, obs):
        return self.o.local_ba(params, kfs, pts, obs)
