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
