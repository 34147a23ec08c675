"""Geometrically consistent synthetic stereo-inertial sequences (no dataset exists here or on the
GPU box): a textured plane rendered through the EuRoC pinhole/rectified-stereo model
from known body poses, with IMU pre-integration between consecutive frames.  Used by the
end-to-end tracking test (extract -> stereo -> projection search -> pose optimisation vs ground
truth) and by bench.py."""
import numpy as np

from . import synth, synth_ba
from .ba_types import VIO_FRAME_DTYPE

W, H = 752, 480
FX, FY, CX, CY, BF = synth_ba.FX, synth_ba.FY, synth_ba.CX, synth_ba.CY, synth_ba.BF
BASELINE = BF / FX
TEXEL = 0.008  # metres per texel of the plane texture
TEX_W, TEX_H = 2304, 1728


class Scene:
    """Plane z = 0 of the world, textured; cameras look at it from above."""

    def __init__(self, seed):
        self.tex = synth.synth_image_f32(seed, TEX_W, TEX_H, n_shapes=2600)
        self.Tcb = np.linalg.inv(synth_ba.EUROC_TBC)
        self.Tbc = synth_ba.EUROC_TBC

    def render(self, Rwc, twc, noise_seed):
        """uint8 image of the pinhole camera at (Rwc, twc), plus per-pixel camera depth."""
        v, u = np.mgrid[0:H, 0:W].astype(np.float64)
        d_c = np.stack([(u - CX) / FX, (v - CY) / FY, np.ones_like(u)], -1)
        d_w = d_c @ Rwc.T
        lam = -twc[2] / d_w[..., 2]  # plane z = 0
        P = twc + lam[..., None] * d_w
        tx = P[..., 0] / TEXEL + TEX_W / 2
        ty = P[..., 1] / TEXEL + TEX_H / 2
        x0 = np.clip(np.floor(tx).astype(np.int64), 0, TEX_W - 2)
        y0 = np.clip(np.floor(ty).astype(np.int64), 0, TEX_H - 2)
        fx = np.clip(tx - x0, 0, 1).astype(np.float32)
        fy = np.clip(ty - y0, 0, 1).astype(np.float32)
        T = self.tex
        img = (T[y0, x0] * (1 - fx) * (1 - fy) + T[y0, x0 + 1] * fx * (1 - fy) +
               T[y0 + 1, x0] * (1 - fx) * fy + T[y0 + 1, x0 + 1] * fx * fy)
        return synth.quantise(img.astype(np.float32), noise_seed), lam

    def stereo(self, Rwb, pwb, noise_seed):
        """(left, right, depth_left) for the body pose (Rwb, pwb)."""
        Rwc = Rwb @ self.Tbc[:3, :3]
        twc = pwb + Rwb @ self.Tbc[:3, 3]
        left, depth = self.render(Rwc, twc, noise_seed)
        right, _ = self.render(Rwc, twc + Rwc @ np.array([BASELINE, 0, 0]), noise_seed + 1)
        return left, right, depth, Rwc, twc


def look_down_pose(rng):
    """A body pose whose camera sees the plane at 3..8 m: camera ~4.5 m above, tilted."""
    tilt = np.deg2rad(rng.uniform(15, 30))
    yaw = rng.uniform(-np.pi, np.pi)
    # camera optical axis pointing down, tilted forward
    Rz = synth_ba.so3_exp(np.array([0, 0, yaw]))
    Rx = synth_ba.so3_exp(np.array([np.pi - tilt, 0, 0]))  # flips z to look at -z_world
    Rwc = Rz @ Rx
    twc = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(4.0, 5.0)])
    Tcb = np.linalg.inv(synth_ba.EUROC_TBC)
    Rwb = Rwc @ Tcb[:3, :3]
    pwb = twc + Rwc @ Tcb[:3, 3]
    return Rwb, pwb


def make_tracking_case(seed, dt_frame=0.05, scene=None):
    """Two consecutive stereo frames with IMU between them.
    returns dict(images0=(L,R), images1=(L,R), pose0/pose1=(Rwb,pwb,Rwc,twc), depth0, vio (a
    VIO_FRAME_DTYPE[1] template with nav = truth, to be perturbed by the caller), truth)."""
    rng = np.random.default_rng(seed + 31337)
    scene = scene or Scene(seed)
    Rj, pj = look_down_pose(rng)
    pi, Ri, vi, vj, bg, ba, meas = synth_ba.imu_motion(rng, pj, Rj, dt_frame)
    L0, R0, depth0, Rwc0, twc0 = scene.stereo(Ri, pi, 10 * seed)
    L1, R1, depth1, Rwc1, twc1 = scene.stereo(Rj, pj, 10 * seed + 5)
    F = np.zeros(1, VIO_FRAME_DTYPE)
    f = F[0]
    Tcb = scene.Tcb
    b = f["base"]
    b["Rcb"] = Tcb[:3, :3].reshape(-1)
    b["tcb"] = Tcb[:3, 3]
    b["fx"], b["fy"], b["cx"], b["cy"], b["bf"] = FX, FY, CX, CY, BF
    synth_ba._nav(b["nav"], pj, synth_ba._R_to_quat(Rj), vj, bg, ba)
    synth_ba._nav(f["nav_last"], pi, synth_ba._R_to_quat(Ri), vi, bg, ba)
    synth_ba.fill_imu(f["imu"], meas)
    f["gw"] = synth_ba.GRAVITY
    f["inv_sigma_bg2"] = 1.0 / synth_ba.IMU_SIGMA[2] ** 2
    f["inv_sigma_ba2"] = 1.0 / synth_ba.IMU_SIGMA[3] ** 2
    f["dt_frames"] = dt_frame
    f["th_depth"] = 35.0
    return dict(images0=(L0, R0), images1=(L1, R1), pose0=(Ri, pi, Rwc0, twc0),
                pose1=(Rj, pj, Rwc1, twc1), depth0=depth0, depth1=depth1, vio=F,
                truth=dict(p=pj, q=synth_ba._R_to_quat(Rj), v=vj))
