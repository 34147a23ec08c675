"""Seeded synthetic motion-only BA problems: a scene of map points at 1-15 m,
a ground-truth body pose, EuRoC extrinsics/intrinsics, pixel noise sigma = scale[level], gross
outliers, 70 % stereo / 30 % mono observations, initial estimate = truth + (3 cm, 1 deg)."""
import numpy as np

from .ba_types import POSE_FRAME_DTYPE, POSE_OBS_DTYPE

# Examples/Stereo/EuRoC/EuRoC_VIO.yaml: Camera.Tbc (rows), Camera.fx.., Camera.bf
EUROC_TBC = np.array([[0.01632106431347947, -0.9998055939694457, 0.01106332348365739, -0.0216401454975],
                      [0.9997100876913918, 0.01651340320778997, 0.0175227875128659, -0.064676986768],
                      [-0.01770207409877133, 0.01077412554806465, 0.9997852543378366, 0.00981073058949],
                      [0, 0, 0, 1.0]])
FX = FY = 435.2046959714599
CX, CY = 367.4517211914062, 252.2008514404297
BF = 47.90639384423901
W, H = 752, 480


def quat_from_rotvec(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.array([1.0, 0, 0, 0])
    return np.concatenate([[np.cos(th / 2)], np.sin(th / 2) * w / th])


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                     a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])


def pose_error(nav_a, nav_b):
    """(translation error in m, rotation error in rad) between two navstate records."""
    dt = float(np.linalg.norm(nav_a["p"] - nav_b["p"]))
    qa, qb = nav_a["q"], nav_b["q"]
    d = abs(float(np.dot(qa, qb)))
    return dt, 2 * np.arccos(min(1.0, d))


def make_pose_problem(seed, n_obs=300, outlier_frac=0.10, stereo_frac=0.70, noise=1.0,
                      pert_t=0.03, pert_r_deg=1.0):
    """returns (frame[1] POSE_FRAME_DTYPE, obs[n] POSE_OBS_DTYPE, truth dict)."""
    rng = np.random.default_rng(seed)
    Tcb = np.linalg.inv(EUROC_TBC)
    Rcb, tcb = Tcb[:3, :3], Tcb[:3, 3]
    q_gt = quat_from_rotvec(rng.normal(0, 0.6, 3))
    p_gt = rng.uniform(-5, 5, 3)
    Rwb = quat_to_R(q_gt)
    # points: uniform in the image, depth 1..15 m
    z = rng.uniform(1.0, 15.0, n_obs)
    u = rng.uniform(20, W - 20, n_obs)
    v = rng.uniform(20, H - 20, n_obs)
    Xc = np.stack([(u - CX) / FX * z, (v - CY) / FY * z, z], 1)
    Xb = (Xc - tcb) @ Rcb  # Rcb^T (Xc - tcb)
    Xw = Xb @ Rwb.T + p_gt
    Xw32 = Xw.astype(np.float32)
    # re-project the float32 points (what the optimiser will see)
    Xc2 = (Xw32.astype(np.float64) - p_gt) @ Rwb @ Rcb.T + tcb
    level = rng.integers(0, 8, n_obs)
    sig = 1.2 ** level
    uu = FX * Xc2[:, 0] / Xc2[:, 2] + CX + rng.normal(0, noise, n_obs) * sig
    vv = FY * Xc2[:, 1] / Xc2[:, 2] + CY + rng.normal(0, noise, n_obs) * sig
    ur = FX * Xc2[:, 0] / Xc2[:, 2] + CX - BF / Xc2[:, 2] + rng.normal(0, noise, n_obs) * sig
    is_out = rng.random(n_obs) < outlier_frac
    uu[is_out] += rng.uniform(-60, 60, is_out.sum())
    vv[is_out] += rng.uniform(-60, 60, is_out.sum())
    mono = rng.random(n_obs) >= stereo_frac
    obs = np.zeros(n_obs, POSE_OBS_DTYPE)
    obs["Xw"] = Xw32
    obs["u"], obs["v"] = uu, vv
    obs["ur"] = np.where(mono, -1.0, ur)
    obs["inv_sigma2"] = (np.float32(1.0) / (np.float32(1.2) ** level).astype(np.float32) ** 2)
    obs["flags"] = (Xc2[:, 2] < 35.0).astype(np.int32)
    frame = np.zeros(1, POSE_FRAME_DTYPE)
    f = frame[0]
    dq = quat_from_rotvec(rng.normal(0, 1, 3) / np.sqrt(3) * np.deg2rad(pert_r_deg))
    f["nav"]["p"] = p_gt + rng.normal(0, 1, 3) / np.sqrt(3) * pert_t
    f["nav"]["q"] = quat_mul(q_gt, dq)
    f["Rcb"] = Rcb.reshape(-1)
    f["tcb"] = tcb
    f["fx"], f["fy"], f["cx"], f["cy"], f["bf"] = FX, FY, CX, CY, BF
    f["obs_begin"], f["n_obs"] = 0, n_obs
    return frame, obs, {"p": p_gt, "q": q_gt, "is_outlier": is_out}
