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
                      pert_t=0.03, pert_r_deg=1.0, rig=None, enc=False):
    """returns (frame[1] POSE_FRAME_DTYPE, obs[n] POSE_OBS_DTYPE, truth dict).
    enc: attach an encoder edge to the last frame (truth["enc"] keeps the record alive).
    rig = (cams, size) from camera_rig(): monocular observations spread over the distorted cameras of the rig
    (Frame::usedistort_), camera index in bits 8..11 of obs.flags; truth["cams"] keeps the array alive."""
    rng = np.random.default_rng(seed)
    if rig is not None:
        return _make_pose_problem_rig(rng, n_obs, outlier_frac, noise, pert_t, pert_r_deg, rig)
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
    gt = {"p": p_gt, "q": q_gt, "is_outlier": is_out}
    if enc:
        gt["enc"] = make_pose_enc(np.random.default_rng(seed + 4242), p_gt, q_gt)
        f["enc"] = gt["enc"].ctypes.data
    return frame, obs, gt


def make_pose_enc(rng, p_cur, q_cur, noise=1.0):
    """vieo_pose_enc for a frame whose TRUE pose is (p_cur, q_cur): a last frame 0.05 s earlier on a smooth motion
    and the wheel-odometry pre-integration between the two (consistent up to the sensor noise)."""
    Rj = quat_to_R(q_cur)
    Ri = Rj @ so3_exp(rng.normal(0, 0.02, 3)).T
    pi_ = p_cur - Rj @ rng.normal(0, 0.03, 3)
    return enc_between(rng, pi_, Ri, p_cur, Rj, noise)


def enc_between(rng, pi_, Ri, pj, Rj, noise=1.0, dt=0.05):
    """vieo_pose_enc: the wheel-odometry pre-integration between the true poses (pi, Ri) and (pj, Rj)."""
    from .ba_types import POSE_ENC_DTYPE
    E = np.zeros(1, POSE_ENC_DTYPE)
    Reb = ENC_RBE.T
    dR = Reb @ Ri.T @ Rj @ ENC_RBE
    dp = Reb @ (Ri.T @ (pj - pi_) - ENC_PBE + Ri.T @ Rj @ ENC_PBE)
    sphi, sp = 2e-3, 5e-3
    e = E[0]
    e["enc"]["dt"] = dt
    e["enc"]["delx"][:3] = so3_log_np(dR) + rng.normal(0, sphi, 3) * noise
    e["enc"]["delx"][3:] = dp + rng.normal(0, sp, 3) * noise
    e["enc"]["Sigma"] = np.diag([sphi ** 2] * 3 + [sp ** 2] * 3).reshape(-1)
    e["qRbe"], e["pbe"] = _R_to_quat(ENC_RBE), ENC_PBE
    e["p_last"], e["q_last"] = pi_, _R_to_quat(Ri)
    return E


def _make_pose_problem_rig(rng, n_obs, outlier_frac, noise, pert_t, pert_r_deg, rig):
    cams, _ = rig
    q_gt = quat_from_rotvec(rng.normal(0, 0.6, 3))
    p_gt = rng.uniform(-5, 5, 3)
    Rwb = quat_to_R(q_gt)
    ci = rng.integers(0, len(cams), n_obs)
    z = rng.uniform(1.0, 15.0, n_obs)
    xn, yn = rng.uniform(-0.7, 0.7, n_obs), rng.uniform(-0.45, 0.45, n_obs)
    obs = np.zeros(n_obs, POSE_OBS_DTYPE)
    level = rng.integers(0, 8, n_obs)
    sig = 1.2 ** level
    is_out = rng.random(n_obs) < outlier_frac
    for i in range(n_obs):
        c = cams[ci[i]]
        Rcb, tcb = c["Rcb"].reshape(3, 3), c["tcb"]
        Xc = np.array([xn[i] * z[i], yn[i] * z[i], z[i]])
        Xw32 = (Rwb @ (Rcb.T @ (Xc - tcb)) + p_gt).astype(np.float32)
        Xc2 = Rcb @ (Rwb.T @ (Xw32.astype(np.float64) - p_gt)) + tcb
        u, v = project_camera(c, Xc2)
        u += rng.normal(0, noise) * sig[i]
        v += rng.normal(0, noise) * sig[i]
        if is_out[i]:
            u += rng.uniform(-60, 60)
            v += rng.uniform(-60, 60)
        o = obs[i]
        o["Xw"], o["u"], o["v"], o["ur"] = Xw32, u, v, -1.0
        o["flags"] = int(Xc2[2] < 35.0) | (int(ci[i]) << 8)
    obs["inv_sigma2"] = (np.float32(1.0) / (np.float32(1.2) ** level).astype(np.float32) ** 2)
    frame = np.zeros(1, POSE_FRAME_DTYPE)
    f = frame[0]
    dq = quat_from_rotvec(rng.normal(0, 1, 3) / np.sqrt(3) * np.deg2rad(pert_r_deg))
    f["nav"]["p"] = p_gt + rng.normal(0, 1, 3) / np.sqrt(3) * pert_t
    f["nav"]["q"] = quat_mul(q_gt, dq)
    f["Rcb"], f["tcb"] = cams[0]["Rcb"], cams[0]["tcb"]
    f["fx"], f["fy"], f["cx"], f["cy"], f["bf"] = cams[0]["fx"], cams[0]["fy"], cams[0]["cx"], cams[0]["cy"], 0
    f["obs_begin"], f["n_obs"] = 0, n_obs
    f["n_cams"], f["cams"] = len(cams), cams.ctypes.data
    return frame, obs, {"p": p_gt, "q": q_gt, "is_outlier": is_out, "cams": cams}


# ----------------------------------------------------------------------------------------------
# Visual-inertial problems.  The IMU measurement is produced by a numpy restatement of
# IMUPreIntegratorBase::update (reference src/Odom/OdomPreIntegrator.h:432-506, mid-point samples as
# in PreIntegration :403-424) so that Delta R/v/p, the bias Jacobians and Sigma_ij are consistent.
IMU_SIGMA = (1.6968e-4, 2.0e-3, 1.9393e-5, 3.0e-3)  # EuRoC_VIO.yaml:13-17 (gyro, acc, bg walk, ba walk)
IMU_FREQ = 200.0
GRAVITY = np.array([0.0, 0.0, -9.81])


def so3_exp(w):
    return quat_to_R(quat_from_rotvec(np.asarray(w, float)))


ENC_PBE = np.array([0.10, 0.02, -0.05])  # body <- encoder extrinsics of the synthetic wheel odometry


def so3_log_np(R):
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    return w if th < 1e-8 else w * th / np.sin(th)


def so3_hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def so3_Jr(w):
    th = np.linalg.norm(w)
    O = so3_hat(w)
    if th < 1e-5:
        return np.eye(3) - 0.5 * O + O @ O / 6.0
    K = so3_hat(w / th)
    return np.eye(3) - (1 - np.cos(th)) / th * K + (1 - np.sin(th) / th) * K @ K


class Preintegrator:
    def __init__(self):
        self.R = np.eye(3)
        self.v = np.zeros(3)
        self.p = np.zeros(3)
        self.Sigma = np.zeros((9, 9))  # order p, v, Phi (mSigmaij)
        self.JgR = np.zeros((3, 3))
        self.Jgv = np.zeros((3, 3))
        self.Jav = np.zeros((3, 3))
        self.Jgp = np.zeros((3, 3))
        self.Jap = np.zeros((3, 3))
        self.dt = 0.0
        # dt_cov_noise_fix = 1: Sigma_g/a are multiplied by freq once (OdomData.h:48-52)
        self.Sg = np.eye(3) * IMU_SIGMA[0] ** 2 * IMU_FREQ
        self.Sa = np.eye(3) * IMU_SIGMA[1] ** 2 * IMU_FREQ

    def update(self, omega, acc, dt):
        dt2 = dt * dt / 2
        dR = so3_exp(omega * dt)
        Jr = so3_Jr(omega * dt)
        sk = so3_hat(acc)
        A = np.eye(9)
        A[6:9, 6:9] = dR.T
        A[3:6, 6:9] = -self.R @ sk * dt
        A[0:3, 6:9] = -self.R @ sk * dt2
        A[0:3, 3:6] = np.eye(3) * dt
        Bg = np.zeros((9, 3))
        Bg[6:9] = Jr * dt
        Ba = np.zeros((9, 3))
        Ba[3:6] = self.R * dt
        Ba[0:3] = self.R * dt2
        self.Sigma = A @ self.Sigma @ A.T + Bg @ self.Sg @ Bg.T + Ba @ self.Sa @ Ba.T
        self.Jap += self.Jav * dt - self.R * dt2
        self.Jgp += self.Jgv * dt - self.R @ sk @ self.JgR * dt2
        self.Jav += -self.R * dt
        self.Jgv += -self.R @ sk @ self.JgR * dt
        self.JgR = dR.T @ self.JgR - Jr * dt
        self.p = self.p + self.v * dt + self.R @ (acc * dt2)
        self.v = self.v + self.R @ (acc * dt)
        R = self.R @ dR
        u, _, vt = np.linalg.svd(R)
        self.R = u @ vt
        self.dt += dt


def imu_motion(rng, pj, Rj, dt_frame):
    """Random smooth motion ending at pose (pj, Rj): constant body rate and world acceleration
    over dt_frame, sampled at IMU_FREQ.  State i is defined so that the NOISE-FREE discrete
    pre-integration has zero residual between (i, j); the returned measurement carries sensor
    noise and bias.  returns (pi, Ri, vi, vj, bg, ba, Preintegrator)."""
    omega = rng.normal(0, 0.4, 3)             # body angular rate (rad/s)
    a_w = rng.normal(0, 1.0, 3)               # world acceleration (m/s^2)
    vi = rng.normal(0, 0.8, 3)
    bg, ba = rng.normal(0, 0.01, 3), rng.normal(0, 0.05, 3)
    n_imu = int(round(dt_frame * IMU_FREQ))
    h = dt_frame / n_imu
    ts = np.arange(n_imu + 1) * h
    Rrel = [so3_exp(omega * t) for t in ts]   # R_i^T R(t)
    rot = Preintegrator()
    for k in range(n_imu):
        rot.update(omega, np.zeros(3), h)
    Ri = Rj @ rot.R.T
    acc_true = [(Ri @ Rrel[k]).T @ (a_w - GRAVITY) for k in range(n_imu + 1)]
    clean = Preintegrator()
    for k in range(n_imu):
        clean.update(omega, (acc_true[k] + acc_true[k + 1]) / 2, h)
    vj = vi + GRAVITY * dt_frame + Ri @ clean.v
    pi = pj - (vi * dt_frame + GRAVITY * dt_frame ** 2 / 2 + Ri @ clean.p)
    sg = IMU_SIGMA[0] * np.sqrt(IMU_FREQ)
    sa = IMU_SIGMA[1] * np.sqrt(IMU_FREQ)
    gyr = [omega + bg + rng.normal(0, sg, 3) for _ in range(n_imu + 1)]
    acc = [acc_true[k] + ba + rng.normal(0, sa, 3) for k in range(n_imu + 1)]
    meas = Preintegrator()
    for k in range(n_imu):
        meas.update((gyr[k] + gyr[k + 1]) / 2 - bg, (acc[k] + acc[k + 1]) / 2 - ba, h)
    meas.samples = (ts, np.array(gyr), np.array(acc))  # the raw samples on [0, dt_frame] (IMUData: t, w, a)
    return pi, Ri, vi, vj, bg, ba, meas


def fill_imu(im, meas):
    im["dt"] = meas.dt
    im["Rij"], im["vij"], im["pij"] = meas.R.reshape(-1), meas.v, meas.p
    for name in ("JgR", "Jgv", "Jav", "Jgp", "Jap"):
        im[name] = getattr(meas, name).reshape(-1)
    im["Sigma"] = meas.Sigma.reshape(-1)



def _nav(rec, p, q, v, bg, ba):
    rec["p"], rec["q"], rec["v"], rec["bg"], rec["ba"] = p, q, v, bg, ba
    rec["dbg"] = 0
    rec["dba"] = 0


def make_vio_problem(seed, n_obs=300, dt_frame=0.05, compute_marg=False, prior=None, enc=False, imu=True, **kw):
    """returns (vio_frame[1] VIO_FRAME_DTYPE, obs, truth).  prior = None -> last state fixed;
    prior = (nav_prior record, H_prior 15x15, nav_last record) -> last state optimised too.
    enc: attach the wheel-odometry edge between the last and the current frame (truth["enc"] keeps the record
    alive); imu = False drops the IMU edge (dt = 0), e.g. to leave the encoder as the only odometry."""
    from .ba_types import VIO_FRAME_DTYPE
    frame, obs, gt = make_pose_problem(seed, n_obs=n_obs, **kw)
    rng = np.random.default_rng(seed + 909)
    F = np.zeros(1, VIO_FRAME_DTYPE)
    f = F[0]
    f["base"] = frame[0]
    # ground-truth state j comes from make_pose_problem; integrate BACKWARDS to get state i
    Rj, pj = quat_to_R(gt["q"]), gt["p"]
    pi, Ri, vi, vj, bg, ba, meas = imu_motion(rng, pj, Rj, dt_frame)
    fill_imu(f["imu"], meas)
    if not imu:
        f["imu"]["dt"] = 0
    qi = _R_to_quat(Ri)
    # current frame: keep the perturbed p/q from make_pose_problem, add v and biases
    f["base"]["nav"]["v"] = vj + rng.normal(0, 0.05, 3)
    f["base"]["nav"]["bg"], f["base"]["nav"]["ba"] = bg, ba
    _nav(f["nav_last"], pi, qi, vi, bg, ba)
    f["gw"] = GRAVITY
    f["inv_sigma_bg2"] = 1.0 / IMU_SIGMA[2] ** 2
    f["inv_sigma_ba2"] = 1.0 / IMU_SIGMA[3] ** 2
    f["dt_frames"] = dt_frame
    f["th_depth"] = 35.0 * BF / FX * 0 + 35.0
    f["compute_marg"] = int(compute_marg)
    if prior is not None:
        nav_prior, H_prior, nav_last = prior
        f["nav_prior"] = nav_prior
        f["H_prior"] = np.asarray(H_prior).reshape(-1)
        f["nav_last"] = nav_last
        f["last_has_prior"] = 1
    gt = dict(gt, v=vj, p_i=pi, q_i=qi, v_i=vi, bg=bg, ba=ba)
    if enc:
        gt["enc"] = enc_between(np.random.default_rng(seed + 4243), pi, Ri, pj, Rj, dt=dt_frame)
        f["base"]["enc"] = gt["enc"].ctypes.data
    return F, obs, gt


def _R_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def _observe(rng, poses, Xw, Rcb, tcb, noise, outlier_frac, stereo_frac, rig=None):
    """Observations (kf [| cam << 24], mp, u, v, ur, inv_sigma2) of the points from the key frames.
    rig = None: the rectified pinhole stereo camera; otherwise (cams, (width, height)) from camera_rig():
    monocular observations in every camera that sees the point."""
    obs_list = []
    for m in range(len(Xw)):
        for k, pose in enumerate(poses):
            Rk, pk = pose[0], pose[1]
            Xb = Rk.T @ (Xw[m] - pk)
            if rig is None:
                views = [(0, Rcb @ Xb + tcb, None)]
            else:
                views = [(ci, c["Rcb"].reshape(3, 3) @ Xb + c["tcb"], c) for ci, c in enumerate(rig[0])]
            for ci, Xck, c in views:
                if Xck[2] < 0.5:
                    continue
                if c is None:
                    uu, vv = FX * Xck[0] / Xck[2] + CX, FY * Xck[1] / Xck[2] + CY
                    w_, h_ = W, H
                else:
                    uu, vv = project_camera(c, Xck)
                    w_, h_ = rig[1]
                    if np.hypot(Xck[0], Xck[1]) / Xck[2] > (0.9 if c["model"] == 1 else 2.0):
                        continue  # outside the field of view the distortion model is meant for
                if not (10 < uu < w_ - 10 and 10 < vv < h_ - 10):
                    continue
                lvl = int(rng.integers(0, 8))
                sig = 1.2 ** lvl
                uo = uu + rng.normal(0, noise) * sig
                vo = vv + rng.normal(0, noise) * sig
                ur = uu - BF / Xck[2] + rng.normal(0, noise) * sig
                if rng.random() < outlier_frac:
                    uo += rng.uniform(-40, 40)
                    vo += rng.uniform(-40, 40)
                mono = rng.random() >= stereo_frac or c is not None
                obs_list.append((k | (ci << 24), m, uo, vo, -1.0 if mono else ur,
                                 1.0 / (np.float32(1.2) ** lvl) ** 2))
    return obs_list


def _keep_points(obs_arr, n_points, n_local):
    """points with >= 2 observations and at least one local observer"""
    m = obs_arr[:, 1].astype(np.int64)
    cnt = np.bincount(m, minlength=n_points)
    loc = np.bincount(m, weights=((obs_arr[:, 0].astype(np.int64) & 0xFFFFFF) < n_local), minlength=n_points)
    return (cnt >= 2) & (loc > 0)


def _observe_span(rng, poses, Xw, anchor, span, n_local, Rcb, tcb, noise, outlier_frac, stereo_frac):
    """_observe for large maps, vectorised per key frame (rectified pinhole only): a point is looked for in the
    local key frames within `span` indices of its anchor and in the non-local (fixed) ones -- the covisibility
    a SLAM map has, instead of every key frame seeing everything along a slow trajectory."""
    rows = []
    for k, pose in enumerate(poses):
        Rk, pk = pose[0], pose[1]
        cand = np.nonzero(np.abs(anchor - k) <= span)[0] if k < n_local else np.arange(len(Xw))
        if len(cand) == 0:
            continue
        Xc = (Xw[cand] - pk) @ Rk @ Rcb.T + tcb
        ok = Xc[:, 2] >= 0.5
        uu = FX * Xc[:, 0] / np.where(ok, Xc[:, 2], 1) + CX
        vv = FY * Xc[:, 1] / np.where(ok, Xc[:, 2], 1) + CY
        ok &= (uu > 10) & (uu < W - 10) & (vv > 10) & (vv < H - 10)
        cand, Xc, uu, vv = cand[ok], Xc[ok], uu[ok], vv[ok]
        n = len(cand)
        lvl = rng.integers(0, 8, n)
        sig = 1.2 ** lvl
        uo = uu + rng.normal(0, noise, n) * sig
        vo = vv + rng.normal(0, noise, n) * sig
        ur = uu - BF / Xc[:, 2] + rng.normal(0, noise, n) * sig
        out = rng.random(n) < outlier_frac
        uo[out] += rng.uniform(-40, 40, out.sum())
        vo[out] += rng.uniform(-40, 40, out.sum())
        mono = rng.random(n) >= stereo_frac
        isig = 1.0 / (np.float32(1.2) ** lvl).astype(np.float64) ** 2
        rows.append(np.stack([np.full(n, k, float), cand.astype(float), uo, vo, np.where(mono, -1.0, ur), isig], 1))
    return np.concatenate(rows) if rows else np.zeros((0, 6))


# ----------------------------------------------------------------------------------------------
def _scene_points(rng, local_poses, n_points, anchors):
    """points 2-12 m in front of `anchors` cameras spread evenly over the local key frames (1: the middle one,
    the local-BA windows; more: maps that cover a long trajectory, the global BAs).  returns (Xw, z)."""
    n_local = len(local_poses)
    ids = [n_local // 2] if anchors <= 1 else [int(round(a)) for a in np.linspace(0, n_local - 1, anchors)]
    z = rng.uniform(2.0, 12.0, n_points)
    u = rng.uniform(-150, W + 150, n_points)
    v = rng.uniform(-100, H + 100, n_points)
    Xc = np.stack([(u - CX) / FX * z, (v - CY) / FY * z, z], 1)
    Xw = np.zeros_like(Xc)
    which = np.arange(n_points) % len(ids)
    for t, a in enumerate(ids):
        Rm, pm = local_poses[a][0], local_poses[a][1]
        Rwc_m = Rm @ EUROC_TBC[:3, :3]
        twc_m = pm + Rm @ EUROC_TBC[:3, 3]
        Xw[which == t] = Xc[which == t] @ Rwc_m.T + twc_m
    _scene_points.anchor = np.asarray(ids)[which]  # anchor key frame of every point (for _observe_span)
    return Xw, z


def make_lba_problem(seed, n_local=10, n_fixed=6, n_points=2000, outlier_frac=0.03, stereo_frac=0.7,
                     noise=1.0, pert_t=0.01, pert_r_deg=0.3, pert_x=0.02, first_fixed=False, rig=None, anchors=1, span=None):
    """Seeded local-BA window: key frames on a smooth trajectory looking at a cloud
    of points 2-12 m ahead; every point is observed by the key frames that see it.
    returns (params[1], kfs[n_kf], points float32[n_mp,3], obs[n_obs] sorted by mp, truth)."""
    from .ba_types import LBA_KEYFRAME_DTYPE, LBA_OBS_DTYPE, LBA_PARAMS_DTYPE
    rng = np.random.default_rng(seed)
    Tcb = np.linalg.inv(EUROC_TBC)
    Rcb, tcb = Tcb[:3, :3], Tcb[:3, 3]
    n_kf = n_local + n_fixed
    # trajectory: fixed key frames first in time (older), then the local window; stored local first
    R0 = quat_to_R(quat_from_rotvec(rng.normal(0, 0.5, 3)))
    p0 = rng.uniform(-2, 2, 3)
    vel = R0 @ np.array([0.0, 0.0, 0.0]) + rng.normal(0, 0.25, 3)
    omega = rng.normal(0, 0.08, 3)
    poses = []
    for k in range(n_kf):
        t = 0.5 * k
        Rk = R0 @ so3_exp(omega * t)
        pk = p0 + vel * t + 0.02 * np.array([np.sin(t), np.cos(1.3 * t), np.sin(0.7 * t)])
        poses.append((Rk, pk))
    order = list(range(n_fixed, n_kf)) + list(range(n_fixed))  # local (newest block) first
    poses = [poses[i] for i in order]
    # points in front of the middle camera (or of several anchors)
    Xw, z = _scene_points(rng, poses[:n_local], n_points, anchors)
    rig_c = camera_rig(rig) if rig else None
    if span is None:
        obs_list = _observe(rng, poses, Xw, Rcb, tcb, noise, outlier_frac, stereo_frac, rig_c)
        obs_arr = np.array(obs_list, dtype=np.float64)
    else:
        obs_arr = _observe_span(rng, poses, Xw, _scene_points.anchor, span, n_local, Rcb, tcb, noise, outlier_frac,
                                stereo_frac)
    # keep points with >= 2 observations and at least one local observer; renumber
    keep = _keep_points(obs_arr, n_points, n_local)
    remap = -np.ones(n_points, int)
    remap[keep] = np.arange(keep.sum())
    obs_arr = obs_arr[keep[obs_arr[:, 1].astype(int)]]
    obs = np.zeros(len(obs_arr), LBA_OBS_DTYPE)
    obs["kf"] = obs_arr[:, 0].astype(np.int32)
    obs["mp"] = remap[obs_arr[:, 1].astype(int)]
    obs["u"], obs["v"], obs["ur"], obs["inv_sigma2"] = obs_arr[:, 2], obs_arr[:, 3], obs_arr[:, 4], obs_arr[:, 5]
    obs = obs[np.argsort(obs["mp"], kind="stable")]
    Xw = Xw[keep]
    kfs = np.zeros(n_kf, LBA_KEYFRAME_DTYPE)
    truth_p, truth_q = [], []
    for k, (Rk, pk) in enumerate(poses):
        q = _R_to_quat(Rk)
        truth_p.append(pk), truth_q.append(q)
        is_fixed = k >= n_local or (first_fixed and k == 0)
        kfs[k]["fixed"] = int(is_fixed)
        if is_fixed:
            kfs[k]["nav"]["p"], kfs[k]["nav"]["q"] = pk, q
        else:
            kfs[k]["nav"]["p"] = pk + rng.normal(0, 1, 3) / np.sqrt(3) * pert_t
            kfs[k]["nav"]["q"] = quat_mul(q, quat_from_rotvec(rng.normal(0, 1, 3) / np.sqrt(3) *
                                                              np.deg2rad(pert_r_deg)))
    pts = (Xw + rng.normal(0, pert_x, Xw.shape)).astype(np.float32)
    params = np.zeros(1, LBA_PARAMS_DTYPE)
    params[0]["Rcb"], params[0]["tcb"] = Rcb.reshape(-1), tcb
    params[0]["fx"], params[0]["fy"], params[0]["cx"], params[0]["cy"], params[0]["bf"] = FX, FY, CX, CY, BF
    params[0]["its0"], params[0]["its1"] = 5, 10
    gt = dict(p=np.array(truth_p), q=np.array(truth_q), X=Xw)
    if rig_c is not None:  # keep the camera array alive with the truth record
        params[0]["n_cams"], params[0]["cams"] = len(rig_c[0]), rig_c[0].ctypes.data
        gt["cams"] = rig_c[0]
    return params, kfs, pts, obs, gt


def make_lba_enc(seed, truth, pairs, dt=0.5, noise=1.0):
    """vieo_lba_enc (LBA_ENC_DTYPE[1]) + its edge array for the key-frame pairs (i = previous, j) of a window whose
    TRUE poses are truth["p"], truth["q"]: wheel-odometry pre-integrations consistent with them up to sensor noise.
    Keep both returned arrays alive while the record is in use."""
    from .ba_types import LBA_ENC_DTYPE, LBA_ENC_EDGE_DTYPE
    rng = np.random.default_rng(seed + 31337)
    edges = np.zeros(len(pairs), LBA_ENC_EDGE_DTYPE)
    for t, (i, j) in enumerate(pairs):
        e = enc_between(rng, truth["p"][i], quat_to_R(truth["q"][i]), truth["p"][j], quat_to_R(truth["q"][j]), noise, dt)
        edges[t]["kf_i"], edges[t]["kf_j"], edges[t]["enc"] = i, j, e[0]["enc"]
    enc = np.zeros(1, LBA_ENC_DTYPE)
    enc[0]["n_edges"], enc[0]["edges"] = len(pairs), edges.ctypes.data
    enc[0]["qRbe"], enc[0]["pbe"] = _R_to_quat(ENC_RBE), ENC_PBE
    return enc, edges


def lba_enc_pairs(n_local, n_kf, with_prev=True):
    """(previous, current) key-frame pairs of make_lba_problem's window: local key frames are stored oldest first,
    the key frame before the window is the last (newest) fixed one."""
    pairs = [(k - 1, k) for k in range(1, n_local)]
    if with_prev and n_kf > n_local:
        pairs.insert(0, (n_kf - 1, 0))
    return pairs


# ----------------------------------------------------------------------------------------------
def imu_forward(rng, pi, Ri, vi, dt_kf, bg, ba, omega_sigma=0.08, acc_sigma=0.3, noise_scale=1.0):
    """One key-frame interval forward in time: constant body rate and world acceleration, sampled at
    IMU_FREQ.  State j is defined so that the NOISE-FREE discrete pre-integration has zero residual;
    the returned measurement carries sensor noise and was integrated with the biases (bg, ba)
    removed.  returns (pj, Rj, vj, Preintegrator)."""
    omega = rng.normal(0, omega_sigma, 3)
    a_w = rng.normal(0, acc_sigma, 3)
    n_imu = int(round(dt_kf * IMU_FREQ))
    h = dt_kf / n_imu
    ts = np.arange(n_imu + 1) * h
    Rrel = [so3_exp(omega * t) for t in ts]
    acc_true = [(Ri @ Rrel[k]).T @ (a_w - GRAVITY) for k in range(n_imu + 1)]
    clean = Preintegrator()
    for k in range(n_imu):
        clean.update(omega, (acc_true[k] + acc_true[k + 1]) / 2, h)
    Rj = Ri @ clean.R
    vj = vi + GRAVITY * dt_kf + Ri @ clean.v
    pj = pi + vi * dt_kf + GRAVITY * dt_kf ** 2 / 2 + Ri @ clean.p
    sg = IMU_SIGMA[0] * np.sqrt(IMU_FREQ) * noise_scale
    sa = IMU_SIGMA[1] * np.sqrt(IMU_FREQ) * noise_scale
    gyr = [omega + bg + rng.normal(0, sg, 3) for _ in range(n_imu + 1)]
    acc = [acc_true[k] + ba + rng.normal(0, sa, 3) for k in range(n_imu + 1)]
    meas = Preintegrator()
    for k in range(n_imu):
        meas.update((gyr[k] + gyr[k + 1]) / 2 - bg, (acc[k] + acc[k + 1]) / 2 - ba, h)
    return pj, Rj, vj, meas


_PVR_TO_PRV = np.r_[0:3, 6:9, 3:6]  # Sigma order (p, v, Phi) -> (p, Phi, v)


def make_lba_vio_problem(seed, n_local=10, n_fixed=5, n_points=1500, outlier_frac=0.03, stereo_frac=0.7,
                         noise=1.0, pert_t=0.01, pert_r_deg=0.3, pert_v=0.03, pert_x=0.02, dt_kf=0.5,
                         first_fixed=False, imu_noise=1.0, with_prev=True, rig=None, anchors=1, span=None, enc=False):
    """Seeded visual-inertial local-BA window: a chain prev-local -> n_local key frames
    integrated forward with consistent IMU pre-integrations, n_fixed older covisible key frames,
    points 2-12 m ahead.  Key-frame order: local (oldest..newest), prev-local (fixed, full nav state),
    other fixed.  returns (params[1], kfs, points f32, close u8, obs sorted by mp, imu edges, truth)."""
    from .ba_types import (LBA_IMU_EDGE_DTYPE, LBA_KEYFRAME_DTYPE, LBA_OBS_DTYPE, LBA_VIO_PARAMS_DTYPE)
    rng = np.random.default_rng(seed)
    Tcb = np.linalg.inv(EUROC_TBC)
    Rcb, tcb = Tcb[:3, :3], Tcb[:3, 3]
    bg, ba = rng.normal(0, 0.01, 3), rng.normal(0, 0.05, 3)
    # chain: index 0 = prev-local, 1..n_local = local window
    R = quat_to_R(quat_from_rotvec(rng.normal(0, 0.5, 3)))
    p = rng.uniform(-2, 2, 3)
    v = rng.normal(0, 0.25, 3)
    chain = [(R, p, v)]
    meas = []
    for k in range(n_local):
        p, R, v, m = imu_forward(rng, chain[-1][1], chain[-1][0], chain[-1][2], dt_kf, bg, ba,
                                 noise_scale=imu_noise)
        chain.append((R, p, v))
        meas.append(m)
    # older fixed key frames: extrapolate backwards on a smooth path
    R0, p0, v0 = chain[0]
    older = []
    for k in range(1, n_fixed):
        t = -dt_kf * k
        older.append((R0 @ so3_exp(rng.normal(0, 0.03, 3) * k), p0 + v0 * t + rng.normal(0, 0.02, 3), np.zeros(3)))
    n_prev = 1 if with_prev else 0
    local = chain[1:]
    fixed = ([chain[0]] if with_prev else []) + older
    poses = local + fixed
    n_kf = len(poses)
    Xw, z = _scene_points(rng, local, n_points, anchors)
    rig_c = camera_rig(rig) if rig else None
    if span is None:
        obs_list = _observe(rng, poses, Xw, Rcb, tcb, noise, outlier_frac, stereo_frac, rig_c)
        obs_arr = np.array(obs_list, dtype=np.float64)
    else:
        obs_arr = _observe_span(rng, poses, Xw, _scene_points.anchor, span, n_local, Rcb, tcb, noise, outlier_frac,
                                stereo_frac)
    keep = _keep_points(obs_arr, n_points, n_local)
    remap = -np.ones(n_points, int)
    remap[keep] = np.arange(keep.sum())
    obs_arr = obs_arr[keep[obs_arr[:, 1].astype(int)]]
    obs = np.zeros(len(obs_arr), LBA_OBS_DTYPE)
    obs["kf"] = obs_arr[:, 0].astype(np.int32)
    obs["mp"] = remap[obs_arr[:, 1].astype(int)]
    obs["u"], obs["v"], obs["ur"], obs["inv_sigma2"] = obs_arr[:, 2], obs_arr[:, 3], obs_arr[:, 4], obs_arr[:, 5]
    obs = obs[np.argsort(obs["mp"], kind="stable")]
    Xw = Xw[keep]
    close = (z[keep] < 8.0).astype(np.uint8)  # track_depth_ < thresh_depth_close
    kfs = np.zeros(n_kf, LBA_KEYFRAME_DTYPE)
    tp, tq, tv = [], [], []
    for k, (Rk, pk, vk) in enumerate(poses):
        q = _R_to_quat(Rk)
        tp.append(pk), tq.append(q), tv.append(vk)
        is_fixed = k >= n_local or (first_fixed and k == 0)
        kfs[k]["fixed"] = int(is_fixed)
        _nav(kfs[k]["nav"], pk, q, vk, bg, ba)
        if not is_fixed:
            kfs[k]["nav"]["p"] = pk + rng.normal(0, 1, 3) / np.sqrt(3) * pert_t
            kfs[k]["nav"]["q"] = quat_mul(q, quat_from_rotvec(rng.normal(0, 1, 3) / np.sqrt(3) *
                                                              np.deg2rad(pert_r_deg)))
            kfs[k]["nav"]["v"] = vk + rng.normal(0, pert_v, 3)
    imu = np.zeros(n_local - (0 if with_prev else 1), LBA_IMU_EDGE_DTYPE)
    t = 0
    for k in range(n_local):
        if k == 0 and not with_prev:
            continue
        imu[t]["kf_i"] = (n_local if k == 0 else k - 1)
        imu[t]["kf_j"] = k
        imu[t]["dt_kf"] = dt_kf
        fill_imu(imu[t]["imu"], meas[k])
        S = meas[k].Sigma[np.ix_(_PVR_TO_PRV, _PVR_TO_PRV)]
        imu[t]["imu"]["Sigma"] = S.reshape(-1)
        if enc:  # wheel-odometry pre-integration of the same pair (EncPreIntegrator: delta Phi, delta p in the
            #       encoder frame), consistent with the true poses
            (Ri, pi_, _), (Rj, pj_, _) = (chain[0] if k == 0 else local[k - 1]), local[k]
            Reb = ENC_RBE.T
            dR = Reb @ Ri.T @ Rj @ ENC_RBE
            dp = Reb @ (Ri.T @ (pj_ - pi_) - ENC_PBE + Ri.T @ Rj @ ENC_PBE)
            sphi, sp = 2e-3, 5e-3
            imu[t]["enc"]["dt"] = dt_kf
            imu[t]["enc"]["delx"][:3] = so3_log_np(dR) + rng.normal(0, sphi, 3)
            imu[t]["enc"]["delx"][3:] = dp + rng.normal(0, sp, 3)
            imu[t]["enc"]["Sigma"] = np.diag([sphi ** 2] * 3 + [sp ** 2] * 3).reshape(-1)
        t += 1
    pts = (Xw + rng.normal(0, pert_x, Xw.shape)).astype(np.float32)
    params = np.zeros(1, LBA_VIO_PARAMS_DTYPE)
    b = params[0]["base"]
    b["Rcb"], b["tcb"] = Rcb.reshape(-1), tcb
    b["fx"], b["fy"], b["cx"], b["cy"], b["bf"] = FX, FY, CX, CY, BF
    b["its0"], b["its1"] = 4, 6
    params[0]["gw"] = GRAVITY
    params[0]["inv_sigma_bg2"] = 1.0 / IMU_SIGMA[2] ** 2
    params[0]["inv_sigma_ba2"] = 1.0 / IMU_SIGMA[3] ** 2
    params[0]["lambda_init"] = 1.0
    params[0]["qRbe"], params[0]["pbe"] = _R_to_quat(ENC_RBE), ENC_PBE
    gt = dict(p=np.array(tp), q=np.array(tq), v=np.array(tv), X=Xw, bg=bg, ba=ba, n_local=n_local)
    if rig_c is not None:
        b["n_cams"], b["cams"] = len(rig_c[0]), rig_c[0].ctypes.data
        gt["cams"] = rig_c[0]
    return params, kfs, pts, close, obs, imu, gt


ENC_RBE = so3_exp(np.array([0.02, -0.03, 0.5]))


# ---- distorted multi-camera rigs (a20: Radtan / KB8 models, per-observation camera) --------------
def camera_rig(name, with_tcr=False, n_cams=None):
    """(CAMERA_DTYPE array, (width, height)) of a rig as the BA edges see it: EdgeReproject::SetParams
    already applied (Rcb = Rccr * Rcrb, tcb = Rccr * tcrb + tcr).  'radtan': the two EuRoC cameras
    (k1 k2 p1 p2); 'kb8': four TUM-VI-like fisheye cameras (k1..k4).
    with_tcr: also the list of 4x4 Tcr (reference camera -> camera i).
    n_cams: None = the rig as described; 2 or 4 = that many cameras (the radtan rig grows two side cameras, the kb8
    rig keeps its first two)."""
    from .ba_types import CAMERA_DTYPE
    Tcb = np.linalg.inv(EUROC_TBC)
    Rcrb, tcrb = Tcb[:3, :3], Tcb[:3, 3]
    if name == "radtan":
        intr = [(458.654, 457.296, 367.215, 248.375, [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]),
                (457.587, 456.134, 379.999, 255.238, [-0.28368365, 0.07451284, -0.00010473, -3.55590700e-05])]
        Tcr = [np.eye(4), np.eye(4)]
        Tcr[1][:3, :3] = so3_exp(np.array([0.002, -0.012, 0.001]))
        Tcr[1][:3, 3] = [-0.110, 0.0004, -0.0008]
        if n_cams == 4:
            intr += [(458.1, 457.0, 370.2, 250.1, intr[0][4]), (457.9, 456.6, 375.3, 252.7, intr[1][4])]
            for rv, t in (((0.0, 0.30, 0.0), (0.05, 0.0, -0.02)), ((0.0, -0.30, 0.0), (-0.16, 0.0, -0.02))):
                T = np.eye(4)
                T[:3, :3] = so3_exp(np.array(rv, float))
                T[:3, 3] = t
                Tcr.append(T)
        model, num_k, size = 1, 2, (752, 480)
    elif name == "kb8":
        k = [0.0034823894, 0.0007150348, -0.0020532361, 0.00020293673]
        intr = [(190.978, 190.973, 254.93, 256.90, k), (190.442, 190.434, 252.60, 254.92, k),
                (191.1, 191.0, 255.5, 255.9, k), (190.7, 190.8, 256.2, 254.1, k)]
        Tcr = [np.eye(4) for _ in range(4)]
        for i, (rv, t) in enumerate([((0, 0, 0), (0, 0, 0)), ((0.003, 0.01, -0.002), (-0.101, 0.001, -0.001)),
                                     ((0.0, 0.35, 0.0), (0.05, 0.0, -0.02)), ((0.0, -0.35, 0.0), (-0.15, 0.0, -0.02))]):
            Tcr[i][:3, :3] = so3_exp(np.array(rv, float))
            Tcr[i][:3, 3] = t
        if n_cams == 2:
            intr, Tcr = intr[:2], Tcr[:2]
        model, num_k, size = 2, 0, (512, 512)
    else:
        raise ValueError(name)
    cams = np.zeros(len(intr), CAMERA_DTYPE)
    for i, (fx, fy, cx, cy, d) in enumerate(intr):
        c = cams[i]
        c["model"], c["num_k"] = model, num_k
        c["fx"], c["fy"], c["cx"], c["cy"] = fx, fy, cx, cy
        c["dist"][:len(d)] = d
        Rccr, tcr = Tcr[i][:3, :3], Tcr[i][:3, 3]
        c["Rcb"] = (Rccr @ Rcrb).reshape(-1)
        c["tcb"] = Rccr @ tcrb + tcr
    if with_tcr:
        return cams, size, Tcr
    return cams, size


def project_camera(cam, Pc):
    """float64 python restatement of the three Project functions (image point only)."""
    x, y, z = Pc
    fx, fy, cx, cy = float(cam["fx"]), float(cam["fy"]), float(cam["cx"]), float(cam["cy"])
    d = cam["dist"].astype(np.float64)
    if cam["model"] == 1:
        nk = int(cam["num_k"])
        xn, yn = x / z, y / z
        r2 = xn * xn + yn * yn
        fd = 1 + sum(d[i] * r2 ** (i + 1) for i in range(nk))
        p1, p2 = d[nk], d[nk + 1]
        xd = xn * fd + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
        yd = yn * fd + 2 * p2 * xn * yn + p1 * (r2 + 2 * yn * yn)
        return fx * xd + cx, fy * yd + cy
    if cam["model"] == 2:
        r = np.hypot(x, y)
        if r > 1e-5:
            th = np.arctan2(r, z)
            t2 = th * th
            thd = th * (1 + t2 * (d[0] + t2 * (d[1] + t2 * (d[2] + t2 * d[3]))))
            return fx * x * thd / r + cx, fy * y * thd / r + cy
    return fx * x / z + cx, fy * y / z + cy
