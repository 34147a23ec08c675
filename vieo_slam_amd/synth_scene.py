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
    """Plane z = 0 of the world, textured; cameras look at it from above.
    relief = (layout seed, n): n thin textured sheets (0.7 - 1.6 m squares) float 0.25 - 1.4 m above the plane around
    the origin, so that depth is not planar and views occlude differently (a floating sheet is a consistent 3-D scene
    for every viewpoint); low_contrast: the texture's contrast falls to 15 - 45 % in smooth patches -- corners there
    have FAST strengths between minThFAST and iniThFAST, the cells that need the second threshold (ORBextractor.cc:
    752-762)."""

    def __init__(self, seed, relief=None, low_contrast=False, tex=None):
        self.tex = synth.synth_image_f32(seed, TEX_W, TEX_H, n_shapes=2600) if tex is None else tex
        if low_contrast:
            rng = np.random.default_rng(seed + 4242)
            g = rng.uniform(0, 1, (TEX_H // 192 + 2, TEX_W // 192 + 2))
            c = np.where(g < 0.3, rng.uniform(0.15, 0.45, g.shape), 1.0)  # 30 % of the 1.5 m patches are dull
            yy, xx = np.mgrid[0:TEX_H, 0:TEX_W]
            fy, fx = yy / 192.0, xx / 192.0
            y0, x0 = fy.astype(int), fx.astype(int)
            wy, wx = (fy - y0).astype(np.float32), (fx - x0).astype(np.float32)
            cm = (c[y0, x0] * (1 - wx) * (1 - wy) + c[y0, x0 + 1] * wx * (1 - wy) + c[y0 + 1, x0] * (1 - wx) * wy +
                  c[y0 + 1, x0 + 1] * wx * wy).astype(np.float32)
            m = np.float32(self.tex.mean())
            self.tex = (m + cm * (self.tex - m)).astype(np.float32)
        self.Tcb = np.linalg.inv(synth_ba.EUROC_TBC)
        self.Tbc = synth_ba.EUROC_TBC
        self.sheets = []
        if relief:
            rng = np.random.default_rng(1000 + relief[0])
            for _ in range(relief[1]):
                half = rng.uniform(0.35, 0.8, 2)
                self.sheets.append(dict(c=rng.uniform(-2.6, 2.6, 2), half=half, z=rng.uniform(0.25, 1.4),
                                        toff=rng.uniform(-600, 600, 2)))

    def with_relief(self, relief):
        """the same texture with another sheet layout (the texture is the expensive part)"""
        s = Scene.__new__(Scene)
        s.tex, s.Tcb, s.Tbc, s.sheets = self.tex, self.Tcb, self.Tbc, []
        s.__init__(0, relief=relief, tex=self.tex)
        return s

    def _sample(self, P, off=(0.0, 0.0)):
        tx = P[..., 0] / TEXEL + TEX_W / 2 + off[0]
        ty = P[..., 1] / TEXEL + TEX_H / 2 + off[1]
        x0 = np.clip(np.floor(tx).astype(np.int64), 0, TEX_W - 2)
        y0 = np.clip(np.floor(ty).astype(np.int64), 0, TEX_H - 2)
        fx = np.clip(tx - x0, 0, 1).astype(np.float32)
        fy = np.clip(ty - y0, 0, 1).astype(np.float32)
        T = self.tex
        return (T[y0, x0] * (1 - fx) * (1 - fy) + T[y0, x0 + 1] * fx * (1 - fy) +
                T[y0 + 1, x0] * (1 - fx) * fy + T[y0 + 1, x0 + 1] * fx * fy)

    def render(self, Rwc, twc, noise_seed):
        """uint8 image of the pinhole camera at (Rwc, twc), plus per-pixel camera depth."""
        v, u = np.mgrid[0:H, 0:W].astype(np.float64)
        d_c = np.stack([(u - CX) / FX, (v - CY) / FY, np.ones_like(u)], -1)
        d_w = d_c @ Rwc.T
        lam = -twc[2] / d_w[..., 2]  # plane z = 0
        img = self._sample(twc + lam[..., None] * d_w)
        Rcw = Rwc.T
        for sh in self.sheets:  # nearest surface along the ray: only inside the sheet's projected bounding box
            cx, cy, hx, hy, z = sh["c"][0], sh["c"][1], sh["half"][0], sh["half"][1], sh["z"]
            cor = np.array([[cx - hx, cy - hy, z], [cx + hx, cy - hy, z], [cx + hx, cy + hy, z], [cx - hx, cy + hy, z]])
            pc = (cor - twc) @ Rcw.T
            if np.any(pc[:, 2] < 0.2):
                continue
            uu, vv = FX * pc[:, 0] / pc[:, 2] + CX, FY * pc[:, 1] / pc[:, 2] + CY
            u0, u1 = int(max(0, np.floor(uu.min()))), int(min(W, np.ceil(uu.max()) + 1))
            v0, v1 = int(max(0, np.floor(vv.min()))), int(min(H, np.ceil(vv.max()) + 1))
            if u0 >= u1 or v0 >= v1:
                continue
            dw = d_w[v0:v1, u0:u1]
            l2 = (z - twc[2]) / dw[..., 2]
            P2 = twc + l2[..., None] * dw
            hit = (np.abs(P2[..., 0] - cx) <= hx) & (np.abs(P2[..., 1] - cy) <= hy) & (l2 > 0) & (l2 < lam[v0:v1, u0:u1])
            if not hit.any():
                continue
            col = self._sample(P2, sh["toff"])
            sub = img[v0:v1, u0:u1]
            sub[hit] = col[hit]
            lam[v0:v1, u0:u1][hit] = l2[hit]
        return synth.quantise(img.astype(np.float32), noise_seed), lam

    def surface_points(self, rng, n):
        """n points on the visible surfaces (plane or sheet tops) around the origin, with their upward normal."""
        P = np.zeros((n, 3))
        P[:, :2] = rng.uniform(-4.0, 4.0, (n, 2))
        for sh in self.sheets:  # later sheets win where they overlap; a point under a sheet is simply occluded from above
            ins = (np.abs(P[:, 0] - sh["c"][0]) <= sh["half"][0]) & (np.abs(P[:, 1] - sh["c"][1]) <= sh["half"][1])
            P[ins, 2] = np.maximum(P[ins, 2], sh["z"])
        return P

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


# ---- frames of a distorted camera rig (BASELINE configs[3] / [4]: Radtan EuRoC stereo, KB8 TUM-VI stereo) --------
def unproject_pixels(cam, W_, H_):
    """unit-plane rays (x, y, 1)[H, W, 3] of every pixel centre of a Radtan / KB8 camera, by vectorised Newton /
    fixed-point inversion of synth_ba.project_camera (float64; generator code, not the reference's UnProject).
    KB8 pixels beyond ~88 degrees get z <= 0 (no ray)."""
    v, u = np.mgrid[0:H_, 0:W_].astype(np.float64)
    fx, fy, cx, cy = float(cam["fx"]), float(cam["fy"]), float(cam["cx"]), float(cam["cy"])
    mx, my = (u - cx) / fx, (v - cy) / fy
    d = cam["dist"].astype(np.float64)
    if cam["model"] == 1:  # radtan: fixed-point iteration on the normalised coordinates
        nk = int(cam["num_k"])
        p1, p2 = d[nk], d[nk + 1]
        x, y = mx.copy(), my.copy()
        for _ in range(30):
            r2 = x * x + y * y
            fd = 1 + sum(d[i] * r2 ** (i + 1) for i in range(nk))
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = 2 * p2 * x * y + p1 * (r2 + 2 * y * y)
            x, y = (mx - dx) / fd, (my - dy) / fd
        return np.stack([x, y, np.ones_like(x)], -1)
    if cam["model"] == 2:  # kb8: Newton on theta
        thd = np.hypot(mx, my)
        th = thd.copy()
        for _ in range(12):
            t2 = th * th
            f = th * (1 + t2 * (d[0] + t2 * (d[1] + t2 * (d[2] + t2 * d[3])))) - thd
            df = 1 + t2 * (3 * d[0] + t2 * (5 * d[1] + t2 * (7 * d[2] + t2 * 9 * d[3])))
            th = th - f / df
        s = np.where(thd > 1e-9, np.sin(th) / np.maximum(thd, 1e-9), 1.0)
        return np.stack([mx * s, my * s, np.cos(th)], -1)  # not normalised to z = 1: direction only
    return np.stack([mx, my, np.ones_like(mx)], -1)


class RigScene(Scene):
    """The textured plane seen through the distorted cameras of synth_ba.camera_rig(name)."""

    def __init__(self, seed, rig="radtan", n_cams=2):
        super().__init__(seed)
        self.cams, (self.W, self.H), Tcr = synth_ba.camera_rig(rig, with_tcr=True, n_cams=n_cams)
        self.Tcr = Tcr
        self.rays = [unproject_pixels(c, self.W, self.H) for c in self.cams]
        # body <- reference camera is EUROC_TBC like the rectified scene

    def render_cam(self, c, Rwcr, twcr, noise_seed):
        """uint8 image of camera c of the rig whose reference camera sits at (Rwcr, twcr)."""
        Trc = np.linalg.inv(self.Tcr[c])
        Rwc = Rwcr @ Trc[:3, :3]
        twc = twcr + Rwcr @ Trc[:3, 3]
        d_w = self.rays[c] @ Rwc.T
        hit = d_w[..., 2] < -1e-3  # looking down at z = 0 from above
        lam = np.where(hit, -twc[2] / np.where(hit, d_w[..., 2], -1.0), 0.0)
        P = twc + lam[..., None] * d_w
        tx = P[..., 0] / TEXEL + TEX_W / 2
        ty = P[..., 1] / TEXEL + TEX_H / 2
        inside = hit & (tx >= 0) & (tx < TEX_W - 1) & (ty >= 0) & (ty < TEX_H - 1) & (self.rays[c][..., 2] > 0.05)
        x0 = np.clip(np.floor(tx).astype(np.int64), 0, TEX_W - 2)
        y0 = np.clip(np.floor(ty).astype(np.int64), 0, TEX_H - 2)
        fx = np.clip(tx - x0, 0, 1).astype(np.float32)
        fy = np.clip(ty - y0, 0, 1).astype(np.float32)
        T = self.tex
        img = (T[y0, x0] * (1 - fx) * (1 - fy) + T[y0, x0 + 1] * fx * (1 - fy) +
               T[y0 + 1, x0] * (1 - fx) * fy + T[y0 + 1, x0 + 1] * fx * fy)
        img = np.where(inside, img, 40.0).astype(np.float32)  # outside the plane / the fisheye circle: flat dark
        return synth.quantise(img, noise_seed)

    def frame(self, Rwb, pwb, noise_seed):
        """images of all cameras for the body pose; returns (images, Rwcr, twcr)."""
        Rwc = Rwb @ self.Tbc[:3, :3]
        twc = pwb + Rwb @ self.Tbc[:3, 3]
        return [self.render_cam(c, Rwc, twc, noise_seed + c) for c in range(len(self.cams))], Rwc, twc


def make_rig_tracking_case(seed, scene, dt_frame=0.05, height=(2.4, 3.0)):
    """Two consecutive rig frames with IMU between them (the rig counterpart of make_tracking_case)."""
    rng = np.random.default_rng(seed + 4242)
    Rj, pj = look_down_pose(rng)
    pj = pj.copy()
    pj[2] += rng.uniform(*height) - 4.5
    pi, Ri, vi, vj, bg, ba, meas = synth_ba.imu_motion(rng, pj, Rj, dt_frame)
    im0, Rwc0, twc0 = scene.frame(Ri, pi, 10 * seed)
    im1, Rwc1, twc1 = scene.frame(Rj, pj, 10 * seed + 5)
    F = np.zeros(1, VIO_FRAME_DTYPE)
    f = F[0]
    Tcb = scene.Tcb
    b = f["base"]
    b["Rcb"] = Tcb[:3, :3].reshape(-1)
    b["tcb"] = Tcb[:3, 3]
    c0 = scene.cams[0]
    b["fx"], b["fy"], b["cx"], b["cy"], b["bf"] = c0["fx"], c0["fy"], c0["cx"], c0["cy"], 0.11 * float(c0["fx"])
    synth_ba._nav(b["nav"], pj, synth_ba._R_to_quat(Rj), vj, bg, ba)
    synth_ba._nav(f["nav_last"], pi, synth_ba._R_to_quat(Ri), vi, bg, ba)
    synth_ba.fill_imu(f["imu"], meas)
    f["gw"] = synth_ba.GRAVITY
    f["inv_sigma_bg2"] = 1.0 / synth_ba.IMU_SIGMA[2] ** 2
    f["inv_sigma_ba2"] = 1.0 / synth_ba.IMU_SIGMA[3] ** 2
    f["dt_frames"] = dt_frame
    f["th_depth"] = 35.0
    from .imu import IMU_SAMPLE_DTYPE
    ts, gyr, acc = meas.samples
    samples = np.zeros(len(ts), IMU_SAMPLE_DTYPE)
    samples["t"], samples["w"], samples["a"] = ts, gyr, acc
    return dict(images0=im0, images1=im1, pose0=(Ri, pi, Rwc0, twc0), pose1=(Rj, pj, Rwc1, twc1), vio=F,
                truth=dict(p=pj, q=synth_ba._R_to_quat(Rj), v=vj), imu_samples=samples, dt_frame=dt_frame)
