"""Trajectory files and the absolute-trajectory-error evaluation of the reference (the wire
formats either side of the hot path, needed to compare a replay with the reference's own result tables).

Writers follow the line formats of
  System::SaveKeyFrameTrajectoryNavState / SaveTrajectoryNavState   reference src/System.cc:34-67, 860-935
      "stamp tx ty tz qx qy qz qw vx vy vz bgx bgy bgz bax bay baz", std::fixed, setprecision(9), bias = b + db
  System::SaveTrajectoryTUM                                          src/System.cc:780-859
      "stamp(6 digits) tx ty tz qx qy qz qw [bg ba]" with float32 pose values at setprecision(9)
  System::SaveKeyFrameTrajectoryTUM                                  src/System.cc:937-988   (7 digits)
  the optional gravity alignment R_Iw of both TUM writers            src/System.cc:834-846

The evaluation restates the TUM RGB-D benchmark tools the reference shells out to
(Examples/RunEuRoC/EvaluateEuRoC_Evaluate.sh:38-57: `evaluate_ate.py gt est --offset O` and the `_scale` variant;
the scripts themselves are third-party and absent from /root/reference): timestamp association (greedy, closest
first, |t_gt - (t_est + offset)| < 0.02 s), Horn's closed-form rigid alignment by SVD, translational error
statistics.  Host-side Python like the reference's tooling; nothing here runs on the GPU."""
import numpy as np


def _fixed(x, digits):
    return "%.*f" % (digits, x)


def write_trajectory_navstate(path, stamps, navs):
    """navs: NAVSTATE_DTYPE records (p, q = (w, x, y, z), v, bg, ba, dbg, dba).  One line per state."""
    with open(path, "w") as f:
        for t, n in zip(stamps, navs):
            q = n["q"]
            vals = [t, *n["p"], q[1], q[2], q[3], q[0], *n["v"], *(n["bg"] + n["dbg"]), *(n["ba"] + n["dba"])]
            f.write(" ".join(_fixed(float(v), 9) for v in vals) + "\n")


def gravity_alignment(gw):
    """R_Iw of the TUM writers (System.cc:834-846): rotates the world so that gravity points along +z."""
    gwn = np.asarray(gw, float) / np.linalg.norm(gw)
    gI = np.array([0.0, 0.0, 1.0])
    a = np.cross(gI, gwn)
    na = np.linalg.norm(a)
    if na == 0:
        return np.eye(3)
    vhat = a / na
    theta = np.arccos(np.clip(gI @ gwn, -1.0, 1.0))
    K = np.array([[0, -vhat[2], vhat[1]], [vhat[2], 0, -vhat[0]], [-vhat[1], vhat[0], 0]])
    RwI = np.eye(3) + np.sin(theta) * K + (1 - np.cos(theta)) * K @ K
    return RwI.T


def write_trajectory_tum(path, stamps, twc, q_xyzw, bg=None, ba=None, keyframes=False):
    """TUM format.  twc / q_xyzw: camera centres and Rwc quaternions (x, y, z, w); stored as float32 like the
    reference's cv::Mat.  keyframes = True: the key-frame writer's 7 digits; bg / ba: the `imu_info` columns."""
    twc = np.asarray(twc, np.float32)
    q = np.asarray(q_xyzw, np.float32)
    dig = 7 if keyframes else 9
    with open(path, "w") as f:
        for i, t in enumerate(stamps):
            cols = [_fixed(float(t), 6)] + [_fixed(float(v), dig) for v in (*twc[i], *q[i])]
            if bg is not None and not keyframes:
                cols += [_fixed(float(v), dig) for v in (*bg[i], *ba[i])]
            f.write(" ".join(cols) + "\n")


def read_trajectory(path):
    """{stamp: [floats]} of a TUM / NavState / ground-truth file: '#' comments skipped, commas or blanks split
    the columns (the benchmark tools' read_file_list)."""
    out = {}
    with open(path) as f:
        for line in f.read().replace(",", " ").replace("\t", " ").split("\n"):
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            cols = [c for c in line.split(" ") if c]
            if len(cols) > 1:
                out[float(cols[0])] = [float(c) for c in cols[1:]]
    return out


def associate(first_stamps, second_stamps, offset=0.0, max_difference=0.02):
    """Pairs (a, b) with |a - (b + offset)| < max_difference, every stamp used once, closest pairs first
    (associate.py of the benchmark tools); returned sorted by a."""
    a = np.sort(np.asarray(list(first_stamps), float))
    b = np.sort(np.asarray(list(second_stamps), float))
    cand = []
    lo = np.searchsorted(b + offset, a - max_difference, "left")
    hi = np.searchsorted(b + offset, a + max_difference, "right")
    for i in range(len(a)):
        for j in range(lo[i], hi[i]):
            d = abs(a[i] - (b[j] + offset))
            if d < max_difference:
                cand.append((d, a[i], b[j]))
    cand.sort()
    used_a, used_b, matches = set(), set(), []
    for d, x, y in cand:
        if x not in used_a and y not in used_b:
            used_a.add(x)
            used_b.add(y)
            matches.append((x, y))
    matches.sort()
    return matches


def align(model, data, with_scale=False):
    """Horn's method: rot, trans (and scale) that map `model` (3 x n, the estimate) onto `data` (ground truth).
    returns (rot, trans, trans_error[n]) or, with_scale, (rot, trans_s, trans_error_s, trans, trans_error, s) as
    evaluate_ate_scale.py does (the rotation is estimated without the scale, then s = <data, R model> / |model|^2)."""
    model = np.asarray(model, float)
    data = np.asarray(data, float)
    mm, dm = model.mean(1, keepdims=True), data.mean(1, keepdims=True)
    mz, dz = model - mm, data - dm
    W = mz @ dz.T  # sum of outer(model_c, data_c)
    U, _, Vh = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    rot = U @ S @ Vh
    trans = dm - rot @ mm
    err = np.sqrt(np.sum((rot @ model + trans - data) ** 2, 0))
    if not with_scale:
        return rot, trans[:, 0], err
    rm = rot @ mz
    s = float(np.sum(dz * rm) / np.sum(mz * mz))
    trans_s = dm - s * rot @ mm
    err_s = np.sqrt(np.sum((s * rot @ model + trans_s - data) ** 2, 0))
    return rot, trans_s[:, 0], err_s, trans[:, 0], err, s


def _stats(e, pairs):
    return {"compared_pose_pairs": int(pairs), "rmse": float(np.sqrt(np.dot(e, e) / len(e))), "mean": float(np.mean(e)),
            "median": float(np.median(e)), "std": float(np.std(e)), "min": float(np.min(e)), "max": float(np.max(e))}


def ate(gt, est, offset=0.0, max_difference=0.02, scale=1.0, with_scale=False):
    """gt / est: {stamp: [tx, ty, tz, ...]} (read_trajectory).  Absolute translational error statistics in the
    units of gt; `scale` multiplies the estimate before the alignment (--scale), with_scale adds the statistics after
    the closed-form scale (evaluate_ate_scale.py: 'rmse' is then the scaled one, 'rmse_no_scale' the plain one)."""
    matches = associate(gt.keys(), est.keys(), offset, max_difference)
    if len(matches) < 2:
        raise ValueError("Couldn't find matching timestamp pairs between groundtruth and estimated trajectory")
    first = np.array([gt[a][:3] for a, _ in matches], float).T
    second = np.array([est[b][:3] for _, b in matches], float).T * scale
    if not with_scale:
        rot, trans, err = align(second, first)
        return dict(_stats(err, len(matches)), rot=rot, trans=trans)
    rot, trans_s, err_s, trans, err, s = align(second, first, True)
    out = dict(_stats(err_s, len(matches)), rot=rot, trans=trans_s, scale=s)
    out["rmse_no_scale"] = _stats(err, len(matches))["rmse"]
    return out


def evaluate_ate(gt_file, est_file, **kw):
    return ate(read_trajectory(gt_file), read_trajectory(est_file), **kw)
