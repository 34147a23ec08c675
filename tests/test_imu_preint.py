"""IMUPreIntegratorBase::PreIntegration + update (reference src/Odom/OdomPreIntegrator.h:226-506):
oracle known-answer tests against the numpy restatement of update() in synth_ba (CPU) and HIP-vs-oracle parity."""
import numpy as np
import pytest

from vieo_slam_amd import synth_ba
from vieo_slam_amd.imu import IMU_NOISE_DTYPE, IMU_SAMPLE_DTYPE


def _noise(fixed=1):
    N = np.zeros(1, IMU_NOISE_DTYPE)
    f = synth_ba.IMU_FREQ if fixed else 1.0
    N[0]["sigma_g"] = (np.eye(3) * synth_ba.IMU_SIGMA[0] ** 2 * f).reshape(-1)
    N[0]["sigma_a"] = (np.eye(3) * synth_ba.IMU_SIGMA[1] ** 2 * f).reshape(-1)
    N[0]["freq_ref"], N[0]["dt_cov_noise_fixed"] = synth_ba.IMU_FREQ, fixed
    return N


def _samples(rng, t0, n, h=0.005, jitter=0.0):
    s = np.zeros(n, IMU_SAMPLE_DTYPE)
    s["t"] = t0 + np.arange(n) * h + rng.uniform(-jitter, jitter, n)
    s["w"] = rng.normal(0, 0.4, (1, 3)) + rng.normal(0, 0.05, (n, 3))
    s["a"] = rng.normal(0, 2.0, (1, 3)) + np.array([0, 0, 9.8]) + rng.normal(0, 0.2, (n, 3))
    return s


def test_oracle_matches_numpy_update_on_aligned_samples(oracle):
    rng = np.random.default_rng(1)
    s = _samples(rng, 10.0, 21)
    bg, ba = rng.normal(0, 0.01, 3), rng.normal(0, 0.05, 3)
    out, prv, st = oracle.imu_preintegrate(_noise(), [s], [s["t"][0]], [s["t"][-1]], [bg], [ba])
    P = synth_ba.Preintegrator()
    for k in range(20):
        P.update((s["w"][k] + s["w"][k + 1]) / 2 - bg, (s["a"][k] + s["a"][k + 1]) / 2 - ba, s["t"][k + 1] - s["t"][k])
    o = out[0]
    assert st[0] == 0 and abs(o["dt"] - 0.1) < 1e-12
    assert np.allclose(o["Rij"].reshape(3, 3), P.R, atol=1e-13) and np.allclose(o["vij"], P.v, atol=1e-13)
    assert np.allclose(o["pij"], P.p, atol=1e-14)
    for k, ref in (("JgR", P.JgR), ("Jgv", P.Jgv), ("Jav", P.Jav), ("Jgp", P.Jgp), ("Jap", P.Jap)):
        assert np.allclose(o[k].reshape(3, 3), ref, atol=1e-13), k
    S = o["Sigma"].reshape(9, 9)
    assert np.allclose(S, P.Sigma, rtol=1e-10, atol=1e-20)
    # mSigmaijPRV is the same covariance with the v and Phi blocks exchanged
    perm = [0, 1, 2, 6, 7, 8, 3, 4, 5]
    assert np.allclose(prv[0], S[np.ix_(perm, perm)], rtol=1e-10, atol=1e-20)
    assert np.allclose(S, S.T, rtol=1e-9) and np.linalg.eigvalsh(S).min() > 0


def test_oracle_partial_intervals_and_status(oracle):
    rng = np.random.default_rng(2)
    s = _samples(rng, 5.0, 30)
    s["w"], s["a"] = s["w"][0], s["a"][0]  # constant measurements: interpolation cannot change them
    bg = ba = np.zeros(3)
    ti, tj = s["t"][3] + 0.002, s["t"][24] + 0.001  # both ends inside a sample interval
    out, prv, st = oracle.imu_preintegrate(_noise(), [s], [ti], [tj], [bg], [ba])
    assert st[0] == 0 and abs(out[0]["dt"] - (tj - ti)) < 1e-12
    # constant rate: R = Exp(w * T) whatever the splitting
    assert np.allclose(out[0]["Rij"].reshape(3, 3), synth_ba.so3_exp(s["w"][0] * (tj - ti)), atol=1e-12)
    # the window may start before the first and end after the last sample
    out2, _, st2 = oracle.imu_preintegrate(_noise(), [s[5:15]], [s["t"][5] - 0.003], [s["t"][14] + 0.004], [bg], [ba])
    assert st2[0] == 0 and abs(out2[0]["dt"] - (s["t"][14] + 0.004 - s["t"][5] + 0.003)) < 1e-12
    # statuses: no samples / a 2 s hole
    hole = s.copy()
    hole["t"][15:] += 2.0
    _, _, st3 = oracle.imu_preintegrate(_noise(), [s[:0], hole], [5.0, hole["t"][0]], [5.1, hole["t"][-1]], [bg] * 2,
                                        [ba] * 2)
    assert st3.tolist() == [1, 2]
    # noise model: per-sample 1/dt scaling equals the fixed one at the reference rate
    a, _, _ = oracle.imu_preintegrate(_noise(1), [s], [s["t"][0]], [s["t"][-1]], [bg], [ba])
    b, _, _ = oracle.imu_preintegrate(_noise(0), [s], [s["t"][0]], [s["t"][-1]], [bg], [ba])
    assert np.allclose(a[0]["Sigma"], b[0]["Sigma"], rtol=1e-6)


def test_oracle_backward_order(oracle):
    """timeStampi > timeStampj (map reuse, OdomPreIntegrator.h:241-262): the samples are walked backwards with
    negative steps.  Constant measurements have closed forms whatever the splitting: R = Exp(w T), and with w = 0
    v = a T, p = a T^2 / 2 (T = tj - ti < 0); the forward run over the same span mirrors them."""
    rng = np.random.default_rng(4)
    s = _samples(rng, 5.0, 30)
    bg = ba = np.zeros(3)
    for ti, tj in ((s["t"][24] + 0.001, s["t"][3] + 0.002),   # both ends inside a sample interval
                   (s["t"][29] + 0.004, s["t"][0] - 0.003),   # beyond both ends of the list
                   (s["t"][20], s["t"][5]),                   # on samples
                   (s["t"][7] + 0.0031, s["t"][7] + 0.0012)):  # inside one interval
        c = s.copy()
        c["w"], c["a"] = s["w"][0], 0.0
        out, _, st = oracle.imu_preintegrate(_noise(), [c], [ti], [tj], [bg], [ba])
        T = tj - ti
        assert st[0] == 0 and T < 0 and abs(out[0]["dt"] - T) < 1e-12, (ti, tj, out[0]["dt"])
        assert np.allclose(out[0]["Rij"].reshape(3, 3), synth_ba.so3_exp(s["w"][0] * T), atol=1e-12)
        c["w"], c["a"] = 0.0, s["a"][0]
        out, _, st = oracle.imu_preintegrate(_noise(), [c], [ti], [tj], [bg], [ba])
        assert np.allclose(out[0]["vij"], s["a"][0] * T, atol=1e-12) and np.allclose(out[0]["pij"], s["a"][0] * T * T / 2, atol=1e-12)
    # varying measurements: forward over [a, b] then backward over [b, a] compose to (nearly) the identity rotation
    a, b = s["t"][2] + 0.001, s["t"][26] + 0.003
    f, _, _ = oracle.imu_preintegrate(_noise(), [s], [a], [b], [bg], [ba])
    r, _, st = oracle.imu_preintegrate(_noise(), [s], [b], [a], [bg], [ba])
    assert st[0] == 0 and abs(f[0]["dt"] + r[0]["dt"]) < 1e-12
    assert np.abs(f[0]["Rij"].reshape(3, 3) @ r[0]["Rij"].reshape(3, 3) - np.eye(3)).max() < 1e-5
    # a hole is still refused
    hole = s.copy()
    hole["t"][15:] += 2.0
    _, _, st = oracle.imu_preintegrate(_noise(), [hole], [hole["t"][-1]], [hole["t"][0]], [bg], [ba])
    assert st[0] == 2


@pytest.mark.gpu
def test_preintegration_parity(oracle):
    from vieo_slam_amd.imu import imu_preintegrate
    rng = np.random.default_rng(3)
    lists, ti, tj = [], [], []
    for k in range(700):
        n = int(rng.integers(0, 40)) if k else 0
        s = _samples(rng, 100.0 + k, n, jitter=0.001)
        if n > 3 and k % 50 == 7:
            s["t"][n // 2:] += 2.0
        lists.append(s)
        if n:
            ti.append(s["t"][0] + rng.uniform(-0.004, 0.012))
            tj.append(s["t"][-1] + rng.uniform(-0.012, 0.004))
            if k % 3 == 2:  # backward order (map reuse)
                ti[-1], tj[-1] = tj[-1], ti[-1]
        else:
            ti.append(0.0), tj.append(1.0)
    bg, ba = rng.normal(0, 0.01, (700, 3)), rng.normal(0, 0.05, (700, 3))
    for fixed in (1, 0):
        o, op, os_ = oracle.imu_preintegrate(_noise(fixed), lists, ti, tj, bg, ba)
        h, hp, hs = imu_preintegrate(_noise(fixed), lists, ti, tj, bg, ba)
        assert np.array_equal(os_, hs) and set(os_.tolist()) >= {0, 1, 2}
        for k in ("dt", "Rij", "vij", "pij", "JgR", "Jgv", "Jav", "Jgp", "Jap"):
            assert np.allclose(o[k], h[k], rtol=1e-11, atol=1e-13), k
        # covariances: relative to each matrix's own scale (tiny off-diagonal entries are differences of products)
        for a, b in ((o["Sigma"], h["Sigma"]), (op.reshape(700, 81), hp.reshape(700, 81))):
            # (a zero-length sub-step with the 1/dt noise model gives inf * 0 = NaN in the reference as well)
            assert np.array_equal(np.isnan(a), np.isnan(b))
            a, b = np.nan_to_num(a), np.nan_to_num(b)
            scale = np.abs(a).max(1, keepdims=True) + 1e-300
            assert (np.abs(a - b) / scale).max() < 1e-10


@pytest.mark.gpu
def test_wave_and_lane_instantiations_agree_bitwise():
    """A call with fewer than 1024 intervals gives every interval a wavefront (the 9 x 9 products spread over the lanes),
    a larger one a lane: every entry is summed by one lane in the same order, so the two must agree bit for bit."""
    from vieo_slam_amd.imu import imu_preintegrate
    rng = np.random.default_rng(11)
    lists, ti, tj = [], [], []
    for k in range(1100):
        n = int(rng.integers(2, 120))
        s = _samples(rng, 50.0 + k, n, jitter=0.001)
        lists.append(s)
        ti.append(s["t"][0] + rng.uniform(-0.004, 0.012))
        tj.append(s["t"][-1] + rng.uniform(-0.012, 0.004))
        if k % 4 == 3:
            ti[-1], tj[-1] = tj[-1], ti[-1]
    bg, ba = rng.normal(0, 0.01, (1100, 3)), rng.normal(0, 0.05, (1100, 3))
    for fixed in (1, 0):
        big, bigp, bigs = imu_preintegrate(_noise(fixed), lists, ti, tj, bg, ba)               # lane per interval
        sm, smp, sms = imu_preintegrate(_noise(fixed), lists[:40], ti[:40], tj[:40], bg[:40], ba[:40])  # wavefront each
        assert np.array_equal(bigs[:40], sms)
        assert big[:40].tobytes() == sm.tobytes()
        assert np.asarray(bigp)[:40].tobytes() == np.asarray(smp).tobytes()
