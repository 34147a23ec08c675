"""GPU parity: visual-inertial PoseOptimization (15- and 30-dim systems, marginal prior) vs the
CPU oracle; 1e-4 on SE(3) (BASELINE.json), identical inlier decisions."""
import numpy as np
import pytest

from vieo_slam_amd import synth_ba
from vieo_slam_amd._lib import DeviceBuffer, check, lib
from vieo_slam_amd.ba_types import VIO_FRAME_DTYPE, VIO_RESULT_DTYPE

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _cmp(oracle, F, obs, marg_rtol=1e-5):
    from vieo_slam_amd.optimizer import Optimizer
    o, oo = oracle.pose_optimization_vio(F, obs)
    h, ho = Optimizer.PoseOptimizationVIO(F, obs)
    dt, dr = synth_ba.pose_error(o["base"]["nav"], h["base"]["nav"])
    assert dt < TOL and dr < TOL, (dt, dr)
    assert np.linalg.norm(o["base"]["nav"]["v"] - h["base"]["nav"]["v"]) < 1e-4
    assert np.linalg.norm(o["base"]["nav"]["dbg"] - h["base"]["nav"]["dbg"]) < 1e-6
    assert np.linalg.norm(o["base"]["nav"]["dba"] - h["base"]["nav"]["dba"]) < 1e-5
    assert o["base"]["status"] == h["base"]["status"]
    assert o["base"]["n_inliers"] == h["base"]["n_inliers"]
    assert np.array_equal(oo, ho)
    assert o["has_marg"] == h["has_marg"]
    if o["has_marg"]:
        Ho, Hh = o["H_marg"].reshape(15, 15), h["H_marg"].reshape(15, 15)
        assert np.allclose(Ho, Hh, rtol=marg_rtol, atol=marg_rtol * np.abs(Ho).max())
    return o, h


@pytest.mark.parametrize("seed,n,marg", [(0, 300, False), (1, 300, True), (2, 80, True),
                                         (3, 900, True), (4, 25, False)])
def test_vio_fixed_last_parity(oracle, seed, n, marg):
    F, obs, gt = synth_ba.make_vio_problem(seed, n_obs=n, compute_marg=marg)
    _cmp(oracle, F, obs)  # (LM iteration counts may differ by a few near convergence: the
    # 0.1 % improvement stop rule is evaluated on sums accumulated in a different order)


# (n: the four-wavefront instance keeps the visual edges on two wavefronts up to 768 edges, on all four beyond)
@pytest.mark.parametrize("seed,n", [(40, 300), (42, 300), (44, 300), (46, 767), (48, 769), (50, 1500)])
def test_vio_free_last_with_prior_parity(oracle, seed, n):
    F0, obs0, _ = synth_ba.make_vio_problem(seed, compute_marg=True)
    r0, _ = oracle.pose_optimization_vio(F0, obs0)
    F1, obs1, _ = synth_ba.make_vio_problem(seed + 1, n_obs=n, compute_marg=True)
    nav_last = F1[0]["nav_last"].copy()
    nav_prior = nav_last.copy()
    nav_last["p"] += 0.004
    nav_last["v"] += 0.015
    F1b, _, _ = synth_ba.make_vio_problem(seed + 1, n_obs=n, compute_marg=True,
                                          prior=(nav_prior, r0["H_marg"].reshape(15, 15), nav_last))
    _cmp(oracle, F1b, obs1, marg_rtol=1e-4)


def test_vio_edge_cases_parity(oracle):
    from vieo_slam_amd.optimizer import Optimizer
    F, obs, _ = synth_ba.make_vio_problem(50, n_obs=2)
    h, _ = Optimizer.PoseOptimizationVIO(F, obs)
    assert h["base"]["n_inliers"] == 0 and h["base"]["status"] == 1
    F, obs, _ = synth_ba.make_vio_problem(51)  # no IMU measurement -> estimate reset per round
    F[0]["imu"]["dt"] = 0
    _cmp(oracle, F, obs)
    F, obs, _ = synth_ba.make_vio_problem(52, n_obs=40, outlier_frac=0.5, compute_marg=True)
    _cmp(oracle, F, obs)  # rescue pass
    F, obs, _ = synth_ba.make_vio_problem(53, n_obs=6, outlier_frac=0.0)  # < 10 edges: one round
    _cmp(oracle, F, obs)


def test_vio_batch_device(oracle):
    B = 12
    frames = np.zeros(B, VIO_FRAME_DTYPE)
    all_obs, begin = [], 0
    for i in range(B):
        F, obs, _ = synth_ba.make_vio_problem(200 + i, n_obs=120 + 41 * i, compute_marg=(i % 2 == 0))
        frames[i] = F[0]
        frames[i]["base"]["obs_begin"] = begin
        begin += len(obs)
        all_obs.append(obs)
    obs = np.concatenate(all_obs)
    dF, dO = DeviceBuffer(frames.nbytes), DeviceBuffer(obs.nbytes)
    dU, dR = DeviceBuffer(len(obs)), DeviceBuffer(B * VIO_RESULT_DTYPE.itemsize)
    dF.upload(frames)
    dO.upload(obs)
    check(lib().vieo_pose_optimization_vio_batch_device(dF.ptr, B, dO.ptr, dU.ptr, dR.ptr, None))
    check(lib().vieo_device_synchronize())
    res = dR.download(VIO_RESULT_DTYPE, (B,))
    outl = dU.download(np.uint8, (len(obs),))
    for i in range(B):
        b, n = frames[i]["base"]["obs_begin"], frames[i]["base"]["n_obs"]
        o, oo = oracle.pose_optimization_vio(frames[i:i + 1], obs)
        dt, dr = synth_ba.pose_error(o["base"]["nav"], res[i]["base"]["nav"])
        assert dt < TOL and dr < TOL
        assert res[i]["base"]["n_inliers"] == o["base"]["n_inliers"]
        assert np.array_equal(oo[b:b + n], outl[b:b + n])


def test_vio_large_batch_one_wavefront_per_frame(oracle):
    """More than 256 frames take the one-wavefront-per-frame kernel: every frame must still equal the
    oracle (distinct problems incl. free last state + prior, marginalisation, rescue pass, few edges)."""
    protos = []
    for i in range(10):
        kw = dict(n_obs=60 + 97 * i, compute_marg=(i % 2 == 0))
        if i == 7:
            kw.update(n_obs=40, outlier_frac=0.5)
        if i == 9:
            kw.update(n_obs=6, outlier_frac=0.0)
        if i % 3 == 1:  # last state free, with a marginal prior (30-dim system)
            F0, obs0, _ = synth_ba.make_vio_problem(400 + i, compute_marg=True)
            r0, _ = oracle.pose_optimization_vio(F0, obs0)
            F1, obs1, _ = synth_ba.make_vio_problem(300 + i, **kw)
            nav_last = F1[0]["nav_last"].copy()
            nav_prior = nav_last.copy()
            nav_last["p"] += 0.004
            nav_last["v"] += 0.015
            kw["prior"] = (nav_prior, r0["H_marg"].reshape(15, 15), nav_last)
        protos.append(synth_ba.make_vio_problem(300 + i, **kw)[:2])
    B = 300
    frames = np.zeros(B, VIO_FRAME_DTYPE)
    all_obs, begin = [], 0
    for i in range(B):
        F, obs = protos[i % len(protos)]
        frames[i] = F[0]
        frames[i]["base"]["obs_begin"] = begin
        begin += len(obs)
        all_obs.append(obs)
    obs = np.concatenate(all_obs)
    dF, dO = DeviceBuffer(frames.nbytes), DeviceBuffer(obs.nbytes)
    dU, dR = DeviceBuffer(len(obs)), DeviceBuffer(B * VIO_RESULT_DTYPE.itemsize)
    dF.upload(frames)
    dO.upload(obs)
    check(lib().vieo_pose_optimization_vio_batch_device(dF.ptr, B, dO.ptr, dU.ptr, dR.ptr, None))
    check(lib().vieo_device_synchronize())
    res = dR.download(VIO_RESULT_DTYPE, (B,))
    outl = dU.download(np.uint8, (len(obs),))
    ref = [oracle.pose_optimization_vio(F, o) for F, o in protos]
    for i in range(B):
        b, n = frames[i]["base"]["obs_begin"], frames[i]["base"]["n_obs"]
        o, oo = ref[i % len(protos)]
        dt, dr = synth_ba.pose_error(o["base"]["nav"], res[i]["base"]["nav"])
        assert dt < TOL and dr < TOL, (i, dt, dr)
        assert res[i]["base"]["n_inliers"] == o["base"]["n_inliers"] and res[i]["base"]["status"] == o["base"]["status"]
        assert np.array_equal(oo[:n], outl[b:b + n])
        assert res[i]["has_marg"] == o["has_marg"]
        if o["has_marg"]:
            Ho = o["H_marg"].reshape(15, 15)
            assert np.allclose(Ho, res[i]["H_marg"].reshape(15, 15), rtol=1e-4, atol=1e-4 * np.abs(Ho).max())


@pytest.mark.parametrize("name,seed,marg", [("radtan", 70, True), ("kb8", 71, True), ("kb8", 72, False)])
def test_vio_camera_rig_parity(oracle, name, seed, marg):
    """a20: the visual edges of a distorted multi-camera rig (camera index in bits 8..11 of obs.flags)."""
    rig = synth_ba.camera_rig(name)
    F, obs, gt = synth_ba.make_vio_problem(seed, n_obs=300, compute_marg=marg, rig=rig)
    assert F[0]["base"]["n_cams"] == len(rig[0])
    o, h = _cmp(oracle, F, obs)
    assert o["base"]["n_inliers"] > 200


@pytest.mark.parametrize("name,seed,n,free_last", [("kb8", 73, 1500, False), ("radtan", 74, 3200, False),
                                                   ("kb8", 75, 2400, True), ("kb8", 76, 699, False),
                                                   ("kb8", 77, 700, False), ("radtan", 78, 9000, False)])
def test_vio_rig_replicas_parity(oracle, name, seed, n, free_last):
    """A rig frame of a small call is optimised by 16 replica workgroups that share the visual edges (from 700 edges on;
    below, one of them takes the frame alone): against the oracle, and against the one-workgroup form of the same
    kernel (vieo_pose_set_replicas(0)) -- only the association order of the visual sums differs."""
    from vieo_slam_amd.optimizer import Optimizer
    rig = synth_ba.camera_rig(name)
    F, obs, _ = synth_ba.make_vio_problem(seed, n_obs=n, compute_marg=True, rig=rig)
    if free_last:
        F0, obs0, _ = synth_ba.make_vio_problem(seed + 100, compute_marg=True)
        r0, _ = oracle.pose_optimization_vio(F0, obs0)
        nav_last = F[0]["nav_last"].copy()
        nav_prior = nav_last.copy()
        nav_last["p"] += 0.004
        nav_last["v"] += 0.015
        F, _, _ = synth_ba.make_vio_problem(seed, n_obs=n, compute_marg=True, rig=rig,
                                            prior=(nav_prior, r0["H_marg"].reshape(15, 15), nav_last))
    L = lib()
    was = L.vieo_pose_set_replicas(1)
    try:
        o, h8 = _cmp(oracle, F, obs, marg_rtol=1e-4 if free_last else 1e-5)
        _, ho8 = Optimizer.PoseOptimizationVIO(F, obs)
        L.vieo_pose_set_replicas(0)
        h1, ho1 = Optimizer.PoseOptimizationVIO(F, obs)
    finally:
        L.vieo_pose_set_replicas(was)
    assert h8["base"]["status"] == 0 and h1["base"]["status"] == 0
    dt, dr = synth_ba.pose_error(h1["base"]["nav"], h8["base"]["nav"])
    assert dt < 1e-9 and dr < 1e-7, (dt, dr)  # (dr: the angle metric resolves 3e-8)
    assert np.array_equal(ho1, ho8) and h1["base"]["n_inliers"] == h8["base"]["n_inliers"]
    H1, H8 = h1["H_marg"].reshape(15, 15), h8["H_marg"].reshape(15, 15)
    assert np.allclose(H1, H8, rtol=1e-7, atol=1e-7 * np.abs(H1).max())


def test_vio_rig_replicas_repeated_launches_and_small_batches(oracle):
    """The exchange records are reused launch after launch (tags carry the launch number), a call of up to 4 rig frames
    is replicated frame by frame, a larger one is not: all give the results of single calls."""
    from vieo_slam_amd.optimizer import Optimizer
    rig = synth_ba.camera_rig("kb8")
    probs = [synth_ba.make_vio_problem(300 + i, n_obs=n, compute_marg=(i % 2 == 0), rig=rig)
             for i, n in enumerate((900, 2000, 650, 1400, 800, 1200))]
    single = [Optimizer.PoseOptimizationVIO(F, obs) for F, obs, _ in probs]
    for rep in range(3):  # the same record set again and again
        for (F, obs, _), (h0, o0) in zip(probs, single):
            h, o = Optimizer.PoseOptimizationVIO(F, obs)
            assert np.array_equal(o, o0) and h.tobytes() == h0.tobytes()  # a fixed order of sums: the same bits every time
    dC = DeviceBuffer(rig[0].nbytes)
    dC.upload(rig[0])
    for B in (2, 4, 6):  # 2 and 4: replicated frame by frame; 6: one workgroup per frame; a rectified frame among them
        frames = np.zeros(B, VIO_FRAME_DTYPE)
        all_obs, begin = [], 0
        plain = synth_ba.make_vio_problem(340, n_obs=400)
        for i in range(B):
            F, obs, _ = plain if i == 1 else probs[i]
            frames[i] = F[0]
            frames[i]["base"]["obs_begin"] = begin
            frames[i]["base"]["cams"] = 0 if i == 1 else dC.ptr
            all_obs.append(obs)
            begin += len(obs)
        obs_cat = np.concatenate(all_obs)
        dF, dO = DeviceBuffer(frames.nbytes), DeviceBuffer(obs_cat.nbytes)
        dU, dR = DeviceBuffer(len(obs_cat)), DeviceBuffer(B * VIO_RESULT_DTYPE.itemsize)
        dF.upload(frames)
        dO.upload(obs_cat)
        check(lib().vieo_pose_optimization_vio_batch_device(dF.ptr, B, dO.ptr, dU.ptr, dR.ptr, None))
        check(lib().vieo_device_synchronize())
        res = dR.download(VIO_RESULT_DTYPE, (B,))
        outl = dU.download(np.uint8, (len(obs_cat),))
        for i in range(B):
            h0, o0 = Optimizer.PoseOptimizationVIO(*plain[:2]) if i == 1 else single[i]
            assert res[i]["base"]["status"] == 0
            dt, dr = synth_ba.pose_error(h0["base"]["nav"], res[i]["base"]["nav"])
            assert dt < 1e-9 and dr < 1e-7, (B, i, dt, dr)  # (dr: the angle metric resolves 3e-8)
            b0 = int(frames[i]["base"]["obs_begin"])
            assert np.array_equal(outl[b0:b0 + len(o0)], o0)


def test_vio_rig_replica_that_never_arrives_degrades_to_one_workgroup(oracle, monkeypatch):
    """A replica workgroup that does not become resident (VIEO_POSE_REPLICA_DROP: replica 5 returns at once, as on a device
    another process fills) must cost ONE time-out, not one per exchange, and must not cost the frame: the waiting replicas
    poison their slots and finish, the host entry repeats the optimisation on one workgroup and returns its result."""
    import time
    from vieo_slam_amd.optimizer import Optimizer
    rig = synth_ba.camera_rig("kb8")
    F, obs, _ = synth_ba.make_vio_problem(410, n_obs=1800, compute_marg=True, rig=rig)
    L = lib()
    was = L.vieo_pose_set_replicas(0)
    try:
        h1, o1 = Optimizer.PoseOptimizationVIO(F, obs)
        L.vieo_pose_set_replicas(1)
        h16, o16 = Optimizer.PoseOptimizationVIO(F, obs)
        monkeypatch.setenv("VIEO_POSE_REPLICA_DROP", "1")
        t0 = time.perf_counter()
        hd, od = Optimizer.PoseOptimizationVIO(F, obs)
        dt_fail = time.perf_counter() - t0
        monkeypatch.delenv("VIEO_POSE_REPLICA_DROP")
        h16b, o16b = Optimizer.PoseOptimizationVIO(F, obs)  # the records are usable again (the poison carries the launch number)
    finally:
        L.vieo_pose_set_replicas(was)
    assert hd["base"]["status"] == 0 and hd.tobytes() == h1.tobytes() and np.array_equal(od, o1)
    assert h16b.tobytes() == h16.tobytes() and np.array_equal(o16b, o16)
    assert dt_fail < 5.0, dt_fail  # one time-out (~0.5 s), not ~36
    print("replica drop: the frame came back after %.2f s on one workgroup" % dt_fail)


@pytest.mark.parametrize("seed,n,kw", [(80, 300, dict(compute_marg=True)), (81, 60, dict(compute_marg=True, noise=2.0)),
                                       (82, 200, dict(imu=False, compute_marg=True)),
                                       (83, 40, dict(outlier_frac=0.5, compute_marg=True)),
                                       (84, 7, dict(outlier_frac=0.0))])
def test_vio_encoder_edge_parity(oracle, seed, n, kw):
    """a16: EdgeEncNavStatePVR between the last frame and the current one (Optimizer.h:345-363), also inside the
    marginal prior (FillCovInv :195-204); without an IMU measurement it is the only odometry edge."""
    F, obs, gt = synth_ba.make_vio_problem(seed, n_obs=n, enc=True, **kw)
    o, h = _cmp(oracle, F, obs)
    F0, _, _ = synth_ba.make_vio_problem(seed, n_obs=n, **kw)
    from vieo_slam_amd.optimizer import Optimizer
    h0, _ = Optimizer.PoseOptimizationVIO(F0, obs)
    assert not np.array_equal(h0["base"]["nav"]["p"], h["base"]["nav"]["p"])  # the edge is in the system


@pytest.mark.parametrize("rigname", [None, "kb8"])
def test_vio_encoder_edge_free_last_parity(oracle, rigname):
    """30-dim system: the encoder edge couples the two PVR vertices and enters B, C and E of the Schur complement."""
    kw = dict(rig=synth_ba.camera_rig(rigname)) if rigname else {}
    F0, obs0, _ = synth_ba.make_vio_problem(90, compute_marg=True)
    r0, _ = oracle.pose_optimization_vio(F0, obs0)
    F1, obs1, _ = synth_ba.make_vio_problem(91, compute_marg=True, **kw)
    nav_last = F1[0]["nav_last"].copy()
    nav_prior = nav_last.copy()
    nav_last["p"] += 0.004
    nav_last["v"] += 0.015
    F1b, _, gt = synth_ba.make_vio_problem(91, compute_marg=True, enc=True,
                                           prior=(nav_prior, r0["H_marg"].reshape(15, 15), nav_last), **kw)
    _cmp(oracle, F1b, obs1, marg_rtol=1e-4)


def test_vio_encoder_mode_and_mixed_device_batch(oracle):
    """vieo_pose_set_encoder_mode: a batch that mixes frames with and without an encoder measurement (and, here,
    rectified and rig frames) needs AUTO -- four kernel instances, each skips the others' frames; under NONE a
    frame that carries a measurement is refused loudly (status VIEO_E_INVALID), never optimised without it."""
    from vieo_slam_amd.ba_types import POSE_ENC_DTYPE
    rig = synth_ba.camera_rig("radtan")
    B = 8
    frames = np.zeros(B, VIO_FRAME_DTYPE)
    encs = np.zeros(B, POSE_ENC_DTYPE)
    keep, all_obs, begin = [], [], 0
    for i in range(B):
        F, obs, gt = synth_ba.make_vio_problem(500 + i, n_obs=100 + 30 * i, compute_marg=(i % 3 == 0), enc=(i % 2 == 0),
                                               **(dict(rig=rig) if i % 4 >= 2 else {}))
        keep.append((F, obs, gt))
        frames[i] = F[0]
        frames[i]["base"]["obs_begin"] = begin
        begin += len(obs)
        all_obs.append(obs)
        if i % 2 == 0:
            encs[i] = gt["enc"][0]
    obs = np.concatenate(all_obs)
    dC, dE = DeviceBuffer(rig[0].nbytes), DeviceBuffer(encs.nbytes)
    dC.upload(rig[0])
    dE.upload(encs)
    for i in range(B):
        frames[i]["base"]["enc"] = dE.ptr + i * POSE_ENC_DTYPE.itemsize if i % 2 == 0 else 0
        frames[i]["base"]["cams"] = dC.ptr if i % 4 >= 2 else 0
    dF, dO = DeviceBuffer(frames.nbytes), DeviceBuffer(obs.nbytes)
    dU, dR = DeviceBuffer(len(obs)), DeviceBuffer(B * VIO_RESULT_DTYPE.itemsize)
    dF.upload(frames)
    dO.upload(obs)
    try:
        for mode in (0, 1, 2):
            check(lib().vieo_pose_set_encoder_mode(mode))
            dR.upload(np.zeros(B, VIO_RESULT_DTYPE))
            check(lib().vieo_pose_optimization_vio_batch_device(dF.ptr, B, dO.ptr, dU.ptr, dR.ptr, None))
            check(lib().vieo_device_synchronize())
            res = dR.download(VIO_RESULT_DTYPE, (B,))
            outl = dU.download(np.uint8, (len(obs),))
            for i in range(B):
                has = i % 2 == 0
                if (mode == 1 and has) or (mode == 2 and not has):
                    assert res[i]["base"]["status"] < 0 and res[i]["base"]["n_inliers"] == 0, (mode, i)
                    continue
                F, ob, _ = keep[i]
                o, oo = oracle.pose_optimization_vio(F, ob)
                b, n = frames[i]["base"]["obs_begin"], frames[i]["base"]["n_obs"]
                dt, dr = synth_ba.pose_error(o["base"]["nav"], res[i]["base"]["nav"])
                assert dt < TOL and dr < TOL, (mode, i, dt, dr)
                assert res[i]["base"]["n_inliers"] == o["base"]["n_inliers"] and res[i]["base"]["status"] == 0
                assert np.array_equal(oo, outl[b:b + n])
                if o["has_marg"]:
                    Ho = o["H_marg"].reshape(15, 15)
                    assert np.allclose(Ho, res[i]["H_marg"].reshape(15, 15), rtol=1e-4, atol=1e-4 * np.abs(Ho).max())
        assert lib().vieo_pose_set_encoder_mode(3) != 0
    finally:
        check(lib().vieo_pose_set_encoder_mode(0))
