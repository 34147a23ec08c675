"""The one-call frame tracker behind the C-ABI (vieo_tracker_*, vieo_track_frame; csrc/tracker.hip): the chained frame
of replay.ChainedReplay issued from C++ -- one upload, the IMU pre-integration beside the extraction, the state
prediction and every piece of bookkeeping on the device, one synchronisation."""
import os
import subprocess

import numpy as np
import pytest

from vieo_slam_amd import replay, synth_ba

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tracker_structs_match_the_header(tmp_path):
    """numpy mirrors of vieo_tracker_params / vieo_track_input / vieo_track_output against the C header (gcc)."""
    from vieo_slam_amd import tracker as tk
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "vieo_hot.h"
int main(void) {
  printf("%zu %zu %zu ", sizeof(vieo_tracker_params), sizeof(vieo_track_input), sizeof(vieo_track_output));
  printf("%zu %zu %zu %zu %zu %zu ", sizeof(vieo_tracker_rig), offsetof(vieo_tracker_rig, cams), offsetof(vieo_tracker_rig, Tcr),
         offsetof(vieo_tracker_params, vision_only), offsetof(vieo_track_input, images), offsetof(vieo_track_output, group_p3d));
  printf("%zu %zu %zu %zu ", offsetof(vieo_tracker_params, Rcb), offsetof(vieo_tracker_params, noise),
         offsetof(vieo_track_input, nav_ref), offsetof(vieo_track_input, local_alias));
  printf("%zu %zu %zu %zu %zu %zu\n", offsetof(vieo_track_output, nav_pred), offsetof(vieo_track_output, imu),
         offsetof(vieo_track_output, first), offsetof(vieo_track_output, second), offsetof(vieo_track_output, ms_gpu),
         offsetof(vieo_track_output, local_track_depth));
  return 0;
}''')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    P, I, O = tk.TRACKER_PARAMS_DTYPE, tk.TRACK_INPUT_DTYPE, tk.TRACK_OUTPUT_DTYPE
    G = tk.TRACKER_RIG_DTYPE
    want = [P.itemsize, I.itemsize, O.itemsize, G.itemsize, G.fields["cams"][1], G.fields["Tcr"][1],
            P.fields["vision_only"][1], I.fields["images"][1], O.fields["group_p3d"][1], P.fields["Rcb"][1], P.fields["noise"][1], I.fields["nav_ref"][1],
            I.fields["local_alias"][1], O.fields["nav_pred"][1], O.fields["imu"][1], O.fields["first"][1],
            O.fields["second"][1], O.fields["ms_gpu"][1], O.fields["local_track_depth"][1]]
    assert got == want


@pytest.mark.gpu
def test_gpu_tracker_replay_equals_the_python_chain(oracle):
    """100 frames, 9 local BAs: the C++ chain against the oracle's stage-by-stage run (BASELINE configs[2]: ATE within
    1e-4) and against the Python-issued chain (same kernels; the state prediction moved from numpy to the device)."""
    from tests.replay_oracle import OracleStages
    from vieo_slam_amd.tracker import TrackerReplay
    n = 100
    seq = replay.Sequence(1, n)
    Ro = replay.Replay(seq, OracleStages(oracle))
    to = Ro.run(n)
    Rc = replay.ChainedReplay(seq, replay.HipStages())
    tc = Rc.run(n)
    Rt = TrackerReplay(seq, replay.HipStages())
    tt = Rt.run(n)
    Rt.close()
    assert len(tt) == n and Rt.stats["lba"] == Ro.stats["lba"] == 9 and Rt.stats["widened"] == 0
    ate = replay.ate_between(tt, to)
    dmax = np.linalg.norm(tt["p"] - to["p"], axis=1).max()
    assert ate <= 1e-4 and dmax <= 1e-4, (ate, dmax)
    rot = max(synth_ba.pose_error(tt[k], to[k])[1] for k in range(n))
    assert rot <= 1e-4, rot
    mc, mt = np.array(Rc.stats["n_matches"]), np.array(Rt.stats["n_matches"])
    ic, it = np.array(Rc.stats["n_inliers"]), np.array(Rt.stats["n_inliers"])
    assert (mc == mt).all(1).mean() > 0.97 and np.abs(mc - mt).max() <= 3, (mc - mt)
    assert (ic == it).mean() > 0.97 and np.abs(ic - it).max() <= 3, (ic - it)
    assert replay.ate_between(tt, tc) <= 1e-5
    ms = np.array(Rt.stats["ms_chain"])
    print("tracker replay: ATE vs oracle %.3e m, vs the Python chain %.3e m; per frame %.2f ms in the call "
          "(GPU %.2f), %.2f ms with the driver" % (ate, replay.ate_between(tt, tc), ms[:, 0].mean(), ms[:, 1].mean(),
                                                     np.mean(Rt.stats["ms_frames"])))


@pytest.mark.gpu
def test_gpu_tracker_wider_window_branch():
    """Fewer than 20 matches in the first search (Tracking.cc:301-309): the call re-runs the chain from the projection
    with 2 x th -- the decisions equal the stage-by-stage replay's, which takes the same branch on the host."""
    from vieo_slam_amd.tracker import TrackerReplay
    n = 24
    seq = replay.Sequence(3, n)
    Rh = replay.Replay(seq, replay.HipStages(), th_last=0.12)
    th = Rh.run(n)
    Rt = TrackerReplay(seq, replay.HipStages(), th_last=0.12)
    tt = Rt.run(n)
    Rt.close()
    assert Rt.stats["widened"] > 0
    mh, mt = np.array(Rh.stats["n_matches"]), np.array(Rt.stats["n_matches"])
    assert (mh == mt).all(1).mean() > 0.9 and np.abs(mh - mt).max() <= 3, (mh - mt)
    assert replay.ate_between(tt, th) <= 1e-5


@pytest.mark.gpu
def test_gpu_tracker_pinned_image_planes_and_local_version():
    """Images decoded straight into the tracker's pinned planes give the same frame as caller-owned buffers; an unchanged
    local_version skips the candidate upload and changes nothing."""
    from vieo_slam_amd.tracker import TrackerReplay
    seq = replay.Sequence(4, 14)
    Ra = TrackerReplay(seq, replay.HipStages())
    ta = Ra.run(14)
    Ra.close()
    Rb = TrackerReplay(seq, replay.HipStages())
    orig = Rb.seq.images

    def into_pinned(k):
        L, R = orig(k)
        Rb.trk.left[:], Rb.trk.right[:] = L, R
        return Rb.trk.left, Rb.trk.right
    Rb.initialise()
    for k in range(1, 14):
        Rb.seq.images = into_pinned if k > 1 else orig
        Rb.step(k)
    Rb.seq.images = orig
    tb = np.array(Rb.traj, ta.dtype)
    Rb.close()
    assert ta.tobytes() == tb.tobytes()


@pytest.mark.gpu
def test_gpu_cpp_replay_without_python(tmp_path):
    """examples/replay_main.cc: the sequential replay as a C++ program on the C-ABI (vieo_track_frame per frame, local BA
    per key frame, the map on the host in C++).  Same sequence, same calls as tracker.TrackerReplay -> the same
    trajectory up to the rounding of the host-side glue (numpy's BLAS product against a plain loop where new map points
    are unprojected)."""
    import json
    from tools.write_sequence import write_sequence
    from vieo_slam_amd.tracker import TrackerReplay
    exe = os.path.join(ROOT, "examples", "replay_main")
    assert os.path.exists(exe), "examples/replay_main is built by __graft_entry__.build()"
    n = 60
    seq = replay.Sequence(1, n)
    path, traj_path = str(tmp_path / "seq.vseq"), str(tmp_path / "traj.bin")
    write_sequence(path, 1, n, seq)
    line = subprocess.check_output([exe, path, traj_path, "--quiet"], timeout=600).decode().strip().splitlines()[-1]
    r = json.loads(line)
    from vieo_slam_amd.ba_types import NAVSTATE_DTYPE
    tc = np.fromfile(traj_path, NAVSTATE_DTYPE)
    Rt = TrackerReplay(seq, replay.HipStages())
    tt = Rt.run(n)
    Rt.close()
    assert len(tc) == len(tt) == n and r["frames"] == n - 1 and r["local_bas"] == Rt.stats["lba"] == 5
    assert r["key_frames"] == len(Rt.kfs) and r["map_points"] == len(Rt.mp_X)
    d = np.linalg.norm(tc["p"] - tt["p"], axis=1)
    assert d.max() <= 1e-6, d.max()
    rot = max(synth_ba.pose_error(tc[k], tt[k])[1] for k in range(n))
    assert rot <= 1e-6, rot
    assert r["max_err_vs_truth_m"] < 1.5e-2
    print("C++ replay: %.3f ms per frame (tracking call %.3f, its GPU part %.3f, local BA %.2f ms each); "
          "max |dp| against the Python driver %.2e m" % (r["ms_per_frame"], r["ms_track_call"], r["ms_track_gpu"],
                                                         r["ms_per_local_ba"], d.max()))


@pytest.mark.gpu
def test_gpu_cpp_replay_local_ba_beside_tracking(oracle, tmp_path):
    """LocalMapping beside Tracking (src/LocalMapping.cc:113-139): examples/replay_main --lba-lag 3 solves a key frame's
    local BA on its own host thread while the next frames are tracked and applies the write-back before the third frame
    after the key frame.  The same lag in the Python driver (solve at once, hold the result back) gives the same map for
    the same frames: the C++ run equals the Python tracker run, and both stay within 1e-4 of the ORACLE replay with
    that lag -- the trajectory the concurrent run is checked against."""
    import json
    from tests.replay_oracle import OracleStages
    from tools.write_sequence import write_sequence
    from vieo_slam_amd.ba_types import NAVSTATE_DTYPE
    from vieo_slam_amd.tracker import TrackerReplay
    exe = os.path.join(ROOT, "examples", "replay_main")
    n, lag = 60, 3
    seq = replay.Sequence(1, n)
    path, traj_path = str(tmp_path / "seq.vseq"), str(tmp_path / "traj.bin")
    write_sequence(path, 1, n, seq)
    line = subprocess.check_output([exe, path, traj_path, "--quiet", "--lba-lag", str(lag)], timeout=600).decode().strip().splitlines()[-1]
    r = json.loads(line)
    tc = np.fromfile(traj_path, NAVSTATE_DTYPE)
    Rt = TrackerReplay(seq, replay.HipStages(), lba_lag=lag)
    tt = Rt.run(n)
    Rt.close()
    Ro = replay.Replay(seq, OracleStages(oracle), lba_lag=lag)
    to = Ro.run(n)
    assert r["lba_lag"] == lag and r["local_bas"] == Rt.stats["lba"] == Ro.stats["lba"] == 5
    d = np.linalg.norm(tc["p"] - tt["p"], axis=1)
    assert d.max() <= 1e-6, d.max()
    ate = replay.ate_between(tc, to)
    assert ate <= 1e-4 and np.linalg.norm(tc["p"] - to["p"], axis=1).max() <= 1e-4, ate
    # the lag changes the run (the frames between a key frame and its write-back see the old map)
    R0 = TrackerReplay(seq, replay.HipStages())
    t0 = R0.run(n)
    R0.close()
    assert 1e-7 < np.linalg.norm(t0["p"] - tt["p"], axis=1).max() < 5e-3
    print("C++ replay, local BA beside tracking (lag %d): %.3f ms per frame (tracking call %.3f), ATE vs the oracle run %.2e m"
          % (lag, r["ms_per_frame"], r["ms_track_call"], ate))


@pytest.mark.gpu
def test_gpu_three_host_threads_concurrently(oracle):
    """the back end must be re-entrant across >= 3 host threads.  Tracking (vieo_track_frame per frame, with
    its key-frame local BAs), LocalMapping-style work (visual-inertial local BAs and a vision-only pose optimisation that
    needs the OTHER kernel instances: a distorted rig) and LoopClosing-style work (Fuse searches) run at the same
    time, each on its own thread and streams; every thread's results equal the ones it gets alone.  The kernel-instance
    modes are per call / per thread now, so the rig optimisation of thread 2 cannot be skipped by the tracker's choice."""
    import threading
    from tests.test_map_point import _fuse_case
    from vieo_slam_amd.map_point import fuse_search
    from vieo_slam_amd.optimizer import Optimizer
    from vieo_slam_amd.tracker import TrackerReplay
    n = 30
    seq = replay.Sequence(6, n)
    for k in range(n):
        seq.images(k)

    def track():
        R = TrackerReplay(seq, replay.HipStages())
        t = R.run(n)
        R.close()
        return t.tobytes()

    win = synth_ba.make_lba_vio_problem(61, n_local=8, n_fixed=4, n_points=900)[:6]
    fr, ob, _ = synth_ba.make_pose_problem(62, n_obs=300, rig=synth_ba.camera_rig("kb8"))

    def mapping():
        out = []
        for _ in range(6):
            navs, pts, erase, res = Optimizer.LocalBundleAdjustmentNavStatePRV(*win)
            r, o = Optimizer.PoseOptimization(fr, ob)
            out.append((navs.tobytes(), pts.tobytes(), erase.tobytes(), r["nav"].tobytes(), o.tobytes()))
        return out

    rng = np.random.default_rng(63)
    FF, keys, urs, descs, P, cams = _fuse_case(rng, None, n_points=3000)

    def fusing():
        out = []
        for _ in range(12):
            bi, bd = fuse_search(FF, keys, urs, descs, P)
            out.append((bi.tobytes(), bd.tobytes()))
        return out

    alone = [track(), mapping(), fusing()]
    assert int(np.frombuffer(alone[1][0][4], np.uint8).size) == len(ob)
    got, errs = [None] * 3, []

    def run(i, fn):
        try:
            got[i] = fn()
        except Exception as e:  # noqa: BLE001
            errs.append((i, e))
    ts = [threading.Thread(target=run, args=(i, fn)) for i, fn in enumerate((track, mapping, fusing))]
    [t.start() for t in ts]
    [t.join(600) for t in ts]
    assert not errs, errs
    assert not any(t.is_alive() for t in ts)
    assert got[0] == alone[0], "the tracked trajectory changed under concurrent local mapping / fusing"
    assert got[1] == alone[1] and got[2] == alone[2]
    assert all(x == alone[1][0] for x in got[1]) and all(x == alone[2][0] for x in got[2])


@pytest.mark.gpu
def test_gpu_tracker_second_stream_is_measured_reported_and_reselected():
    """The tracker's second stream overlaps the first only when the runtime serves them from different hardware queues.
    vieo_tracker_get_stats reports the overlap ratio measured when the stream was chosen; streams the process creates (and
    uses) AFTER the tracker must not slow the frame down -- the frame time is asserted -- and vieo_tracker_reprobe measures
    again and re-selects when the streams have come to share a queue."""
    import ctypes
    from vieo_slam_amd._lib import check, lib
    from vieo_slam_amd.tracker import TrackerReplay
    n = 44
    seq = replay.Sequence(3, n)
    R = TrackerReplay(seq, replay.HipStages())
    st0 = R.trk.stats()
    assert st0["side_stream_selections"] == 1 and 0.5 < st0["side_stream_ratio"] < 2.6, st0
    R.initialise()
    for k in range(1, 22):
        R.before_frame(k), R.step(k)
    base = float(np.median([g for _, g in R.stats["ms_chain"][8:]]))
    # eight streams created and used after the tracker (a LocalMapping thread's BA stream, a viewer, a second tracker ...)
    L = lib()
    streams = []
    buf = ctypes.c_void_p()
    check(L.vieo_dev_malloc(ctypes.byref(buf), 1 << 20))
    host = np.zeros(1 << 20, np.uint8)
    for _ in range(8):
        s = ctypes.c_void_p()
        check(L.vieo_stream_create(ctypes.byref(s)))
        check(L.vieo_memcpy_h2d_async(buf, host.ctypes.data, host.nbytes, s))
        check(L.vieo_stream_synchronize(s))
        streams.append(s)
    n_before = len(R.stats["ms_chain"])
    for k in range(22, n):
        R.before_frame(k), R.step(k)
    after = float(np.median([g for _, g in R.stats["ms_chain"][n_before:]]))
    st1 = R.trk.reprobe()
    assert st1["side_stream_checks"] >= 1 and st1["ms_gpu_median"] > 0
    assert st1["side_stream_ratio"] < 1.6, st1          # the two streams run side by side (after a re-selection if needed)
    assert after <= 1.25 * base + 0.05, (base, after, st0, st1)
    print("tracker streams: ratio %.2f at creation, %.2f after 8 later streams (%d selections, %d checks); GPU ms per frame %.3f -> %.3f"
          % (st0["side_stream_ratio"], st1["side_stream_ratio"], st1["side_stream_selections"], st1["side_stream_checks"], base, after))
    for s in streams:
        L.vieo_stream_destroy(s)
    L.vieo_dev_free(buf)
    R.close()


@pytest.mark.gpu
def test_gpu_tracker_frame_pipelining_is_bit_identical(tmp_path):
    """vieo_track_input.next_left / next_right: the next frame's ExtractORB x 2 + ComputeStereoMatches run on a third stream
    beside this frame's searches and optimisations, the next call adopts them (use_prefetched).  Same kernels on the same
    data: every output of every frame is what the unpipelined tracker returns, from Python and in the C++ replay; a call
    that does not want a pending prefetch discards it; use_prefetched without one is refused."""
    import json
    from tools.write_sequence import write_sequence
    from vieo_slam_amd._lib import VieoError
    from vieo_slam_amd.tracker import TrackerReplay
    n = 34
    seq = replay.Sequence(5, n)
    Ra = TrackerReplay(seq, replay.HipStages())
    ta = Ra.run(n)
    Rb = TrackerReplay(seq, replay.HipStages(), prefetch=True)
    tb = Rb.run(n)
    assert ta.tobytes() == tb.tobytes()
    assert Ra.stats["n_matches"] == Rb.stats["n_matches"] and Ra.stats["n_inliers"] == Rb.stats["n_inliers"]
    assert np.array_equal(Ra.last.keys.view(np.uint8), Rb.last.keys.view(np.uint8)) and np.array_equal(Ra.last.uright, Rb.last.uright)
    # ... and the pre-integrations run ahead (next_imu) were used wherever the reference turned out to be the last frame
    sb = Rb.trk.stats()
    assert sb["frames_prefetched"] == n - 2 and 0 < sb["preints_ahead_used"] <= n - 2
    assert Ra.trk.stats()["preints_ahead_used"] == 0
    # the pre-integration alone run ahead (next_imu without next images): same bytes again
    Rd = TrackerReplay(seq, replay.HipStages(), preint_ahead=True)
    td = Rd.run(n)
    sd = Rd.trk.stats()
    assert td.tobytes() == ta.tobytes() and sd["frames_prefetched"] == 0 and sd["preints_ahead_used"] > 0
    Rd.close()
    # a pending prefetch is discarded by a call that brings its own images; use_prefetched without one is an error
    Rb2 = TrackerReplay(seq, replay.HipStages(), prefetch=True)
    Rb2.initialise()
    Rb2._n_run = n
    for k in range(1, 6):
        Rb2.step(k)
    Rb2._prefetched = False  # frame 6 arrives with its own images although frame 5's call prefetched it
    for k in range(6, 12):
        Rb2.step(k)
    assert np.array(Rb2.traj, ta.dtype).tobytes() == ta[:12].tobytes()
    Rc = TrackerReplay(seq, replay.HipStages(), prefetch=True)
    Rc.initialise()
    Rc._n_run, Rc._prefetched = n, True
    with pytest.raises(VieoError):
        Rc.step(1)
    for R in (Ra, Rb, Rb2, Rc):
        R.close()
    exe = os.path.join(ROOT, "examples", "replay_main")
    path = str(tmp_path / "seq.vseq")
    write_sequence(path, 5, n, seq)
    out = {}
    for pf in (0, 1):
        traj = str(tmp_path / ("t%d.bin" % pf))
        line = subprocess.check_output([exe, path, traj, "--quiet", "--lba-lag", "3", "--prefetch", str(pf), "--warmup", "8"], timeout=600).decode().strip().splitlines()[-1]
        out[pf] = (json.loads(line), open(traj, "rb").read())
    assert out[0][1] == out[1][1] and out[1][0]["prefetch"] == 1
    print("frame pipelining: C++ replay %.3f -> %.3f ms per frame" % (out[0][0]["ms_per_frame"], out[1][0]["ms_per_frame"]))


@pytest.mark.gpu
def test_gpu_cpp_replay_pipelined_equals_unpipelined_bytes(tmp_path):
    """Frame pipelining in the C++ replay -- the next frame's extraction, stereo stage and pre-integration run ahead, the
    latter also when the next call's reference is the newest key frame behind a local-BA write-back
    (vieo_track_input.next_ref_bias: the caller names the reference's bias once LocalMapping's solve has finished) -- changes
    no output: the trajectories of --prefetch 1 and --prefetch 0 are the same bytes."""
    from tools.write_sequence import write_sequence
    exe = os.path.join(ROOT, "examples", "replay_main")
    n = 60
    seq = replay.Sequence(1, n)
    path = str(tmp_path / "seq.vseq")
    write_sequence(path, 1, n, seq)
    out = {}
    for pf in (0, 1):
        traj = str(tmp_path / ("traj%d.bin" % pf))
        subprocess.check_output([exe, path, traj, "--quiet", "--lba-lag", "8", "--prefetch", str(pf)], timeout=600)
        out[pf] = open(traj, "rb").read()
    assert len(out[0]) == n * 176 and out[0] == out[1]
