"""examples/dropin_replay.cc: the sequential replay driven only through the entries behind the reference's own class
members (ORBextractor::operator() per camera thread, Frame::ComputeStereoMatches, the two ORBmatcher::SearchByProjection
overloads, Optimizer::PoseOptimization x 2, LocalBundleAdjustmentNavStatePRV on the LocalMapping thread) -- the path that
needs no change in Tracking.cc / LocalMapping.cc.  Checked against the Python twin (replay.Replay on the same entries) and
against the ORACLE replay (BASELINE configs[2]: ATE within 1e-4 of the reference path)."""
import json
import os
import subprocess

import numpy as np
import pytest

from vieo_slam_amd import replay, synth_ba

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "dropin_replay")


def _run(path, traj_path, *flags):
    line = subprocess.check_output([EXE, path, traj_path, "--quiet", *flags], timeout=900).decode().strip().splitlines()[-1]
    from vieo_slam_amd.ba_types import NAVSTATE_DTYPE
    return json.loads(line), np.fromfile(traj_path, NAVSTATE_DTYPE)


def test_dropin_replay_is_built():
    assert os.path.exists(EXE), "examples/dropin_replay is built by __graft_entry__.build()"


@pytest.mark.gpu
def test_gpu_dropin_replay_resident_and_host_pointer_forms_agree_with_python_and_oracle(oracle, tmp_path):
    from tests.replay_oracle import OracleStages
    from tools.write_sequence import write_sequence
    n = 60
    seq = replay.Sequence(1, n)
    path = str(tmp_path / "seq.vseq")
    write_sequence(path, 1, n, seq)
    r1, t1 = _run(path, str(tmp_path / "t1.bin"), "--resident", "1")
    r0, t0 = _run(path, str(tmp_path / "t0.bin"), "--resident", "0")
    assert r1["resident"] == 1 and r0["resident"] == 0 and r1["frames"] == r0["frames"] == n - 1
    # the resident entries are the host-pointer entries on the same data: identical trajectories
    assert t1.tobytes() == t0.tobytes()
    # the Python twin, resident and not, agree with each other exactly and with the C++ host up to its glue's rounding
    Rp = replay.Replay(seq, replay.HipStages(resident=True))
    tp = Rp.run(n)
    assert Rp.S.resident_calls >= 3 * (n - 1) + 1  # stereo + two searches per tracked frame, the first frame's stereo
    Rq = replay.Replay(seq, replay.HipStages())
    tq = Rq.run(n)
    assert tp.tobytes() == tq.tobytes()
    assert r1["local_bas"] == Rp.stats["lba"] == 5 and r1["key_frames"] == len(Rp.kfs) and r1["map_points"] == len(Rp.mp_X)
    d = np.linalg.norm(t1["p"] - tp["p"], axis=1)
    assert d.max() <= 5e-5, d.max()  # (plain C++ loops against numpy in the host glue: both within 1e-4 of the oracle below)
    # ... and the oracle replay: BASELINE's "ATE within 1e-4 of ref"
    Ro = replay.Replay(seq, OracleStages(oracle))
    to = Ro.run(n)
    ate = replay.ate_between(t1, to)
    dmax = np.linalg.norm(t1["p"] - to["p"], axis=1).max()
    rot = max(synth_ba.pose_error(t1[k], to[k])[1] for k in range(n))
    assert ate <= 1e-4 and dmax <= 1e-4 and rot <= 1e-4, (ate, dmax, rot)
    assert r1["max_err_vs_truth_m"] < 1.5e-2
    print("drop-in replay (C++): resident %.3f ms per frame, host-pointer form %.3f; stages %s; ATE vs oracle %.2e m"
          % (r1["ms_per_frame"], r0["ms_per_frame"], r1["stage_ms_per_frame"], ate))


@pytest.mark.gpu
def test_gpu_dropin_replay_local_ba_beside_tracking_vs_oracle(oracle, tmp_path):
    """LocalMapping on its own host thread (src/LocalMapping.cc:113-139) with the reproducible hand-over of
    examples/replay_main.cc (--lba-lag): the oracle replay applies the same lag."""
    from tests.replay_oracle import OracleStages
    from tools.write_sequence import write_sequence
    n, lag = 60, 6
    seq = replay.Sequence(1, n)
    path = str(tmp_path / "seq.vseq")
    write_sequence(path, 1, n, seq)
    r, t = _run(path, str(tmp_path / "t.bin"), "--lba-lag", str(lag), "--warmup", "12")
    Ro = replay.Replay(seq, OracleStages(oracle), lba_lag=lag)
    to = Ro.run(n)
    assert r["lba_lag"] == lag and r["local_bas"] == Ro.stats["lba"] == 5
    ate = replay.ate_between(t, to)
    assert ate <= 1e-4 and np.linalg.norm(t["p"] - to["p"], axis=1).max() <= 1e-4, ate
    print("drop-in replay, local BA beside tracking (lag %d): %.3f ms per frame, ATE vs the oracle run %.2e m"
          % (lag, r["ms_per_frame"], ate))
