"""Writes the synthetic stereo-inertial sequence of vieo_slam_amd/replay.py (replay.Sequence: rendered 752 x 480 stereo
frames at 20 Hz, IMU samples at 200 Hz, true states) as ONE binary file for examples/replay_main.cc, the C++ replay
that runs without Python.  Layout (little endian):
  char magic[8] = "VSEQ0001"; int32 n_frames, width, height, n_imu; double dt, t0;
  vieo_imu_noise noise (160 B); double bg[3], ba[3]; double Tbc[16], Tcb[16] (row-major 4x4; Tcb = numpy's inverse, as the Python driver uses it);
  double fx, fy, cx, cy, bf, baseline, th_depth, pad;
  vieo_imu_sample imu[n_imu]; double truth[n_frames][10] (p, q wxyz, v); uint8 images[n_frames][2][height][width].
A camera-rig sequence (replay_modes.RigSequence: 2..4 distorted cameras) is "VSEQ0002": the same header (width / height of
the rig's cameras; fx..cy of camera 0, bf, baseline, th_depth = 35), then int32 n_cams, nfeatures; vieo_tracker_params;
vieo_tracker_rig (both as tracker.rig_params builds them -- the C++ program hands them to vieo_tracker_create_rig as they
are); then imu, truth, uint8 images[n_frames][n_cams][height][width].  examples/replay_modes.cc reads both.
usage: python tools/write_sequence.py out.vseq [--seed 1] [--frames 100] [--rig radtan|kb8 --cams 2 --features 1200]"""
import argparse
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vieo_slam_amd import replay, synth_ba  # noqa: E402
from vieo_slam_amd import synth_scene as sc  # noqa: E402


def write_sequence(path, seed=1, n_frames=100, seq=None):
    seq = seq or replay.Sequence(seed, n_frames)
    with open(path, "wb") as f:
        f.write(b"VSEQ0001")
        f.write(struct.pack("<iiii", n_frames, replay.W, replay.H, len(seq.imu)))
        f.write(struct.pack("<dd", seq.dt, seq.t0))
        f.write(seq.noise.tobytes())
        f.write(np.asarray(seq.bg, np.float64).tobytes() + np.asarray(seq.ba, np.float64).tobytes())
        f.write(np.asarray(synth_ba.EUROC_TBC, np.float64).tobytes())
        f.write(np.ascontiguousarray(np.linalg.inv(synth_ba.EUROC_TBC), np.float64).tobytes())
        f.write(struct.pack("<dddddddd", sc.FX, sc.FY, sc.CX, sc.CY, sc.BF, sc.BASELINE, replay.TH_DEPTH, 0.0))
        f.write(seq.imu.tobytes())
        tr = np.zeros((n_frames, 10))
        for k in range(n_frames):
            t = seq.truth(k)
            tr[k, :3], tr[k, 3:7], tr[k, 7:] = t["p"], t["q"], t["v"]
        f.write(tr.tobytes())
        for k in range(n_frames):
            L, R = seq.images(k)
            f.write(np.ascontiguousarray(L, np.uint8).tobytes())
            f.write(np.ascontiguousarray(R, np.uint8).tobytes())
    return seq


def write_rig_sequence(path, seq, nfeatures, th_last=7.0, th_local=2.0, max_local_points=16384, th_depth=35.0):
    """seq: replay_modes.RigSequence.  The tracker parameters are the ones replay_modes.RigTrackerReplay creates its tracker with."""
    from vieo_slam_amd.tracker import rig_params
    scene = seq.scene
    nc = len(scene.cams)
    prm, rg = rig_params(scene, nfeatures, max_local_points=max_local_points, th_last=th_last, th_local=th_local, noise=seq.noise[0],
                         th_depth=th_depth)
    c0 = scene.cams[0]
    with open(path, "wb") as f:
        f.write(b"VSEQ0002")
        f.write(struct.pack("<iiii", seq.n_frames, scene.W, scene.H, len(seq.imu)))
        f.write(struct.pack("<dd", seq.dt, seq.t0))
        f.write(seq.noise.tobytes())
        f.write(np.asarray(seq.bg, np.float64).tobytes() + np.asarray(seq.ba, np.float64).tobytes())
        f.write(np.asarray(synth_ba.EUROC_TBC, np.float64).tobytes())
        f.write(np.ascontiguousarray(np.linalg.inv(synth_ba.EUROC_TBC), np.float64).tobytes())
        f.write(struct.pack("<dddddddd", float(c0["fx"]), float(c0["fy"]), float(c0["cx"]), float(c0["cy"]), float(prm[0]["bf"]),
                            float(prm[0]["baseline"]), th_depth, 0.0))
        f.write(struct.pack("<ii", nc, nfeatures))
        f.write(prm.tobytes())
        f.write(rg.tobytes())
        f.write(seq.imu.tobytes())
        tr = np.zeros((seq.n_frames, 10))
        for k in range(seq.n_frames):
            t = seq.truth(k)
            tr[k, :3], tr[k, 3:7], tr[k, 7:] = t["p"], t["q"], t["v"]
        f.write(tr.tobytes())
        for k in range(seq.n_frames):
            for im in seq.images(k):
                f.write(np.ascontiguousarray(im, np.uint8).tobytes())
    return seq


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--rig", default=None, help="radtan | kb8: a distorted camera rig (VSEQ0002)")
    ap.add_argument("--cams", type=int, default=2)
    ap.add_argument("--features", type=int, default=1200)
    a = ap.parse_args()
    if a.rig:
        from vieo_slam_amd import replay_modes
        write_rig_sequence(a.out, replay_modes.RigSequence(a.seed, a.frames, a.rig, a.cams), a.features)
    else:
        write_sequence(a.out, a.seed, a.frames)
    print("wrote", a.out, os.path.getsize(a.out), "bytes")
