"""The ctypes mirror's record layouts against the C header: sizeof / offsetof of the one-call tracker's records as gcc sees
include/vieo_hot.h must be what the numpy dtypes of vieo_slam_amd/tracker.py say (no GPU: the header is plain C)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r"""
#include <stdio.h>
#include <stddef.h>
#include "vieo_hot.h"
int main(void) {
  printf("input %zu\n", sizeof(vieo_track_input));
  printf("input.next_left %zu\n", offsetof(vieo_track_input, next_left));
  printf("input.use_prefetched %zu\n", offsetof(vieo_track_input, use_prefetched));
  printf("input.next_n_imu %zu\n", offsetof(vieo_track_input, next_n_imu));
  printf("input.next_imu %zu\n", offsetof(vieo_track_input, next_imu));
  printf("input.next_t_cur %zu\n", offsetof(vieo_track_input, next_t_cur));
  printf("input.next_images %zu\n", offsetof(vieo_track_input, next_images));
  printf("input.images %zu\n", offsetof(vieo_track_input, images));
  printf("input.nav_ref %zu\n", offsetof(vieo_track_input, nav_ref));
  printf("stats %zu\n", sizeof(vieo_tracker_stats));
  printf("output %zu\n", sizeof(vieo_track_output));
  printf("params %zu\n", sizeof(vieo_tracker_params));
  printf("rig %zu\n", sizeof(vieo_tracker_rig));
  printf("keypoint %zu\n", sizeof(vieo_keypoint));
  printf("navstate %zu\n", sizeof(vieo_navstate));
  printf("imu_sample %zu\n", sizeof(vieo_imu_sample));
  return 0;
}
"""


def test_tracker_records_match_the_header(tmp_path):
    from vieo_slam_amd import tracker
    from vieo_slam_amd.ba_types import NAVSTATE_DTYPE
    src = tmp_path / "sizes.c"
    src.write_text(PROG)
    exe = str(tmp_path / "sizes")
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    got = dict(line.rsplit(" ", 1) for line in subprocess.check_output([exe]).decode().strip().splitlines())
    got = {k: int(v) for k, v in got.items()}
    D = tracker.TRACK_INPUT_DTYPE
    assert got["input"] == D.itemsize, (got["input"], D.itemsize)
    for f in ("next_left", "use_prefetched", "next_n_imu", "next_imu", "next_t_cur", "next_images", "images", "nav_ref"):
        assert got["input." + f] == D.fields[f][1], (f, got["input." + f], D.fields[f][1])
    assert got["stats"] == tracker.Tracker.STATS_DTYPE.itemsize
    assert got["output"] == tracker.TRACK_OUTPUT_DTYPE.itemsize and got["params"] == tracker.TRACKER_PARAMS_DTYPE.itemsize
    assert got["rig"] == tracker.TRACKER_RIG_DTYPE.itemsize
    assert got["navstate"] == NAVSTATE_DTYPE.itemsize
    assert got["imu_sample"] == tracker.IMU_SAMPLE_DTYPE.itemsize and got["keypoint"] == 28
