#!/usr/bin/env python3
"""Absolute trajectory error between a ground-truth file and an estimated trajectory, as the reference's
Examples/RunEuRoC/EvaluateEuRoC_Evaluate.sh:38-57 runs it:

    python tools/evaluate_ate.py groundtruth.txt KeyFrameTrajectoryIMU.txt [--offset 0.2] [--scale 1.0]
        [--max_difference 0.02] [--with_scale] [--verbose]

Prints the RMSE in the ground truth's units, or every statistic with --verbose."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vieo_slam_amd import trajectory


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("first_file", help="ground truth trajectory (format: timestamp tx ty tz qx qy qz qw)")
    ap.add_argument("second_file", help="estimated trajectory (same leading columns)")
    ap.add_argument("--offset", type=float, default=0.0, help="time offset added to the timestamps of the second file")
    ap.add_argument("--scale", type=float, default=1.0, help="scaling factor for the second trajectory")
    ap.add_argument("--max_difference", type=float, default=0.02)
    ap.add_argument("--with_scale", action="store_true", help="also estimate the scale in closed form")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    r = trajectory.evaluate_ate(a.first_file, a.second_file, offset=a.offset, max_difference=a.max_difference,
                                scale=a.scale, with_scale=a.with_scale)
    if not a.verbose:
        print("%f" % r["rmse"])
        return
    print("compared_pose_pairs %d pairs" % r["compared_pose_pairs"])
    for k in ("rmse", "mean", "median", "std", "min", "max"):
        print("absolute_translational_error.%s %f m" % (k, r[k]))
    if a.with_scale:
        print("scale %f" % r["scale"])
        print("absolute_translational_error_no_scale.rmse %f m" % r["rmse_no_scale"])


if __name__ == "__main__":
    main()
