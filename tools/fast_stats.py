"""What k_fast's cells cost on a workload: counters of a private build (tools/build_fast_stats.sh, -DVIEO_FAST_STATS).
VIEO_LIB_PATH=vieo_slam_amd/libvieo_hot_faststats.so python tools/fast_stats.py [frames]"""
import ctypes
import sys

import numpy as np

from vieo_slam_amd._lib import lib, check
from vieo_slam_amd.pipeline import FramePipeline, make_cases

NAMES = ["cells", "cells that went on to minThFAST", "cells evaluated densely", "batches of 64 exact strengths",
         "batches of pass C", "(unused)", "compass survivors", "corners kept"]


def read():
    out = (ctypes.c_ulonglong * 8)()
    L = lib()
    L.vieo_fast_stats.restype = ctypes.c_int
    check(L.vieo_fast_stats(out), "vieo_fast_stats")
    return np.array(list(out), np.int64)


def report(tag, s):
    print(tag)
    for n, v in zip(NAMES, s):
        print("  %-34s %10d   %.3f per cell" % (n, v, v / max(1, s[0])))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for wl in ("r3", "r2"):
        cases = make_cases(min(64 if wl == "r3" else 8, B), seed0=1, workload=wl)
        P = FramePipeline(cases, B, seed=0, workload=wl)
        read()
        P.step()
        P.sync()
        report("bench workload %s, %d stereo frames" % (wl, B), read())


if __name__ == "__main__":
    main()
