"""Writes tests/golden/ba_golden.npz from the CPU oracle on the seeded cases of ba_cases.py.

Like orb_golden.npz these fixtures pin the ORACLE (and through the GPU test the HIP path) against regressions;
they are not outputs of the reference, which cannot be built here ("parity unpinned").
Run from the repo root:  python tests/golden/make_ba_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib  # noqa: E402
from tests.golden import ba_cases  # noqa: E402


def oracle_api(o):
    return {"pose": o.pose_optimization, "pose_vio": o.pose_optimization_vio, "lba": o.local_ba,
            "lba_vio": o.local_ba_vio, "gba_vio": o.global_ba_vio, "fisheye": o.stereo_fisheye,
            "tri": o.search_for_triangulation}


if __name__ == "__main__":
    out = {name: arr for name, arr, _ in ba_cases.cases(oracle_api(oracle_lib.load()))}
    for k, v in out.items():
        print(k, v.shape, v.dtype)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ba_golden.npz"), **out)
