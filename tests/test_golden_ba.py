"""Committed golden vectors of the optimisation / fisheye-stereo paths (tests/golden/ba_golden.npz, written by
tests/golden/make_ba_golden.py from the oracle): the oracle must still reproduce them (CPU), and the HIP path must
match them within the parity tolerance (GPU)."""
import os

import numpy as np
import pytest

from tests.golden import ba_cases
from tests.golden.make_ba_golden import oracle_api

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ba_golden.npz")


def _compare(api, tight):
    g = np.load(GOLD)
    seen = set()
    for name, arr, tol in ba_cases.cases(api):
        ref = g[name]
        seen.add(name)
        assert arr.shape == ref.shape, name
        if tol == 0:
            assert np.array_equal(arr, ref), name
        elif name.endswith("n_erase"):
            assert abs(int(arr[0]) - int(ref[0])) <= (0 if tight else tol), name
        else:
            assert np.abs(arr.astype(np.float64) - ref).max() <= (1e-9 if tight else tol), name
    assert seen == set(g.files)


def test_oracle_reproduces_committed_ba_golden(oracle):
    _compare(oracle_api(oracle), tight=True)


@pytest.mark.gpu
def test_hip_matches_committed_ba_golden():
    from vieo_slam_amd.matching import compute_stereo_fisheye_matches
    from vieo_slam_amd.optimizer import Optimizer as O
    from vieo_slam_amd.tri_search import SearchForTriangulation
    api = {"pose": O.PoseOptimization, "pose_vio": O.PoseOptimizationVIO, "lba": O.LocalBundleAdjustment,
           "lba_vio": O.LocalBundleAdjustmentNavStatePRV, "gba_vio": O.GlobalBundleAdjustmentNavStatePRV,
           "fisheye": compute_stereo_fisheye_matches, "tri": SearchForTriangulation}
    _compare(api, tight=False)
