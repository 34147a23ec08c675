"""Writes tests/golden/orb_golden.npz from the CPU oracle on seeded synthetic frames.

The reference repository holds no golden vectors for the extractor and cannot be built here
(OpenCV absent), so these fixtures pin the ORACLE (and, through the parity tests, the HIP path)
against regressions; they are not outputs of the reference itself ("parity unpinned").
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib  # noqa: E402
from vieo_slam_amd import synth  # noqa: E402

if __name__ == "__main__":
    o = oracle_lib.load()
    out = {}
    for tag, (seed, w, h, nfeat, lap) in {"euroc": (1000, 752, 480, 1200, None),
                                          "tumvi": (1001, 512, 512, 1500, (0, 511))}.items():
        e = o.extractor(nfeat)
        mono, kps, desc = e(synth.synth_image(seed, w, h), lapping=lap)
        out[tag + "_mono"] = np.int32(mono)
        out[tag + "_kps"] = kps
        out[tag + "_desc"] = desc
        print(tag, mono, len(kps))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "orb_golden.npz"), **out)
