"""Second half of the pinning recipe (tools/rebaseline_with_reference.sh): the reference's own ORBextractor
(oracle/_ref/libvieo_ref.so, built by oracle/ref_build/) against the restated oracle (oracle/_build/liboracle.so) on
the seeded frames behind tests/golden/orb_golden.npz, stage by stage: pyramid levels -> key points -> descriptors.
Prints the first differing stage per case; --write re-baselines tests/golden/orb_golden.npz from the REFERENCE.

What cannot match even with a real OpenCV: DistributeOctTree sorts (size, node pointer) pairs
(src/ORBextractor.cc:647), so ties between equally full nodes are broken by HEAP ADDRESS in the reference
(about 570 ties per EuRoC frame); the oracle and the HIP path break them by creation order.  Key points are
therefore compared as sets first (that is what the address order can change only through the N-feature cut),
and the report counts the ties the oracle saw (tie_count)."""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib  # noqa: E402
from vieo_slam_amd import synth  # noqa: E402

REF_SO = os.path.join(ROOT, "oracle", "_ref", "libvieo_ref.so")
CASES = {"euroc": (1000, 752, 480, 1200, None), "tumvi": (1001, 512, 512, 1500, (0, 511))}


def ref_extractor(nfeat):
    """The oracle's Python wrapper bound to the reference build: same hooks, same signatures."""
    L = ctypes.CDLL(REF_SO)
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    L.vo_orb_create.restype = P
    L.vo_orb_create.argtypes = [I, F, I, I, I]
    L.vo_orb_destroy.argtypes = [P]
    L.vo_orb_extract.argtypes = [P, P, I, I, I, P, P, P, I, P]
    L.vo_orb_scale_factor.argtypes = [P, I]
    L.vo_orb_scale_factor.restype = F
    L.vo_orb_level_size.argtypes = [P, I, P, P]
    L.vo_orb_get_plane.argtypes = [P, I, I, P]
    return oracle_lib.OracleExtractor(L, nfeat, 1.2, 8, 20, 7)


def key_set(kps):
    return {(float(k["x"]), float(k["y"]), int(k["octave"])) for k in kps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="re-baseline tests/golden/orb_golden.npz from the reference")
    args = ap.parse_args()
    if not os.path.exists(REF_SO):
        sys.exit("no %s: build it with tools/rebaseline_with_reference.sh (needs OpenCV >= 4.5)" % REF_SO)
    orc = oracle_lib.load()
    out, worst = {}, "identical"
    for tag, (seed, w, h, nfeat, lap) in CASES.items():
        img = synth.synth_image(seed, w, h)
        eo, er = orc.extractor(nfeat), ref_extractor(nfeat)
        mo, ko, do = eo(img, lapping=lap)
        mr, kr, dr = er(img, lapping=lap)
        first = None
        for lvl in range(8):
            if not np.array_equal(eo.plane(lvl, 0), er.plane(lvl, 0)):
                first = "pyramid level %d (cv::resize INTER_LINEAR restated in oracle/ocv_prims.hpp)" % lvl
                break
        if first is None and key_set(ko) != key_set(kr):
            d = key_set(ko) ^ key_set(kr)
            first = ("key-point set: %d of %d differ (FAST / NMS / quadtree; the oracle saw %d address-order ties)"
                     % (len(d), len(kr), eo.tie_count()))
        if first is None and (mo != mr or not np.array_equal(ko.view(np.uint8), kr.view(np.uint8))):
            first = "key-point order / angle / response (IC_Angle: cv::fastAtan2 restated)"
        if first is None and not np.array_equal(do, dr):
            first = "descriptors (GaussianBlur Q8.8 kernel / steered BRIEF rounding)"
        print("%s: %s" % (tag, first or "reference == oracle, bit for bit (%d key points)" % len(kr)))
        if first:
            worst = first
        out[tag + "_mono"], out[tag + "_kps"], out[tag + "_desc"] = np.int32(mr), kr, dr
    if args.write:
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "orb_golden.npz"), **out)
        with open(os.path.join(ROOT, "tests", "golden", "MANIFEST.md"), "a") as f:
            f.write("\n* orb_golden.npz re-baselined from the REFERENCE build (oracle/_ref/libvieo_ref.so); "
                    "first difference to the restated oracle at that time: %s\n" % worst)
        print("tests/golden/orb_golden.npz now holds the reference's outputs")


if __name__ == "__main__":
    main()
