"""ExtractORB of the bench's 4096 images: one launch chain over the whole batch against chains over sub-batches of S images
on one or two extractor handles (alternating streams) -- the sub-batch's pyramid / blurred planes (2.7 MB per image) are
then re-used scratch that fits the 256 MiB Infinity Cache.  Wall time per 4096 images, mean of N rounds.
python tools/time_extractor_sub.py [frames] [rounds]"""
import sys
import time

import numpy as np

from vieo_slam_amd.orb_extractor import ORBextractor
from vieo_slam_amd.pipeline import FramePipeline, make_cases, W, H
from vieo_slam_amd._lib import check, lib


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    P = FramePipeline(make_cases(min(64, B), seed0=1, workload="r3"), B, seed=0, workload="r3")
    n_img, cap = P.n_img, P.cap
    L = lib()

    def whole():
        P.ext.extract_batch_device(P.d_img.ptr, n_img, W, H, W, W * H, P.d_kp.ptr, P.d_desc.ptr, cap, P.d_cnt.ptr)
        P.ext.sync()

    def timed(f):
        for _ in range(2):
            f()
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        return 1e3 * (time.perf_counter() - t0) / n

    ref = timed(whole)
    cnt_ref = P.d_cnt.download(np.int32, (n_img, 2)).copy()
    kp_ref = P.d_kp.download(np.uint8, (n_img * cap * 28,)).copy()
    print("whole batch (%d images, one chain): %.2f ms" % (n_img, ref))
    for S in (32, 64, 128, 256, 512, 1024):
        for n_h in (1, 2):
            exts = [ORBextractor(1200, 1.2, 8, 20, 7) for _ in range(n_h)]

            def sub():
                for i, a in enumerate(range(0, n_img, S)):
                    e = exts[i % n_h]
                    m = min(S, n_img - a)
                    e.extract_batch_device(P.d_img.ptr + a * W * H, m, W, H, W, W * H, P.d_kp.ptr + a * cap * 28,
                                           P.d_desc.ptr + a * cap * 32, cap, P.d_cnt.ptr + a * 8)
                for e in exts:
                    e.sync()
            ms = timed(sub)
            ok = np.array_equal(P.d_cnt.download(np.int32, (n_img, 2)), cnt_ref) and \
                np.array_equal(P.d_kp.download(np.uint8, (n_img * cap * 28,)), kp_ref)
            print("sub-batches of %4d images on %d handle(s): %.2f ms  (%.2fx)  identical keys: %s" % (S, n_h, ms, ms / ref, ok))
            del exts


if __name__ == "__main__":
    main()
