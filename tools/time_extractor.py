"""ExtractORB alone on the bench's batch (2048 stereo frames = 4096 images in HBM): ms per kernel stage, mean of N calls.
python tools/time_extractor.py [frames] [calls]   (VIEO_LIB_PATH picks the build: A/B runs)"""
import sys

import numpy as np

from vieo_slam_amd.pipeline import FramePipeline, make_cases, W, H


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    P = FramePipeline(make_cases(min(64, B), seed0=1, workload="r3"), B, seed=0, workload="r3")
    call = lambda: P.ext.extract_batch_device(P.d_img.ptr, P.n_img, W, H, W, W * H, P.d_kp.ptr, P.d_desc.ptr, P.cap, P.d_cnt.ptr)
    for _ in range(3):
        call()
    P.ext.sync()
    P.ext.enable_timing(True)
    for _ in range(n):
        call()
    P.ext.sync()
    ms = P.ext.stage_ms_all()[-n:]
    print({k: round(float(np.mean([m[k] for m in ms])), 3) for k in ms[0]})


if __name__ == "__main__":
    main()
