#!/usr/bin/env python3
"""Runs only the batched extractor a few times (for rocprofv3 counter passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vieo_slam_amd import synth
from vieo_slam_amd._lib import DeviceBuffer
from vieo_slam_amd.orb_extractor import ORBextractor
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
base = [synth.synth_image(1000 + i) for i in range(8)]
imgs = np.stack([base[i % 8] for i in range(B)])
e = ORBextractor(1200, 1.2, 8, 20, 7)
cap = e.max_keypoints()
d_img = DeviceBuffer(imgs.nbytes); d_img.upload(imgs)
d_kp, d_desc, d_cnt = DeviceBuffer(B * cap * 28), DeviceBuffer(B * cap * 32), DeviceBuffer(B * 8)
e.enable_timing(True)
for _ in range(reps):
    e.extract_batch_device(d_img.ptr, B, 752, 480, 752, 752 * 480, d_kp.ptr, d_desc.ptr, cap, d_cnt.ptr)
e.sync()
ms = e.stage_ms_all()
print({k: round(float(np.mean([m[k] for m in ms[1:]])), 4) for k in ms[0]})
import zlib
from vieo_slam_amd.orb_extractor import KEYPOINT_DTYPE
cnt = d_cnt.download(np.int32, (B, 2))
kp = d_kp.download(KEYPOINT_DTYPE, (B, cap))
ds = d_desc.download(np.uint8, (B, cap, 32))
crc = 0
for b in range(min(B, 16)):
    n = cnt[b, 0]
    crc = zlib.crc32(kp[b, :n].tobytes(), crc)
    crc = zlib.crc32(ds[b, :n].tobytes(), crc)
print("keys", int(cnt[:, 0].sum()), "crc %08x" % crc)
