#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X hot path on EuRoC-shaped synthetic stereo frames.

    python bench.py --gpus N --steps K --warmup W [--batch B]

One process per GPU (the driver launches N>1 through torch.distributed.run).  A *step* is one pass
of the hot path over one batch of B stereo frames that already sits in HBM; frames are independent,
so ranks shard them with no data-path collective ("weak" scaling: B frames per rank per step).
Rank 0 prints ONE JSON line.  The stages inside the timed region are listed in config.workload --
stages of BASELINE.json's metric that are not built yet are named there as missing, never faked.

roofline: per-kernel time is measured live with HIP events on the library's own stream across the
timed steps (vieo_orb_stage_ms); achieved = algorithmic bytes per launch / that time.
cpu_baseline: the CPU oracle (a port of the reference path, oracle/) rebuilt -O3 -march=native on
this host and timed on a bounded sample of the same frames, threaded like the reference (one thread
per camera, src/Frame.cc:259-278).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 752, 480          # EuRoC (Examples/Stereo/EuRoC/EuRoC_VIO.yaml:68-69)
NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH = 1200, 1.2, 8, 20, 7   # EuRoC_VIO.yaml:138-151
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def level_sizes():
    s, out = np.float32(1.0), []
    inv = []
    for l in range(NLEVELS):
        inv.append(np.float32(1.0) / s)
        s = np.float32(s * np.float64(np.float32(SCALE)))
    for l in range(NLEVELS):
        out.append((int(np.rint(np.float32(W) * inv[l])), int(np.rint(np.float32(H) * inv[l]))))
    return out


def algorithmic_bytes_per_image():
    """DESIGN.md 'algorithmic bytes': what each kernel must move per camera image."""
    px = [w * h for w, h in level_sizes()]
    n_kp = NFEAT
    return {
        "pyramid": sum(px[l - 1] + px[l] for l in range(1, NLEVELS)),   # read l-1, write l
        "fast": sum(px),                                                # every level read once
        "blur": 2 * sum(px),                                            # read + write every level
        "quadtree": 0,                                                  # candidate lists, L2-resident
        "describe": n_kp * (709 + 512 + 28 + 32),                       # disc + 512 taps + outputs
        # SURVEY 8d whole-extractor figure: input + levels 1.. + outputs
        "total": px[0] + sum(px[1:]) + n_kp * 60,
    }


def make_frames(n_pairs, seed0=1000):
    from vieo_slam_amd import synth
    base = min(n_pairs, 8)
    pairs = [synth.synth_stereo_pair(seed0 + i, W, H)[:2] for i in range(base)]
    imgs = np.empty((n_pairs, 2, H, W), np.uint8)
    for i in range(n_pairs):
        l, r = pairs[i % base]
        if i >= base:  # cheap distinct variants: shifted content + fresh sensor noise
            sh = 3 * (i // base)
            rng = np.random.default_rng(seed0 + 7000 + i)
            l = np.clip(np.roll(l, sh, 1).astype(np.int16) + rng.integers(-2, 3, l.shape), 0, 255)
            r = np.clip(np.roll(r, sh, 1).astype(np.int16) + rng.integers(-2, 3, r.shape), 0, 255)
        imgs[i, 0], imgs[i, 1] = l, r
    return imgs


def cpu_baseline(imgs, budget_s=12.0):
    """Reference-shaped CPU timing: one thread per camera runs the oracle extractor."""
    from tests import oracle_lib
    try:
        path = oracle_lib.build(native=True)
    except Exception:
        path = oracle_lib.build(native=False)
    orc = oracle_lib.Oracle(path)
    ex = [orc.extractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH) for _ in range(2)]
    n_done, t0 = 0, time.perf_counter()
    while n_done < len(imgs):
        pair = imgs[n_done]
        th = [threading.Thread(target=ex[c], args=(pair[c],)) for c in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        n_done += 1
        if time.perf_counter() - t0 > budget_s and n_done >= 4:
            break
    dt = time.perf_counter() - t0
    return {"value": n_done / dt, "unit": "frames/s", "cores": 2, "kind": "port",
            "sample": "%d synthetic stereo frames 752x480, ORB extraction x2 cameras, oracle "
                      "-O3 -march=native, 1 thread per camera (nproc=%d)" % (n_done, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="stereo frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch  # loaded first so the process uses one HIP runtime
    import torch.distributed as dist
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from vieo_slam_amd._lib import DeviceBuffer
    from vieo_slam_amd.orb_extractor import ORBextractor, STAGES

    B = a.batch
    imgs = make_frames(B, seed0=1000 + 100000 * rank)
    n_img = 2 * B
    ext = ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)
    cap = ext.max_keypoints()
    d_img = DeviceBuffer(imgs.nbytes)
    d_img.upload(imgs)
    d_kp, d_desc, d_cnt = DeviceBuffer(n_img * cap * 28), DeviceBuffer(n_img * cap * 32), DeviceBuffer(n_img * 8)

    def step():
        ext.extract_batch_device(d_img.ptr, n_img, W, H, W, W * H, d_kp.ptr, d_desc.ptr, cap, d_cnt.ptr)

    def sync_all():
        ext.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    ext.enable_timing(True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    stage = ext.stage_ms_all()
    cnt = d_cnt.download(np.int32, (n_img, 2))
    if rank == 0:
        avg = {k: float(np.mean([s[k] for s in stage])) for k in STAGES}
        ab = algorithmic_bytes_per_image()
        dom = max((k for k in STAGES if k != "total"), key=lambda k: avg[k])
        achieved = ab[dom] * n_img / (avg[dom] * 1e-3) / 1e9 if avg[dom] > 0 else 0.0
        out = {
            "metric": "frontend+localBA frames/sec on EuRoC MH05 stereo-VIO; ATE vs ref",
            "value": B * a.steps * world / dt,
            "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {
                "workload": "EuRoC-shaped synthetic stereo 752x480, 1200 feats, 1.2x8 levels, FAST 20/7: "
                            "ORBextractor x2 cameras per frame ONLY; stereo match, projection search, "
                            "PoseOptimization and LocalBA are NOT yet in the timed region",
                "stereo_frames_per_gpu_per_step": B, "parallelism": "frames sharded 1 batch/GPU, no collective",
                "mean_keypoints_per_image": float(cnt[:, 0].mean()),
            },
            "stage_ms_per_step": avg,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": ab[dom] * n_img,
                         "avg_launch_ms": avg[dom]},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(imgs)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
