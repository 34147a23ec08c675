#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X hot path on EuRoC-shaped synthetic stereo-inertial frames.

    python bench.py --gpus N --steps K --warmup W [--batch B]

Workload (BASELINE.json configs[1]/[2]: "EuRoC MH05 stereo-VIO, 1200 feats, 1xMI355X"): per stereo
frame, in the reference's call order
    ORBextractor x2 -> ComputeStereoMatches -> SearchByProjection(last frame) -> PoseOptimization(VIO)
    -> SearchByProjection(local map) -> PoseOptimization(VIO, bComputeMarg)
plus one LocalBundleAdjustmentNavStatePRV (Optimizer.cc:21-769: 10 local key frames with PR + V + Bias
vertices chained by IMU pre-integrations, the key frame before the window and 5 more fixed observers,
2000 points) per `--lba-every` frames (a key frame every <= 10 frames at 20 Hz), the
windows of a step issued as one lock-step batch from a host thread like the reference's LocalMapping
thread.

One process per GPU (the driver launches N>1 through torch.distributed.run).  A *step* is one pass
of that path over a batch of B independent frames that already sit in HBM; ranks shard frames
with no data-path collective ("weak" scaling: B frames per rank per step).  Rank 0 prints ONE
JSON line.

Besides the batched headline the line carries
  single_stream: the SEQUENTIAL replay (frame t's pose, map points and marginal prior feed frame t+1, one
      LocalBundleAdjustmentNavStatePRV per 10 frames with write-back) on ONE stream of frames, run by the C++ program
      examples/replay_main on the C-ABI (vieo_track_frame: a frame's tracking as one chain of launches, one copy up, one
      back, one synchronisation): ms per frame, frames/s, the ratio to the same replay on the CPU oracle
      (vs_cpu_single_stream), the reference's own published front-end figure beside it (reference_readme_anchor), and the
      ATE of that trajectory against the oracle's (BASELINE configs[2]: "ATE within 1e-4 of ref");
      `drop_in` inside it: examples/dropin_replay, the same replay through the per-member entries behind the reference's own
      signatures (nothing changes in Tracking.cc); `concurrent_trackers`: 8 / 32 independent sequences on the one GPU;
  single_stream_rig / single_stream_vision_only: SEQUENCE replays of the other BASELINE configurations (distorted rigs of
      2 / 4 cameras with the visual-inertial local BA, rectified stereo without IMU with the vision-only local BA), one
      vieo_track_frame call per frame, ATE against the oracle's replay;
  pcie_inclusive: the batched step again with the step's images arriving from pinned host memory on a copy stream
      (double-buffered, overlapped with the previous step's kernels); inclusive: the same WITH the step's local BAs.

roofline: stage/kernel time is measured live with HIP events on the library's own stream across
the timed steps; achieved = algorithmic bytes per launch / average launch duration of
the dominant kernel.  cpu_baseline: the CPU oracle (a port of the reference path, oracle/) rebuilt
-O3 -march=native on this host and timed on a bounded sample of the same frames, threaded like the
reference (one thread per camera for extraction, src/Frame.cc:259-278; the rest on one thread).
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
BOUNDS = np.array([0, W, 0, H], np.float32)


def level_sizes():
    s, inv = np.float32(1.0), []
    for _ in range(NLEVELS):
        inv.append(np.float32(1.0) / s)
        s = np.float32(s * np.float64(np.float32(SCALE)))
    return [(int(np.rint(np.float32(W) * i)), int(np.rint(np.float32(H) * i))) for i in inv]


def algorithmic_bytes_per_image():
    """DESIGN.md 'algorithmic bytes': what each extractor kernel must move per camera image."""
    px = [w * h for w, h in level_sizes()]
    return {
        "pyramid": sum(px[l - 1] + px[l] for l in range(1, NLEVELS)),   # read l-1, write l
        "fast": sum(px),                                                # every level read once
        "blur": 2 * sum(px),                                            # read + write every level
        "quadtree": 0,                                                  # candidate lists, L2-resident
        "describe": NFEAT * (709 + 512 + 28 + 32),                      # disc + 512 taps + outputs
        "total": px[0] + sum(px[1:]) + NFEAT * 60,                      # SURVEY 8d whole-extractor figure
    }


FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (dense) peak, v_mfma_f64_16x16x4_f64


def _latest_profile(suffix):
    """Newest committed profiles/r<N>*<suffix> (the PMC passes are separate rocprofv3 runs, MI355X_MICROARCH.md)."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*" + suffix)):
        m = re.match(r"r(\d+)", os.path.basename(f))
        if m and (best is None or int(m.group(1)) >= best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def _counters_match_source(d):
    """A committed counter file describes the kernels of ONE source text: it records sha256[:16] of the kernel's source
    file (tools/pmc_round5.sh), and its figures are quoted only while that file is unchanged -- counters of an older
    kernel are not evidence about this one (round 4 quoted round-3 counters)."""
    import hashlib
    shas = d.get("source_sha16")
    if not isinstance(shas, dict) or not shas:
        return False
    for rel, sha in shas.items():
        try:
            with open(os.path.join(ROOT, rel), "rb") as f:
                if hashlib.sha256(f.read()).hexdigest()[:16] != sha:
                    return False
        except OSError:
            return False
    return True


def pmc_traffic(kernel, n_img):
    """HBM bytes per launch of the kernel from the committed PMC passes (profiles/r*_pmc_extractor.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of the same launch shape, gfx950 correction
    FETCH x 2), scaled to this run's images per launch; None if the file is not there."""
    path = _latest_profile("_pmc_extractor.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if not _counters_match_source(d):
            return None  # counters of another source text: no traffic figure rather than a stale one
        k = d["kernels"][kernel]
        per_img = (2.0 * k["fetch_kb"] + k["write_kb"]) * 1024.0 * k.get("launches", 1) / d["images_per_launch"]
        return per_img * n_img
    except (OSError, KeyError, ValueError):
        return None


def valu_issue(kernel, n_img, launch_ms):
    """VALU-issue view of an extractor kernel: its vector instruction count per SIMD from the committed SQ counter pass
    (profiles/r*_pmc_fast.json, tools/pmc_fast.sh: SQ_INSTS_VALU in its own rocprofv3 --pmc run), scaled to this run's
    images per launch, against the issue slots of THIS run's launch duration: the fraction of a SIMD's vector issue
    the kernel fills if all its instructions were full-rate (1.9 cycles per wave instruction on gfx950) and if all were
    half-rate (3.4); the kernel's mix lies between.  None if the file is absent or describes another kernel."""
    path = _latest_profile("_pmc_fast.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("kernel") != "k_" + kernel or launch_ms <= 0 or not _counters_match_source(d):
            return None
        insts = d["valu_insts_per_simd"] * n_img / d["images_per_launch"]
        cyc = launch_ms * 1e-3 * 2.4e9
        c = d["issue_cycles_per_valu_inst"]
        # round 6 (tools/fast_issue.py): the loops' instruction classes weighted by measured trip counts give the half-rate
        # share of the kernel's vector instructions, hence ONE issue fraction instead of the bracket below
        half = (d.get("instruction_classes") or {}).get("valu_half_rate_share")
        frac = insts * ((1 - half) * c["full_rate"] + half * c["half_rate"]) / cyc if half is not None else None
        return {"issue_fraction": frac, "valu_half_rate_share": half,
                "wave_cycle_accounting": {k: v for k, v in (d.get("wave_cycle_accounting") or {}).items() if isinstance(v, float)},
                "valu_insts_per_simd_per_launch": insts, "launch_cycles_at_2p4_ghz": cyc,
                "valu_insts_per_cycle_per_simd": insts / cyc,
                "issue_fraction_if_all_full_rate": insts * c["full_rate"] / cyc,
                "issue_fraction_if_all_half_rate": insts * c["half_rate"] / cyc,
                "lane_fill": d.get("lane_fill"),
                "source": os.path.relpath(path, os.path.dirname(os.path.abspath(__file__))) + " (" + d["source"] + ")",
                "note": "the kernel moves its algorithmic bytes once (traffic / algorithmic bytes ~ 1.1) and is bound by "
                        "vector instruction issue and resident wavefronts, not by HBM: `frac` above is reported against "
                        "the HBM peak because SURVEY 8(d) prices this kernel in bytes"}
    except (OSError, KeyError, TypeError, ValueError):
        return None


def mfma_busy():
    """SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES of k_lba_schur from the committed counter pass
    (profiles/r*_pmc_lba_schur.json, collected by tools/pmc_lba_schur.sh in its own rocprofv3 --pmc run); None if
    absent."""
    path = _latest_profile("_pmc_lba_schur.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return d if _counters_match_source(d) else None  # counters of another lba.hip are not evidence about this one
    except (OSError, TypeError, ValueError):
        return None


def cpu_baseline(P, lba_problems, lba_every, budget_s=20.0):
    """The same chain on the host with the CPU oracle, threaded like the reference: 2 threads for
    extraction, the tracking thread for the rest, and a LocalMapping thread running one LBA per
    `lba_every` frames concurrently."""
    import queue
    from tests import oracle_lib
    from vieo_slam_amd import synth_scene as sc
    from vieo_slam_amd.ba_types import POSE_OBS_DTYPE, SBP_CAMERA_DTYPE
    try:
        path = oracle_lib.build(native=True)
    except Exception:
        path = oracle_lib.build(native=False)
    orc = oracle_lib.Oracle(path)
    ex = [orc.extractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH) for _ in range(2)]
    cams = P.d_cams.download(SBP_CAMERA_DTYPE, (P.B,))
    xyz = P.d_xyz.download(np.float32, (P.B, 2 * P.cap, 3))
    cap = P.cap

    def obs_from(mp, b, k, ur):
        idx = np.nonzero(mp >= 0)[0]
        o = np.zeros(len(idx), POSE_OBS_DTYPE)
        o["Xw"] = xyz[b][mp[idx]]
        o["u"], o["v"], o["ur"] = k["x"][idx], k["y"][idx], ur[idx]
        o["inv_sigma2"] = P.inv_sigma2[k["octave"][idx]]
        return o, idx

    lq = queue.Queue()

    def local_mapping():
        while True:
            i = lq.get()
            if i is None:
                return
            orc.local_ba_vio(*lba_problems[i % len(lba_problems)])
    lm_thread = threading.Thread(target=local_mapping)
    if lba_every > 0 and lba_problems:
        lm_thread.start()
    n_done, t0 = 0, time.perf_counter()
    while n_done < P.B:
        b = n_done
        out = [None, None]

        def run(c):
            out[c] = ex[c](P.imgs_host[b, c])
        th = [threading.Thread(target=run, args=(c,)) for c in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        (_, k1, d1), (_, kr, dr) = out
        ur, _ = orc.stereo_match(ex[0], ex[1], k1, d1, kr, dr, sc.BASELINE, sc.BF)
        n0 = int(P.f1_host[b]["base"]["n_obs"] * 0 + np.count_nonzero(np.any(P.pts_host[b]["desc"] != 0, axis=1)))
        q1 = orc.sbp_project_last_frame(P.pts_host[b][:n0], cams[b:b + 1])
        _, a1 = orc.search_by_projection(0, q1, k1, ur, d1, None, BOUNDS)
        mp = np.where(a1 >= 0, a1, -1)
        o1, i1 = obs_from(mp, b, k1, ur)
        F1 = P.f1_host[b:b + 1].copy()
        F1[0]["base"]["n_obs"] = len(o1)
        r1, ol1 = orc.pose_optimization_vio(F1, o1)
        mp[i1[ol1 != 0]] = -1
        taken = (mp >= 0).astype(np.uint8)
        _, a2 = orc.search_by_projection(1, P.q2_host[b][:n0], k1, ur, d1, taken, BOUNDS, nn_ratio=0.8)
        mp = np.where(a2 >= 0, cap + a2, mp)
        o2, _ = obs_from(mp, b, k1, ur)
        F2 = F1.copy()
        F2[0]["base"]["nav"] = r1["base"]["nav"]
        F2[0]["base"]["n_obs"] = len(o2)
        F2[0]["compute_marg"] = 1
        orc.pose_optimization_vio(F2, o2)
        n_done += 1
        if lba_every > 0 and lba_problems and n_done % lba_every == 0:
            lq.put(n_done // lba_every)
        if time.perf_counter() - t0 > budget_s and n_done >= 4:
            break
    if lm_thread.is_alive() or (lba_every > 0 and lba_problems):
        lq.put(None)
        lm_thread.join()
    dt = time.perf_counter() - t0
    return {"value": n_done / dt, "unit": "frames/s", "cores": 3, "kind": "port",
            "sample_short": "%d stereo frames of the timed batch + 1 local BA per %d frames, CPU oracle -O3 -march=native, threads as in "
                            "the reference (2 extraction + tracking + LocalMapping), %.0f s" % (n_done, lba_every, dt),
            "sample": "%d of the benchmark's stereo frames through the same chain with the CPU oracle "
                      "(-O3 -march=native), threaded like the reference: extraction on 1 thread per "
                      "camera, stereo match / projection searches / 2x PoseOptimization on the tracking "
                      "thread, one LocalBundleAdjustmentNavStatePRV per %d frames on a LocalMapping thread (nproc=%d)"
                      % (n_done, lba_every, os.cpu_count())}


README_FRONTEND_MS = {"value": 35.0, "source": "/root/reference/README.md:60 'Final MH05 mean time cost per frame of "
                      "frontend(ms): 43.X -> 11.X -- 35.X' (i9-14900HX virtual box, 16 cores, 2024/9/9); the reference's own "
                      "published figure for the front end of this configuration, NOT measured here"}


README_FRONTEND_DIST_MS = {"value": 43.0, "source": "/root/reference/README.md:44-46,60: the same line's first figure, 43.X ms, is the "
                           "front end of the DEFAULT (distorted stereo, EuRoC_VIO_dist*.yaml) MH05 configuration; NOT measured here"}


def _rig_tracker_inputs(fe, fr0, mps, case):
    """mLastFrame / the local map of the one-call rig tracker from a stage-by-stage frame 0 (as tests/test_tracker_rig.py)"""
    from vieo_slam_amd.map_point import FRUSTUM_POINT_DTYPE
    pts = fe.last_frame_points(fr0, mps)
    has = mps["key_mp"] >= 0
    pts["reserved"][has, 0] = mps["first_key"][mps["key_mp"][has]] + 1
    z = fr0.fe["group_p3d"][np.nonzero(fr0.fe["group_good"])[0]][:, 2].astype(np.float32)
    last_depth = np.full(fr0.N, np.inf, np.float32)
    last_depth[has] = z[mps["key_mp"][has]]
    _, P = fe._frustum(np.eye(3, 4), mps, case["pose0"])
    return pts, last_depth, np.ascontiguousarray(P, FRUSTUM_POINT_DTYPE), mps["first_key"].astype(np.int32)


def rig_single_stream(rig, n_cams, nfeat, seed, n_cases=3, reps=10):
    """One-call tracker of a distorted camera rig (vieo_tracker_create_rig + vieo_track_frame): ms per rig frame over
    `n_cases` rendered frame pairs x `reps` calls each (the last frame's map points come from a stage-by-stage frame 0,
    untimed).  There is no rig SEQUENCE driver (map management for rigs is the caller's): this is the tracking call's
    latency on single frames, the figure the rectified replay reports as ms_per_frame_tracking_call."""
    from vieo_slam_amd import synth_ba
    from vieo_slam_amd import synth_scene as sc
    from vieo_slam_amd.pipeline_rig import RigFrontEnd
    from vieo_slam_amd.tracker import Tracker, rig_params
    scene = sc.RigScene(seed, rig, n_cams)
    fe = RigFrontEnd(scene, nfeat)
    ms, gpu, errs, keys, m1, m2, inl = [], [], [], [], [], [], []
    trk = None
    for i in range(n_cases):
        case = sc.make_rig_tracking_case(seed + 10 * i, scene)
        fr0 = fe.make_frame(case["images0"])
        mps = fe.make_map_points(fr0, case["pose0"][2], case["pose0"][3])
        pts, ld, P, alias = _rig_tracker_inputs(fe, fr0, mps, case)
        if trk is None:
            prm, rg = rig_params(scene, nfeat, max_local_points=4096)
            trk = Tracker(prm, rg)
        nav = case["vio"][0]["nav_last"]
        call = lambda: trk.track(None, None, case["imu_samples"], 0.0, case["dt_frame"], nav, nav, None, pts, ld, P, mps["desc"],
                                 alias, i + 1, images=case["images1"])
        call()
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            o, v = call()
            t.append(1e3 * (time.perf_counter() - t0))
        ms.append(float(np.median(t))), gpu.append(float(o["ms_gpu"]))
        errs.append(float(synth_ba.pose_error(o["second"]["base"]["nav"], case["truth"])[0]))
        keys.append(int(o["n_keys"])), m1.append(int(o["n_matches_last"])), m2.append(int(o["n_matches_local"]))
        inl.append(int(o["second"]["base"]["n_inliers"]))
    trk.close()
    v = float(np.mean(ms))
    return {"config": "%d-camera %s rig, %d features per camera, %dx%d" % (n_cams, rig, nfeat, scene.W, scene.H),
            "ms_per_rig_frame": v, "rig_frames_per_s": 1e3 / v, "ms_per_rig_frame_gpu": float(np.mean(gpu)),
            "ms_per_case": ms, "host_syncs_per_frame": 1, "mean_keys_per_frame": float(np.mean(keys)),
            "mean_matches_last_frame": float(np.mean(m1)), "mean_matches_local_map": float(np.mean(m2)),
            "mean_pose_inliers": float(np.mean(inl)), "max_position_error_vs_truth_m": float(max(errs)),
            "path": "ONE vieo_track_frame call per rig frame: one copy up, ExtractORB x n_cams (one batch launch chain) + IMU "
                    "pre-integration beside it, ComputeStereoFishEyeMatches on the device (knn-2 of every camera pair, pair "
                    "triangulation, FillMatchesFromPair as a speculative-parallel walk, all-camera re-triangulation), "
                    "PredictNavStateByIMU, SearchByProjection(last frame, camera loop) -> PoseOptimization(rig) -> isInFrustum + "
                    "queries -> SearchByProjection(local map) -> PoseOptimization(rig, marg), one copy back, ONE synchronisation"}


def rig_frontend_batch(rig="kb8", n_cams=4, nfeat=1500, seed=300, n_frames=256, steps=5):
    """BASELINE configs[3] shape, batched: B rig frames device-resident through the WHOLE tracking step
    (pipeline_rig_batch.RigFramePipeline: ExtractORB x B x n_cams in one batch, ComputeStereoFishEyeMatches of the batch as
    five launches, both projection searches with the camera loop, both rig pose optimisations; no host round trip), and
    the roofline of the stereo stage's dense kernel: k_knn2 = the Hamming brute-force search north_star names."""
    import ctypes
    from vieo_slam_amd import synth_ba
    from vieo_slam_amd import synth_scene as sc
    from vieo_slam_amd._lib import DeviceBuffer, check, lib
    from vieo_slam_amd.pipeline_rig_batch import RigFramePipeline
    L = lib()
    scene = sc.RigScene(seed, rig, n_cams)
    cases = [sc.make_rig_tracking_case(seed + 10 * i, scene) for i in range(3)]
    P = RigFramePipeline(scene, cases, nfeat, n_frames, seed=seed)
    P.step()
    P.sync()
    P.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        P.step()
    P.sync()
    dt = (time.perf_counter() - t0) / steps
    stage = P.stage_ms_all()
    R = P.results()
    hdr = R["hdr"]
    cnt = P.d_cnt.download(np.int32, (n_frames, n_cams, 2))
    errs = [float(synth_ba.pose_error(R["r2"][b]["base"]["nav"], P.truth[b])[0]) for b in range(n_frames)]
    # ---- k_knn2 alone between two events on the pipeline's stream
    st, cap, n_img = P.stream, P.cap, n_frames * n_cams
    n_pairs = n_cams * (n_cams - 1) // 2
    d_idx, d_dist = DeviceBuffer(n_frames * n_pairs * cap * 8), DeviceBuffer(n_frames * n_pairs * cap * 8)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    check(L.vieo_event_create(ctypes.byref(e0))), check(L.vieo_event_create(ctypes.byref(e1)))
    K = 20
    for k in range(K + 2):
        if k == 2:
            check(L.vieo_event_record(e0, st))
        check(L.vieo_hamming_knn2_rig_batch_device(P.d_desc.ptr, P.d_cnt.ptr, cap, n_cams, n_frames, d_idx.ptr, d_dist.ptr, st))
    check(L.vieo_event_record(e1, st))
    P.sync()
    ms = ctypes.c_float()
    check(L.vieo_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    launch_ms = ms.value / K
    alg, ops = 0.0, 0.0
    for f in range(n_frames):
        for i in range(n_cams - 1):
            for j in range(i + 1, n_cams):
                ni, mi, nj, mj = cnt[f, i, 0], cnt[f, i, 1], cnt[f, j, 0], cnt[f, j, 1]
                if mi >= ni or mj >= nj:
                    continue
                nq, nt = float(ni - mi), float(nj - mj)
                alg += (nq + nt) * 32 + nq * 16  # both descriptor matrices once + two (index, distance) pairs per query
                ops += nq * nt * 8
    gbs = alg / (launch_ms * 1e-3) / 1e9
    L.vieo_event_destroy(e0), L.vieo_event_destroy(e1)
    d_idx.free(), d_dist.free()
    P.close_all()
    return {
        "config": "BASELINE configs[3] shape: %d frames of a %d-camera %s rig, %d features per camera, device-resident, the whole "
                  "tracking step: ExtractORB x %d images in one batch, ComputeStereoFishEyeMatches of the batch (5 launches), "
                  "SearchByProjection(last frame) with the camera loop, PoseOptimization(VIO, rig), isInFrustum + queries, "
                  "SearchByProjection(local map), PoseOptimization(VIO, rig, marg); no host round trip; noise replicas of 3 "
                  "rendered rig frame pairs" % (n_frames, n_cams, rig, nfeat, n_img),
        "rig_frames_per_s": n_frames / dt, "ms_per_step": 1e3 * dt, "camera_images_per_s": n_img / dt,
        "stage_ms_per_step": {k: float(np.mean([s_[k] for s_ in stage])) for k in P.STAGES},
        "mean_keys_per_camera": float(cnt[:, :, 0].mean()), "mean_stereo_groups_per_frame": float(hdr[:, 0].mean()),
        "mean_pose_inliers": float(np.mean(R["r2"]["base"]["n_inliers"])), "median_position_error_vs_truth_m": float(np.median(errs)),
        "fill_matches_walk": {"rows_per_frame": float(hdr[:, 5].mean()), "wavefront_steps_per_frame": float(hdr[:, 6].mean()),
                              "note": "FillMatchesFromPair's order-dependent group tables on the device: rows applied per "
                                      "speculative-parallel step = rows / steps (sequential form: 1)"},
        "roofline_knn2": {"bound": "hbm", "kernel": "k_knn2_mfma (cv::BFMatcher knnMatch k = 2 of every camera pair of every frame, one "
                                                    "launch; Hamming = |a| + |b| - 2 a.b with the bits as int8 0 / 1 on "
                                                    "v_mfma_i32_32x32x32_i8, 256 queries per workgroup as resident B fragments, "
                                                    "train tiles expanded once per workgroup in LDS; VIEO_KNN2_MFMA=0: the "
                                                    "popcount kernel k_knn2)" if os.environ.get("VIEO_KNN2_MFMA", "1") != "0" else
                                                    "k_knn2 (popcount form: query rows in registers, train rows through LDS tiles)",
                          "mfma": {"bound": "mfma", "int8_ops_per_launch": ops / 8 * 256 * 2, "achieved": ops / 8 * 256 * 2 / (launch_ms * 1e-3) / 1e12,
                                   "peak": 4404.0, "unit": "TOP/s", "frac": ops / 8 * 256 * 2 / (launch_ms * 1e-3) / 1e12 / 4404.0,
                                   "peak_source": "cdna_hip_programming.md: i8 32x32 MFMA, 4404 TOPS measured floor"},
                          "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                          "algorithmic_bytes_per_launch": alg, "avg_launch_ms": launch_ms, "traffic": None,
                          "xor_popcount_word_ops_per_launch": ops, "word_ops_per_s": ops / (launch_ms * 1e-3),
                          "note": "SURVEY 8d prices the search by its bytes ((N_q + N_t) 32 + N_q 16 per pair): at N = 1500 "
                                  "that is 120 KB against 18 M word operations, so the kernel is bound by the integer issue "
                                  "rate (word_ops_per_s), not by HBM; the fraction of the byte roofline is reported as asked"}}


def schur_useful_flops(window):
    """SURVEY 8d's sparsity-aware count of the Schur reduction of ONE LM trial of a local-BA window: per landmark with k free
    observers 60 (3x3 inverse) + 108 k (W Hll^-1) + 216 k (k + 1) / 2 (the landmark's blocks of Hpp) + 36 k (gradient)."""
    kfs, obs = window[1], window[4]
    free = kfs["fixed"] == 0
    kf = obs["kf"] & 0xFFFFFF
    k = np.bincount(obs["mp"][free[kf]], minlength=len(window[2])).astype(np.float64)
    k = k[np.bincount(obs["mp"], minlength=len(window[2])) > 0]
    return float(np.sum(60 + 108 * k + 216 * k * (k + 1) / 2 + 36 * k))


LBA_LAG = 8  # frames between a key frame and the write-back of its local BA in the single_stream leg (0: inline).
# (8 since round 5: with frames near 1.05 ms a 5 ms solve plus its window assembly does not fit into 6 of them -- the
# tracker waited 0.06 ms per frame for the write-back; the figure with 6 is reported beside it)


def single_stream_leg(seq, n_frames):
    """The sequential replay on the C-ABI (see the module docstring).  Headline: examples/replay_main, the replay as a
    C++ program WITHOUT Python -- vieo_track_frame per frame (one chain of launches, one synchronisation), local BA per
    key frame on the LocalMapping thread, the map on the host in C++.  Beside it `drop_in`: examples/dropin_replay, the
    same replay driven ONLY through the entries behind the reference's own class members (the path that changes nothing
    in Tracking.cc), and the same replay driven from Python stage by stage.  The oracle's run of the same replay (a worker
    process of the cpu_baseline leg) fills in the ATE."""
    import subprocess
    import tempfile
    from tools.write_sequence import write_sequence
    from vieo_slam_amd import replay, synth_ba
    from vieo_slam_amd.ba_types import NAVSTATE_DTYPE
    for k in range(n_frames):
        seq.images(k)  # rendering is not part of either timing
    exe = os.path.join(ROOT, "examples", "replay_main")
    exe_d = os.path.join(ROOT, "examples", "dropin_replay")
    if not os.path.exists(exe) or not os.path.exists(exe_d):
        raise RuntimeError("examples/replay_main / dropin_replay missing: run __graft_entry__.build()")

    def run(cmd):
        return json.loads(subprocess.check_output(cmd, timeout=1800).decode().strip().splitlines()[-1])
    with tempfile.TemporaryDirectory() as tmp:
        path, traj = os.path.join(tmp, "seq.vseq"), os.path.join(tmp, "traj.bin")
        write_sequence(path, seq.seed, n_frames, seq)
        runs = [run([exe, path, traj, "--warmup", "16", "--quiet", "--lba-lag", str(LBA_LAG), "--prefetch", "1"]) for _ in range(3)]
        r_plain = run([exe, path, traj + ".plain", "--warmup", "16", "--quiet", "--lba-lag", str(LBA_LAG), "--prefetch", "0"])
        plain_identical = open(traj, "rb").read() == open(traj + ".plain", "rb").read()
        r_lag6 = run([exe, path, traj + ".lag6", "--warmup", "16", "--quiet", "--lba-lag", "6", "--prefetch", "1"])
        # (the trajectory of ANY run with this lag is the same: the lag, not the timing, decides which map a frame sees)
        th = np.fromfile(traj, NAVSTATE_DTYPE)
        r_inline = run([exe, path, traj + ".inline", "--warmup", "16", "--quiet"])
        d_runs = [run([exe_d, path, traj + ".dropin", "--warmup", "16", "--quiet", "--lba-lag", str(LBA_LAG)]) for _ in range(3)]
        td = np.fromfile(traj + ".dropin", NAVSTATE_DTYPE)
        d_host = run([exe_d, path, traj + ".dropin0", "--warmup", "16", "--quiet", "--lba-lag", str(LBA_LAG), "--resident", "0"])
        conc = concurrent_trackers_leg(path)
    r = runs[int(np.argsort([x["ms_per_frame"] for x in runs])[1])]  # the median run's record; the figure below is the MEAN
    mean_ms = float(np.mean([x["ms_per_frame"] for x in runs]))
    d = d_runs[int(np.argsort([x["ms_per_frame"] for x in d_runs])[1])]
    d_mean = float(np.mean([x["ms_per_frame"] for x in d_runs]))
    Rs = replay.Replay(seq, replay.HipStages())
    Rs.run(min(12, n_frames))
    n_py = min(n_frames, 100)
    Rs = replay.Replay(seq, replay.HipStages())
    t0 = time.perf_counter()
    ts = Rs.run(n_py)
    t_s = time.perf_counter() - t0
    err = max(synth_ba.pose_error(th[k], seq.truth(k))[0] for k in range(n_frames))
    drop_in = {
        "ms_per_frame": d_mean, "frames_per_s": 1e3 / d_mean, "ms_per_frame_all_runs": [x["ms_per_frame"] for x in d_runs],
        "ms_per_frame_last_200": float(np.mean([x["ms_per_frame_last_200"] for x in d_runs])),
        "ms_per_frame_host_pointer_form": d_host["ms_per_frame"], "stage_ms_per_frame": d["stage_ms_per_frame"],
        "sequential_calls_per_frame": d["sequential_calls_per_frame"], "lba_lag_frames": LBA_LAG,
        "ms_per_local_ba_mean": d["ms_per_local_ba"], "key_frames": d["key_frames"], "map_points": d["map_points"],
        "ate_vs_oracle_m": None, "max_position_difference_vs_oracle_m": None,
        "max_position_difference_vs_one_call_path_m": float(np.linalg.norm(td["p"] - th["p"], axis=1).max()),
        "reference_readme_anchor": dict(README_FRONTEND_MS, ratio_to_this_ms_per_frame=README_FRONTEND_MS["value"] / d_mean),
        "path": "examples/dropin_replay (C++, no Python): the frame member by member, as Tracking calls them -- "
                "ORBextractor::operator() per camera on its own host thread (vieo_orb_extract), Frame::ComputeStereoMatches "
                "(vieo_stereo_match_rectified_resident), FrameBase::PreIntegration (vieo_imu_preintegrate_batch), "
                "PredictNavStateByIMU on the host, ORBmatcher::SearchByProjection(Frame&, const Frame&) "
                "(vieo_search_by_projection_last_frame_resident), Optimizer::PoseOptimization (vieo_pose_optimization_vio), "
                "Frame::isInFrustum (vieo_is_in_frustum_batch), SearchByProjection(Frame&, vector<MapPoint*>&) "
                "(vieo_search_by_projection_resident), PoseOptimization(bComputeMarg); LocalBundleAdjustmentNavStatePRV on the "
                "LocalMapping thread.  The resident frame: keys / descriptors / pyramid / uright / window grid stay in the "
                "extractor handles the Frame points at; ms_per_frame_host_pointer_form re-uploads them in every call "
                "(round 4's shims).  Nothing in Tracking.cc / LocalMapping.cc changes for this path (shim/*.cc)."}
    return {
        "frames": n_frames, "local_bas": r["local_bas"],
        "ms_per_frame": mean_ms, "frames_per_s": 1e3 / mean_ms,
        "ms_per_frame_last_200": float(np.mean([x["ms_per_frame_last_200"] for x in runs])),
        "ms_per_frame_median_frame": r["ms_per_frame_median"], "ms_per_frame_p99_frame": r["ms_per_frame_p99"],
        "lba_windows": r["lba_windows"],
        "local_ba": "beside tracking (src/LocalMapping.cc:113-139): solved on its own host thread on the bundle-adjustment "
                    "stream, write-back before the %d-th frame after the key frame; the oracle replay it is compared with "
                    "applies the same lag" % LBA_LAG,
        "lba_lag_frames": LBA_LAG,
        "frame_pipelining": {
            "on": True, "ms_per_frame_without": r_plain["ms_per_frame"], "ms_per_frame_tracking_call_without": r_plain["ms_track_call"],
            "trajectory_bytes_identical_without": bool(plain_identical),
            "ms_per_frame_write_back_6_frames_behind": r_lag6["ms_per_frame"],
            "note": "vieo_track_input.next_left / next_right / next_imu: frame k + 1's ExtractORB x 2 + ComputeStereoMatches (third "
                    "stream) and its pre-integration (second stream) run beside frame k's searches and optimisations; the "
                    "next call adopts them.  A dataset player / a camera driver one frame ahead has the next frame in hand; "
                    "outputs are bit for bit those of the unpipelined calls"},
        "caller_ms_per_frame": r.get("caller_ms_per_frame"),
        "ms_per_frame_local_ba_inline": r_inline["ms_per_frame"], "vs_cpu_threaded": None,
        "ms_per_frame_all_runs": [x["ms_per_frame"] for x in runs],
        "ms_per_frame_tracking_call": float(np.mean([x["ms_track_call"] for x in runs])),
        "ms_per_frame_tracking_call_gpu": float(np.mean([x["ms_track_gpu"] for x in runs])),
        "ms_per_frame_without_local_ba": r["ms_frame_without_local_ba"], "ms_per_local_ba_mean": r["ms_per_local_ba"],
        # the vieo_local_bundle_adjustment_vio call alone (as the Python replays time it); the key-frame pair's pre-integration
        # (LocalMapping::ProcessNewKeyFrame's in the reference) and the whole LocalMapping job (flattening the window +
        # pre-integration + the call + MapPoint::UpdateNormalAndDepth) beside it
        "ms_per_key_frame_preintegration": r.get("ms_per_key_frame_preintegration"),
        "ms_per_local_mapping_job": r.get("ms_per_local_mapping_job"),
        "host_syncs_per_frame": 1, "frames_with_the_wider_search_window": r["widened"],
        "key_frames": r["key_frames"], "map_points": r["map_points"],
        "ate_vs_oracle_m": None, "max_position_difference_vs_oracle_m": None, "vs_cpu_single_stream": None,
        "max_position_error_vs_truth_m": float(err),
        "max_position_difference_vs_stage_by_stage_m": float(np.linalg.norm(th["p"][:n_py] - ts["p"], axis=1).max()),
        "reference_readme_anchor": dict(README_FRONTEND_MS, ratio_to_this_ms_per_frame=README_FRONTEND_MS["value"] / mean_ms,
                                        ms_per_frame_for_30x=README_FRONTEND_MS["value"] / 30.0,
                                        # the README figure is the FRONT END's time per frame (the tracking thread; local
                                        # mapping runs beside it in the reference): like for like it is this replay's
                                        # tracking call, not its whole loop with the local BAs in line
                                        ratio_to_this_tracking_call=README_FRONTEND_MS["value"] / float(np.mean([x["ms_track_call"] for x in runs]))),
        "drop_in": drop_in,
        "concurrent_trackers": dict(conc, note="examples/replay_main --trackers N: N independent sequences (each its own host "
                                               "thread, vieo_tracker, map, LocalMapping thread) on this one GPU, 60 frames each; "
                                               "frames_per_s_all_trackers = aggregate, ms_per_frame_latency_* = a frame's wall time "
                                               "inside its own tracker's loop; trajectories asserted identical"),
        "stage_by_stage": {"ms_per_frame": 1e3 * t_s / (n_py - 1), "frames": n_py,
                           "latency_ms_per_frame_tracking_median": float(np.median(Rs.stats["ms_frames"]))},
        "path": "examples/replay_main (C++, no Python): per frame ONE vieo_track_frame call = one copy up from pinned "
                "memory, IMU pre-integration on a second stream beside extraction x2 -> stereo, PredictNavStateByIMU on the "
                "device, SearchByProjection(last frame) -> PoseOptimization -> isInFrustum + queries from the optimised pose "
                "in HBM -> SearchByProjection(local map) -> PoseOptimization(marg), one copy back, ONE host synchronisation; "
                "per key frame (every 10 frames) vieo_imu_preintegrate_batch + vieo_local_bundle_adjustment_vio (on the "
                "LocalMapping thread) + vieo_update_normal_and_depth_batch; ms_per_frame is the MEAN of three runs of the wall "
                "time of the whole loop incl. the C++ map bookkeeping, the write-backs and any wait for a local BA that is not "
                "finished at its frame; stage_by_stage = the same replay driven from Python with one synchronous "
                "host-pointer call per stage",
    }, th, td


# ---------------------------------------------------------------- oracle replays in worker processes
def _oracle_replay_worker(task):
    """(spawned process, CPU only) renders a sequence and replays it on the CPU oracle: the checker of a sequence leg.
    task = (kind, seed, n_frames, lag, rig, n_cams, nfeat, oracle_so).  Returns the oracle's trajectory, the seconds its
    replay took, and the rendered images (the parent's HIP run then needs no second rendering)."""
    kind, seed, n, lag, rig, nc, nfeat, so = task
    from tests import oracle_lib
    from tests.replay_oracle import OracleRigStages, OracleStages, OracleVisionStages
    from vieo_slam_amd import replay, replay_modes as rm
    orc = oracle_lib.Oracle(so)
    if kind == "vio":
        seq = replay.Sequence(seed, n)
        R = replay.Replay(seq, OracleStages(orc), lba_lag=lag)
    elif kind == "vision":
        seq = replay.Sequence(seed, n)
        R = rm.VisionReplay(seq, OracleVisionStages(orc), lba_lag=lag)
    else:
        seq = rm.RigSequence(seed, n, rig, nc)
        R = rm.RigReplay(seq, OracleRigStages(orc, nfeat, nc), nfeat, lba_lag=lag)
    imgs = [seq.images(k) for k in range(n)]
    t0 = time.perf_counter()
    traj = R.run(n)
    dt = time.perf_counter() - t0
    return {"task": task[:7], "traj": traj.tobytes(), "seconds": dt, "images": imgs, "lba": R.stats["lba"],
            "key_frames": len(R.kfs), "map_points": len(R.mp_X),
            "stats": {"n_matches": [tuple(int(x) for x in m) for m in R.stats["n_matches"]],
                      "n_inliers": [int(x) for x in R.stats["n_inliers"]]}}


class OracleReplays:
    """The oracle's replays of the sequence legs, started in spawned worker processes right after the timed region and
    collected when a leg needs them (the GPU box has 256 hardware threads; the workers never touch the GPU)."""

    def __init__(self, tasks):
        import multiprocessing as mp
        from tests import oracle_lib
        try:
            so = oracle_lib.build(native=True)
        except Exception:
            so = oracle_lib.build(native=False)
        self.pool = mp.get_context("spawn").Pool(min(len(tasks), 6))
        self.res = {name: self.pool.apply_async(_oracle_replay_worker, (t + (so,),)) for name, t in tasks.items()}

    def get(self, name):
        from vieo_slam_amd.ba_types import NAVSTATE_DTYPE
        r = self.res[name].get(timeout=1800)
        r["traj"] = np.frombuffer(r["traj"], NAVSTATE_DTYPE).copy()
        return r

    def close(self):
        self.pool.terminate()


def sequence_leg(kind, orc, n, lag, rig=None, nc=None, nfeat=None, seed=1):
    """A SEQUENCE replay of one of the other BASELINE configurations with the local BA in the loop
    (vieo_slam_amd/replay_modes.py): map growth from the stereo groups / stereo depths, a key frame every 10 frames, its
    local BA applied `lag` frames later, one vieo_track_frame call per frame; ATE against the oracle's replay of the same
    sequence (run in a worker process)."""
    from vieo_slam_amd import replay, replay_modes as rm, synth_ba
    if kind == "vision":
        seq = replay.Sequence(seed, n)
    else:
        seq = rm.RigSequence(seed, n, rig, nc)
    for k, im in enumerate(orc["images"]):
        seq._img[k] = im
    mk = (lambda pf: rm.VisionTrackerReplay(seq, rm.HipVisionStages(), lba_lag=lag, prefetch=pf)) if kind == "vision" else \
        (lambda pf: rm.RigTrackerReplay(seq, rm.HipRigStages(nfeat, nc), nfeat, lba_lag=lag, prefetch=pf))
    R = mk(True)
    R.run(min(14, n))  # code objects, scratch buffers, the LocalMapping thread's arenas
    R.close()
    Rn = mk(False)  # without frame pipelining: the same outputs bit for bit, the extraction inside the call
    tn = Rn.run(n)
    Rn.close()
    ms_plain = np.array(Rn.stats["ms_chain"])
    R = mk(True)
    t0 = time.perf_counter()
    th = R.run(n)
    dt = time.perf_counter() - t0
    R.close()
    cpp = cpp_sequence_leg(kind, seq, n, lag, nfeat, th)
    to = orc["traj"]
    # 1e-4 is the bar "for the same inputs": up to the first frame whose integer decisions (matches, inliers) differ from
    # the oracle's run the two work on the same map; behind it a flipped decision has changed the inputs
    flip = replay.first_decision_flip(R.stats, orc["stats"])
    upto = n if flip is None else flip
    ms = np.array(R.stats["ms_chain"])
    shapes = np.array(R.stats.get("lba_shapes", [(0, 0, 0, 0)]))
    err = max(synth_ba.pose_error(th[k], seq.truth(k))[0] for k in range(n))
    return {
        "frames": n, "local_bas": R.stats["lba"], "key_frames": len(R.kfs), "map_points": int(len(R.mp_X)), "lba_lag_frames": lag,
        "ms_per_frame_tracking_call": float(ms[:, 0].mean()), "ms_per_frame_tracking_call_gpu": float(ms[:, 1].mean()),
        "ms_per_frame_tracking_call_last_half": float(ms[len(ms) // 2:, 0].mean()),
        "frame_pipelining": {"on": True, "ms_per_frame_tracking_call_without": float(ms_plain[:, 0].mean()),
                             "trajectory_bytes_identical_without": bool(th.tobytes() == tn.tobytes())},
        "ms_per_frame": cpp.get("ms_per_frame"), "cpp": cpp,
        "ms_per_frame_python_loop": 1e3 * dt / (n - 1),
        "ms_per_local_ba_mean": float(np.mean(R.stats["ms_lba"])) if R.stats["ms_lba"] else None,
        "lba_windows": {"mean_key_frames": float(shapes[:, 0].mean()), "max_key_frames": int(shapes[:, 0].max()),
                        "max_fixed_key_frames": int(shapes[:, 1].max()), "mean_points": float(shapes[:, 2].mean()),
                        "mean_observations": float(shapes[:, 3].mean())},
        "mean_matches_last_frame": float(np.mean([m[0] for m in R.stats["n_matches"]])),
        "mean_matches_local_map": float(np.mean([m[1] for m in R.stats["n_matches"]])),
        "mean_pose_inliers": float(np.mean(R.stats["n_inliers"])),
        "ate_vs_oracle_m": replay.ate_between(th, to),
        "max_position_difference_vs_oracle_m": float(np.linalg.norm(th["p"] - to["p"], axis=1).max()),
        "first_frame_with_a_different_integer_decision": flip,
        "max_position_difference_vs_oracle_until_then_m": float(np.linalg.norm(th["p"][:upto] - to["p"][:upto], axis=1).max()),
        "oracle_replay_frames_per_s": (n - 1) / orc["seconds"], "oracle_over_tracking_call": orc["seconds"] / (n - 1) * 1e3 / float(ms[:, 0].mean()),
        "max_position_error_vs_truth_m": float(err), "host_syncs_per_frame": 1,
        "note": "ms_per_frame_tracking_call = wall time of the ONE vieo_track_frame call per frame (the C entry's own clock); "
                "ms_per_frame = the WHOLE loop of examples/replay_modes (C++, no Python: map bookkeeping on the host, the "
                "local BA on its LocalMapping thread; median of 3 runs, details under cpp); ms_per_frame_python_loop = the same "
                "replay driven from Python (numpy map bookkeeping, local BAs issued from Python)",
    }


def cpp_sequence_leg(kind, seq, n, lag, nfeat, th):
    """The same sequence replay as a C++ program (examples/replay_modes.cc: rig / --vision): whole-loop ms per frame with
    frame pipelining, LocalMapping on its own host thread; its trajectory against the Python tracker replay's (th)."""
    import subprocess
    import tempfile
    from tools.write_sequence import write_rig_sequence, write_sequence
    from vieo_slam_amd.ba_types import NAVSTATE_DTYPE
    exe = os.path.join(ROOT, "examples", "replay_modes")
    if not os.path.exists(exe):
        return {"error": "examples/replay_modes missing: run __graft_entry__.build()"}
    try:
        with tempfile.TemporaryDirectory() as tmp:
            path, traj = os.path.join(tmp, "seq.vseq"), os.path.join(tmp, "traj.bin")
            if kind == "vision":
                write_sequence(path, seq.seed, n, seq)
            else:
                write_rig_sequence(path, seq, nfeat)
            cmd = [exe, path, traj, "--warmup", "14", "--quiet", "--lba-lag", str(lag), "--prefetch", "1"] + (["--vision"] if kind == "vision" else [])
            runs = [json.loads(subprocess.check_output(cmd, timeout=1800).decode().strip().splitlines()[-1]) for _ in range(3)]
            tc = np.fromfile(traj, NAVSTATE_DTYPE)
        r = sorted(runs, key=lambda x: x["ms_per_frame"])[1]
        d = np.linalg.norm(tc["p"] - th["p"], axis=1)
        return dict(r, ms_per_frame_runs=[x["ms_per_frame"] for x in runs],
                    max_position_difference_vs_python_tracker_replay_m=float(d.max()),
                    max_position_difference_vs_python_first_16_frames_m=float(d[:16].max()),
                    path="examples/replay_modes (C++, no Python): per frame ONE vieo_track_frame call, the next frame's extraction "
                         "queued beside it; per key frame the local BA on the LocalMapping thread, applied %d frames later" % lag)
    except Exception as e:  # (the Python leg's numbers stand on their own)
        return {"error": repr(e)}


def concurrent_trackers_leg(path, n_list=(8, 32), frames=60):
    """N independent sequences on ONE GPU: examples/replay_main --trackers N (every tracker its own host thread, streams,
    map and LocalMapping thread, all replaying the same file: identical trajectories are asserted by the program)."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "replay_main")
    out = {}
    for nt in n_list:
        line = subprocess.check_output([exe, path, "--frames", str(frames), "--trackers", str(nt), "--quiet", "--lba-lag", str(LBA_LAG)],
                                       timeout=900).decode().strip().splitlines()[-1]
        out["n%d" % nt] = json.loads(line)
    return out



def multi_gpu_legs(rank, world, dist, torch, reps=3):
    """SURVEY 8e "report both modes": besides the replica mode of the headline, what the driver's --gpus N runs measure
    in the same process group (and what --gpus 1 exercises with a one-rank communicator):
      sharded_local_ba / sharded_full_ba: ONE window / ONE full BA in System::FinalGBA's form (scale vertex) with its
          landmarks spread over the ranks, the reduced pose system summed by the in-library RCCL all-reduce on the
          bundle-adjustment stream (sharding.RcclComm -> vieo_rccl_*), rank 0's poses against the unsharded call on its
          own GPU, and the time of a torch.distributed all-reduce of the same size for scale;
      rig_replicas: BASELINE configs[3] -- a 4-camera distorted (KB8) rig chain, one sequence of frames per rank, no
          collective: extract x 4 -> ComputeStereoFishEyeMatches -> SearchByProjection x 2 -> PoseOptimization x 2."""
    from vieo_slam_amd import sharding, synth_ba
    from vieo_slam_amd import synth_scene as sc
    from vieo_slam_amd._lib import DeviceBuffer
    from vieo_slam_amd.optimizer import Optimizer
    from vieo_slam_amd.pipeline_rig import RigFrontEnd

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def mx(v):
        return sharding.max_over_ranks(dist, v, device="cuda")

    out = {"rccl_ranks": world}
    comm = sharding.RcclComm(rank, world)
    try:
        for name, gba in (("sharded_local_ba", False), ("sharded_full_ba", True)):
            if gba:
                win = synth_ba.make_lba_vio_problem(901, n_local=60, n_fixed=1, n_points=6000, anchors=30, span=5)[:6]
                win = (win[0], win[1], (win[2] / np.float32(1.03)).astype(np.float32)) + tuple(win[3:])
            else:
                win = synth_ba.make_lba_vio_problem(900, n_local=10, n_fixed=40, n_points=2000)[:6]
            shard, mine = sharding.shard_window(win, rank, world)
            n = Optimizer.sharded_buffer_doubles([shard])
            buf = DeviceBuffer(8 * n)

            def call():
                if gba:
                    return Optimizer.GlobalBundleAdjustmentNavStatePRVSharded(shard, buf.ptr, n, None, 5, True, comm=comm.handle,
                                                                              bScaleOpt=True)
                return Optimizer.LocalBundleAdjustmentNavStatePRVSharded([shard], buf.ptr, n, None, comm=comm.handle)[0]
            times = []
            for i in range(reps + 1):
                barrier()
                t = time.perf_counter()
                r = call()
                barrier()
                if i:
                    times.append(mx(time.perf_counter() - t))
            nf = int((win[1]["fixed"] == 0).sum())
            n_sys = (6 * nf + (1 if gba else 0)) * (6 * nf + 1 + (1 if gba else 0)) + 42 * nf + ((6 * nf + 2) if gba else 0)
            tt = torch.zeros(n_sys, dtype=torch.float64, device="cuda")
            us = None
            if world > 1:
                for _ in range(5):
                    dist.all_reduce(tt)
                barrier()
                t = time.perf_counter()
                for _ in range(50):
                    dist.all_reduce(tt)
                torch.cuda.synchronize()
                us = mx(time.perf_counter() - t) / 50 * 1e6
            leg = {"ms_per_call": 1e3 * float(np.mean(times)), "key_frames": len(win[1]), "free_key_frames": nf,
                   "points_total": len(win[2]), "points_this_rank": len(mine), "observations_total": len(win[4]),
                   "lm_trials": int((r[2] if gba else r[3])["lm_trials"]),
                   "allreduce_doubles_per_trial": n_sys,
                   "torch_allreduce_of_that_size_us": us,
                   "exchange": "in-library ncclAllReduce(sum, f64) on the bundle-adjustment stream (vieo_rccl_*), one per LM "
                               "trial + one of 4 scalars"}
            if rank == 0:
                if gba:
                    ref = Optimizer.GlobalBundleAdjustmentNavStatePRV(win[0], win[1], win[2], win[4], win[5], 5, True, bScaleOpt=True)
                    t = time.perf_counter()
                    Optimizer.GlobalBundleAdjustmentNavStatePRV(win[0], win[1], win[2], win[4], win[5], 5, True, bScaleOpt=True)
                    leg["recovered_scale"], leg["recovered_scale_unsharded"] = r[3], ref[3]
                else:
                    ref = Optimizer.LocalBundleAdjustmentNavStatePRV(*win)
                    t = time.perf_counter()
                    Optimizer.LocalBundleAdjustmentNavStatePRV(*win)
                leg["ms_per_call_unsharded_one_gpu"] = 1e3 * (time.perf_counter() - t)
                leg["max_pose_difference_vs_unsharded"] = float(max(max(synth_ba.pose_error(ref[0][k], r[0][k]))
                                                                    for k in range(len(win[1]))))
            barrier()
            buf.free()
            out[name] = leg
    finally:
        comm.close()
    # ---- configs[3]: rig replicas, one sequence of rig frames per rank through the ONE-CALL tracker (no collective)
    barrier()
    leg = rig_single_stream("kb8", 4, 1500, 300 + rank, n_cases=3, reps=5)
    barrier()
    ms_max = mx(leg["ms_per_rig_frame"])
    out["rig_replicas"] = {"config": "BASELINE configs[3]: 4-camera KB8 rig, 1500 features per camera, one sequence per rank, one "
                                     "vieo_track_frame call per rig frame (vieo_tracker_create_rig)",
                           "rig_frames_per_s_all_ranks": world * 1e3 / ms_max, "ms_per_rig_frame": ms_max,
                           "ms_per_rig_frame_gpu_rank0": leg["ms_per_rig_frame_gpu"],
                           "max_position_error_vs_truth_m_rank0": leg["max_position_error_vs_truth_m"]}
    # what the driver's 1 -> 8 table needs from THIS run: per mode the figure that scales (replica modes: x N expected;
    # sharded modes: ms per call, whatever is measured -- SURVEY 8e expects the latency-bound all-reduce to cap them)
    out["speedup_vs_n1_expected_inputs"] = {
        "n_gpus": world,
        "replicas_headline": "value of this line (frames/s, weak scaling: expected N x the N = 1 value)",
        "rig_replicas_frames_per_s": out["rig_replicas"]["rig_frames_per_s_all_ranks"],
        "sharded_local_ba_ms_per_call": out.get("sharded_local_ba", {}).get("ms_per_call"),
        "sharded_full_ba_ms_per_call": out.get("sharded_full_ba", {}).get("ms_per_call"),
        "note": "divide the N-GPU figure by the N = 1 run's figure of the same key (ms: the other way round)"}
    return out


def pcie_leg(P, steps, warmup, lba=None):
    """The batched step with its images uploaded from pinned host memory on a copy stream, double-buffered: while
    step s runs on the pipeline's stream, step s+1's images travel over PCIe.  lba = (pool, run_lba, chunks): the step's
    local-BA windows are issued beside it exactly as in the headline (the `inclusive` leg: H2D + front end + LocalBA)."""
    import ctypes
    from vieo_slam_amd._lib import DeviceBuffer, check, lib
    L = lib()
    nbytes = P.imgs_host.nbytes
    pin = ctypes.c_void_p()
    check(L.vieo_host_alloc_pinned(ctypes.byref(pin), nbytes), "pinned")
    ctypes.memmove(pin.value, P.imgs_host.ctypes.data, nbytes)
    bufs = [P.d_img, DeviceBuffer(nbytes)]
    cs = ctypes.c_void_p()
    check(L.vieo_stream_create(ctypes.byref(cs)), "stream")
    ev_up = [ctypes.c_void_p(), ctypes.c_void_p()]
    ev_done = [ctypes.c_void_p(), ctypes.c_void_p()]
    for e in ev_up + ev_done:
        check(L.vieo_event_create(ctypes.byref(e)))
    d_img0 = P.d_img

    def upload(i):  # images of the step that will read buffer i
        check(L.vieo_stream_wait_event(cs, ev_done[i]))  # the step that last read this buffer has finished
        check(L.vieo_memcpy_h2d_async(bufs[i].ptr, pin, nbytes, cs))
        check(L.vieo_event_record(ev_up[i], cs))

    for i in range(2):
        check(L.vieo_event_record(ev_done[i], P.stream))
    upload(0)
    t0 = None
    futs = []
    for s_ in range(warmup + steps):
        if s_ == warmup:
            P.sync()
            check(L.vieo_stream_synchronize(cs))
            t0 = time.perf_counter()
        i = s_ % 2
        upload(1 - i)                                  # next step's images travel now
        check(L.vieo_stream_wait_event(P.stream, ev_up[i]))
        P.d_img = bufs[i]
        P.step()
        check(L.vieo_event_record(ev_done[i], P.stream))
        if lba and s_ >= warmup:
            futs += [lba[0].submit(lba[1], c) for c in lba[2]]
    for f in futs:
        f.result()
    P.sync()
    check(L.vieo_stream_synchronize(cs))
    dt = time.perf_counter() - t0
    P.d_img = d_img0
    check(L.vieo_stream_destroy(cs))
    check(L.vieo_host_free_pinned(pin))
    return {"value": P.B * steps / dt, "unit": "frames/s", "ms_per_step": 1e3 * dt / steps,
            "h2d_bytes_per_step": nbytes, "h2d_GBps_sustained": nbytes * (steps + 1) / dt / 1e9,
            "note": ("H2D + front end + LocalBA in one timed region: the step's images from pinned host memory on a copy "
                     "stream (double-buffered against the step's kernels) and the step's local-BA windows issued from the host "
                     "threads as in the headline") if lba else
                    ("front end only (no LocalBA threads), images from pinned host memory on a copy stream, "
                     "double-buffered against the step's kernels")}



# ---------------------------------------------------------------- the ONE line the driver parses
LINE_BUDGET = 4000   # characters; BENCH_r04's 19 KB line parsed, BENCH_r05's 27 KB line did not: stay far from the edge


def _num(v, digits=6):
    """Numbers of the compact line: floats to `digits` significant digits, numpy scalars to Python's."""
    if isinstance(v, (bool, np.bool_)):
        return bool(v)
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        v = float(v)
        return float("%.*g" % (digits, v)) if np.isfinite(v) else None
    return v


def _pick(d, keys):
    return {k: _num(d.get(k)) for k in keys if isinstance(d, dict) and k in d}


def failed_legs(out):
    """Names of the legs whose record is an {"error": ...}: a broken leg must be visible in the line, not buried."""
    bad = []
    for k, v in out.items():
        if isinstance(v, dict):
            if "error" in v:
                bad.append(k)
            bad += ["%s.%s" % (k, k2) for k2, v2 in v.items() if isinstance(v2, dict) and "error" in v2]
    return bad


def compact_line(out, detail_path=None):
    """The contract's JSON line from the full record: headline, config (names the workload, <= 300 characters), roofline
    of the dominant kernel, the MFMA roofline of the Schur contraction, the CPU baseline, the parity sample's verdicts and
    the single-stream figures.  Everything else (stage tables, rig / vision legs, drop_in, pcie, multi_gpu, notes) is the
    detail record: bench_detail.json beside this script, echoed to stderr."""
    c = out.get("config", {})
    line = {k: _num(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                           "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    wl = "BASELINE configs[2]: EuRoC MH05 stereo-VIO 752x480, 1200 feats; per frame ORBextractor x2 + ComputeStereoMatches + " \
         "SearchByProjection x2 + PoseOptimization(VIO) x2; + 1 LocalBundleAdjustmentNavStatePRV per %s frames; frames resident in HBM" \
         % c.get("lba_every", 10)
    line["config"] = dict(_pick(c, ("workload_id", "stereo_frames_per_gpu_per_step", "local_ba_windows_per_step", "hip_streams_per_gpu",
                                    "mean_keypoints_per_image", "ate_rmse_vs_truth_m")), workload=wl[:300])
    r = out.get("roofline") or {}
    line["roofline"] = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                 "avg_launch_ms"))
    if isinstance(r.get("alone"), dict):
        line["roofline"]["frac_alone"] = _num(r["alone"].get("frac"))
    if isinstance(r.get("valu_issue"), dict) and r["valu_issue"].get("issue_fraction") is not None:
        line["roofline"]["valu_issue_frac"] = _num(r["valu_issue"]["issue_fraction"])
    m = out.get("roofline_mfma") or {}
    line["roofline_mfma"] = dict(_pick(m, ("achieved", "peak", "unit", "frac", "frac_useful", "avg_launch_ms")), kernel="k_lba_schur")
    if isinstance(m.get("alone"), dict):
        line["roofline_mfma"]["alone"] = _pick(m["alone"], ("achieved", "frac", "frac_useful", "avg_launch_ms"))
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample_short") or cb.get("sample", ""))[:200]
        if isinstance(cb.get("single_stream"), dict):
            line["cpu_baseline"]["single_stream_value"] = _num(cb["single_stream"].get("value"))
        if line.get("value") and cb.get("value"):
            line["vs_cpu_baseline"] = _num(out["value"] / cb["value"], 4)
    ps = out.get("parity_sample")
    if isinstance(ps, dict) and "error" not in ps:
        lw = ps.get("local_ba_windows", [])
        line["parity_sample"] = dict(_pick(ps, ("keypoint_bytes_equal", "matches_equal", "inliers_equal", "max_se3_error")),
                                     frames_checked=len(ps.get("frames_checked", [])),
                                     lba_erase_flags_equal=all(w["erase_flags_equal"] for w in lw) if lw else None,
                                     lba_lm_trials_equal=all(w["lm_trials"][0] == w["lm_trials"][1] for w in lw) if lw else None,
                                     lba_max_se3_error=_num(max(w["max_se3_error"] for w in lw)) if lw else None)
    ss = out.get("single_stream")
    if isinstance(ss, dict) and "error" not in ss:
        line["single_stream"] = _pick(ss, ("frames", "ms_per_frame", "ms_per_frame_tracking_call", "ms_per_local_ba_mean",
                                           "ate_vs_oracle_m", "max_position_difference_vs_oracle_m", "vs_cpu_single_stream",
                                           "vs_cpu_threaded"))
        if isinstance(ss.get("drop_in"), dict):
            line["single_stream"]["drop_in_ms_per_frame"] = _num(ss["drop_in"].get("ms_per_frame"))
    # the other BASELINE configurations as sequences: whole-loop ms per frame of the C++ replays (examples/replay_modes)
    oc, bad_cpp = {}, []
    legs = dict(out.get("single_stream_rig") or {}, configs0_vision_only=out.get("single_stream_vision_only"))
    for name, leg in legs.items():
        if isinstance(leg, dict) and isinstance(leg.get("cpp"), dict):
            if "error" in leg["cpp"]:
                bad_cpp.append("%s.cpp" % name)
            else:
                oc[name] = _num(leg["cpp"].get("ms_per_frame"))
    if oc:
        line["other_configs_ms_per_frame"] = oc
    line["legs_failed"] = failed_legs(out) + bad_cpp
    line["detail"] = detail_path
    return line


def emit(out, stream=None):
    """Rank 0: the detail record to bench_detail.json (+ gpurun_out/ when that directory exists) and to stderr, then the
    compact line as the LAST line of stdout."""
    import ctypes
    stream = stream or sys.stdout
    detail = json.dumps(out)
    paths = [os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    written = None
    for p in paths:
        try:
            with open(p, "w") as f:
                f.write(detail + "\n")
            written = written or os.path.relpath(p, ROOT)
        except OSError:
            pass
    line = json.dumps(compact_line(out, written), separators=(",", ":"))
    if len(line) > LINE_BUDGET:
        raise RuntimeError("bench line is %d characters (budget %d): trim compact_line()" % (len(line), LINE_BUDGET))
    try:
        ctypes.CDLL(None).fflush(None)  # (RCCL's version banner sits in C stdio's buffer: out with it BEFORE the line)
    except OSError:
        pass
    sys.stderr.write("bench_detail: " + detail + "\n")
    sys.stderr.flush()
    stream.flush()
    stream.write(line + "\n")
    stream.flush()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2048, help="stereo frames per GPU per step")
    ap.add_argument("--workload", choices=("r3", "r2"), default="r3",
                    help="r3 (default): SURVEY 8d as written -- 64 distinct cases with non-planar depth and low-contrast "
                         "patches, isInFrustum + query construction for 4-5 k local-map candidates per frame inside the step, "
                         "local-BA windows with 40 fixed key frames (~21 k edges), every 4th one bLarge (25 local key "
                         "frames); r2: the shape of rounds 1-2 (8 views of one textured plane, host-precomputed local-map "
                         "queries, 6 fixed key frames / ~12 k edges), kept so that the numbers stay comparable")
    ap.add_argument("--base-cases", type=int, default=0, help="distinct rendered cases per GPU (0: 64 for r3, 8 for r2)")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent frame pipelines per GPU, each on its own HIP stream (the batch "
                         "is split between them so latency-bound stages overlap extraction: 2 streams give "
                         "+3..5 %% frames/s, but then the per-kernel event times of stream 0 include the other "
                         "stream's kernels, so the default keeps the roofline attribution clean)")
    ap.add_argument("--lba-every", type=int, default=10, help="frames per LocalBundleAdjustment (0 = none)")
    ap.add_argument("--lba-threads", type=int, default=4,
                    help="host threads issuing LBA batches (each call = the windows of one step; with the bundle-adjustment "
                         "stream at the lowest priority up to four calls are in flight behind the front end)")
    ap.add_argument("--lba-mixed", action="store_true",
                    help="one lock-step call per step with ordinary and bLarge windows mixed (default: one call per class)")
    ap.add_argument("--lba-batch", type=int, default=0,
                    help="windows per lock-step LBA call (0 = all windows of a step in one call)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-stream-frames", type=int, default=400,
                    help="frames of the sequential replay leg (0 = skip); rank 0 at N = 1 only.  400 frames = 40 key frames: "
                         "the local-BA windows reach 10 free key frames + their fixed observers")
    ap.add_argument("--sequence-frames", type=int, default=100,
                    help="frames of the rig / vision-only sequence replays (0 = skip); rank 0 at N = 1 only")
    ap.add_argument("--no-pcie-leg", action="store_true")
    ap.add_argument("--no-rig-legs", action="store_true",
                    help="skip single_stream_rig / rig_batch / single_stream_vision_only (rank 0 at N = 1 only)")
    ap.add_argument("--parity-sample", type=int, default=8,
                    help="frames of the timed batch recomputed on the CPU oracle after the timed region (0 = none)")
    ap.add_argument("--no-multi-gpu-legs", action="store_true",
                    help="skip the landmark-sharded local / full BA and the configs[3] rig-replica legs (they run on every "
                         "rank after the timed region, with a one-rank communicator at --gpus 1)")
    a = ap.parse_args()

    from vieo_slam_amd import sharding
    rank, world, local = sharding.env_rank()
    import torch  # loaded first so the process uses one HIP runtime
    import torch.distributed as dist
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from vieo_slam_amd.orb_extractor import STAGES as ORB_STAGES
    from vieo_slam_amd.pipeline import FramePipeline, make_cases

    B = a.batch
    S = max(1, min(a.streams, B))
    n_base = a.base_cases if a.base_cases > 0 else (64 if a.workload == "r3" else 8)
    cases = make_cases(min(n_base, B), seed0=sharding.rank_seed(rank), workload=a.workload)
    sizes = [B // S + (1 if i < B % S else 0) for i in range(S)]
    pipes = [FramePipeline(cases[i % len(cases):] + cases[:i % len(cases)], sizes[i], seed=rank * 16 + i, workload=a.workload)
             for i in range(S)]
    P = pipes[0]
    n_img = 2 * P.B

    # ---- local BA windows (one per `lba_every` frames), issued by host threads
    from concurrent.futures import ThreadPoolExecutor
    from vieo_slam_amd import synth_ba
    from vieo_slam_amd.optimizer import Optimizer
    n_lba = (B + a.lba_every - 1) // a.lba_every if a.lba_every > 0 else 0
    if a.workload == "r3":  # 10 local + 40 fixed key frames, ~1600 points / ~21 k edges; every 4th window bLarge
        lba_problems = []
        for i in range(min(8, n_lba)):
            large = i % 4 == 3
            w = synth_ba.make_lba_vio_problem(500 + 7 * rank + i, n_local=25 if large else 10, n_fixed=40, n_points=2000)[:6]
            w[0][0]["large"] = int(large)
            if large:  # ORB3_STRATEGY_OPT_WIDER: Nlocal *= 2.5, optit = {2, 2} (Optimizer.cc:45-52)
                w[0][0]["base"]["its0"], w[0][0]["base"]["its1"] = 2, 2
            lba_problems.append(w)
    else:
        lba_problems = [synth_ba.make_lba_vio_problem(500 + 7 * rank + i, n_local=10, n_fixed=6, n_points=2000)[:6]
                        for i in range(min(8, n_lba))]
    pool = ThreadPoolExecutor(max_workers=max(1, a.lba_threads))
    lba_ms = []

    lba_chunk = a.lba_batch if a.lba_batch > 0 else max(1, n_lba)
    chunks = [list(range(i, min(i + lba_chunk, n_lba))) for i in range(0, n_lba, lba_chunk)]
    # Windows of one solver class per lock-step call (a call advances at the pace of its slowest class: the bLarge
    # windows' four heavy rounds -- one-workgroup solves of 0.4 ms, off-diagonal Schur tiles -- held the ordinary windows'
    # seventeen light ones back and vice versa).  --lba-mixed keeps the one mixed call per step.
    if a.workload == "r3" and not a.lba_mixed and a.lba_batch <= 0:
        is_large = [int(lba_problems[i % len(lba_problems)][0][0]["large"]) != 0 for i in range(n_lba)]
        small = [i for i in range(n_lba) if not is_large[i]]
        large = [i for i in range(n_lba) if is_large[i]]
        ns = max(1, int(os.environ.get("VIEO_BENCH_LBA_SMALL_CALLS", "1")))
        per = (len(small) + ns - 1) // ns if small else 0
        chunks = [small[k:k + per] for k in range(0, len(small), per)] if small else []
        chunks = [c for c in ([large] + chunks if os.environ.get("VIEO_BENCH_LBA_LARGE_FIRST") else chunks + [large]) if c]

    from vieo_slam_amd._lib import check as _check, lib as _lib

    def run_lba(idx):
        _check(_lib().vieo_set_device(local), "vieo_set_device")  # worker threads start on device 0
        t = time.perf_counter()
        r = Optimizer.LocalBundleAdjustmentNavStatePRVBatch([lba_problems[i % len(lba_problems)] for i in idx])
        lba_ms.append((time.perf_counter() - t) * 1e3 / len(idx))
        return [x[3] for x in r]

    def sync_all():
        for q in pipes:
            q.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        for q in pipes:
            q.step()
    if n_lba:
        list(pool.map(run_lba, chunks * (2 * a.lba_threads)))  # every worker thread creates its stream
    lba_ms.clear()
    P.enable_timing(True)
    P.ext.enable_timing(True)
    Optimizer.enable_kernel_timing(2 if os.environ.get("VIEO_BENCH_LBA_COUNT_ONLY") else True)
    sync_all()
    t0 = time.perf_counter()
    futs = []
    skip_fe = bool(os.environ.get("VIEO_BENCH_SKIP_FE"))  # experiment: the bundle-adjustment engine's capacity alone
    for _ in range(a.steps):
        for q in pipes:
            if not skip_fe:
                q.step()
        futs += [pool.submit(run_lba, c) for c in chunks]
    lba_res = [r for f in futs for r in f.result()]
    sync_all()
    dt = time.perf_counter() - t0
    dt = sharding.max_over_ranks(dist, dt, device="cuda")

    stage = P.stage_ms_all()
    orb = P.ext.stage_ms_all()
    res = P.results()
    # every bundle-adjustment kernel class of the timed region (the engine's own events on its streams), then -- the
    # engine's stream has the lowest priority, so those events mostly measure waiting -- the duration of the same
    # launches ALONE: one more batch of the step's windows.  Both before the multi-GPU legs, which run the same kernels.
    lba_k, schur_flops = Optimizer.kernel_times()
    lba_alone, schur_alone = {}, None
    if n_lba:
        for c_ in chunks:  # the step's calls one after the other, alone on the GPU
            run_lba(c_)
        k1, f1 = Optimizer.kernel_times()
        for k, v in k1.items():
            dn, dm = v["launches"] - lba_k[k]["launches"], v["ms"] - lba_k[k]["ms"]
            lba_alone[k] = dm / dn if dn > 0 else 0.0
        ds = k1["lba.schur"]["ms"] - lba_k["lba.schur"]["ms"]
        if ds > 0:
            tf = (f1 - schur_flops) / (ds * 1e-3) / 1e12
            schur_alone = {"achieved": tf, "frac": tf / FP64_MFMA_PEAK_TFLOPS, "avg_launch_ms": lba_alone["lba.schur"],
                           "windows_per_launch": "/".join(str(len(c_)) for c_ in chunks),
                           "note": "one lock-step batch of the step's windows after the timed region"}
    Optimizer.enable_kernel_timing(False)
    mg = None
    if not a.no_multi_gpu_legs:
        try:
            mg = multi_gpu_legs(rank, world, dist, torch)
        except Exception as e:  # a leg must not take the headline down with it
            mg = {"error": repr(e)}
    if rank == 0:
        avg = {k: float(np.mean([s[k] for s in stage])) for k in P.STAGES}
        oavg = {k: float(np.mean([s[k] for s in orb])) for k in ORB_STAGES}
        ab = algorithmic_bytes_per_image()
        # every kernel (class) of the path, ms per step: the extractor's kernels and the front-end stages from the
        # events of stream 0, the bundle-adjustment kernel classes as collected above
        # (stream 0 carries P.B of the step's B frames: its launch times are scaled to the whole step for the ranking)
        share = B / float(P.B)
        kern = {("orb." + k): v * share for k, v in oavg.items() if k != "total"}
        kern.update({("frontend." + k): v * share for k, v in avg.items() if k not in ("extract", "total")})
        kern.update({k: lba_alone.get(k, 0.0) * v["launches"] / a.steps for k, v in lba_k.items()})
        dom = max(kern, key=kern.get)
        launches = lba_k[dom]["launches"] / a.steps if dom in lba_k else share
        # algorithmic bytes per launch of the kernels that are HBM-bound by design
        nfree = float(np.mean([int((p[1]["fixed"] == 0).sum()) for p in lba_problems])) if lba_problems else 0
        nmp_w, nobs_w = np.mean([len(p[2]) for p in lba_problems]) if lba_problems else 0, \
            np.mean([len(p[4]) for p in lba_problems]) if lba_problems else 0
        lba_ab = {"lba.build": n_lba * (6 * nfree * 3 * nmp_w * 8 + 40 * nobs_w + 96 * nmp_w)}  # dense BB + edges + H_ll, b_l
        if dom.startswith("orb."):
            dk = dom.split(".")[1]
            alg_bytes = ab[dk] * n_img
        else:
            dk, alg_bytes = None, lba_ab.get(dom)
        launch_ms = kern[dom] / launches if launches else 0.0
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9 if (alg_bytes and launch_ms > 0) else None
        roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if achieved else None,
                "traffic": pmc_traffic(dk, n_img) if dk else None, "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": launch_ms}
        if dk:
            roof["valu_issue"] = valu_issue(dk, n_img, launch_ms)
        if dom == "lba.schur" and schur_alone:  # the one MFMA kernel of the path: priced in FLOPs, not bytes
            roof = {"bound": "mfma", "kernel": dom, "achieved": schur_alone["achieved"], "peak": FP64_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": schur_alone["frac"], "traffic": None,
                    "flops": "dense 2 np (np + 1) 3 n_mp per window and LM trial", "avg_launch_ms": launch_ms,
                    "note": "duration of the launch alone (the bundle-adjustment stream has the lowest priority; in situ "
                            "its events mostly measure waiting, see roofline_mfma)"}
        roof["chosen_over"] = ("all kernels of the path: extractor kernels, front-end stages (1-3 kernels each) and "
                               "bundle-adjustment kernel classes, by ms per step")
        # useful / dense FLOPs of one LM trial over the windows of a step
        useful_ratio = None
        if lba_problems:
            us = sum(schur_useful_flops(lba_problems[i % len(lba_problems)]) for i in range(n_lba))
            de = 0.0
            for i in range(n_lba):
                w = lba_problems[i % len(lba_problems)]
                npv = 6.0 * int((w[1]["fixed"] == 0).sum())
                de += 2 * npv * (npv + 1) * 3 * len(w[2])
            useful_ratio = us / de if de > 0 else None
        sch = lba_k.get("lba.schur", {"ms": 0.0, "launches": 0})
        sch_tf = schur_flops / (sch["ms"] * 1e-3) / 1e12 if sch["ms"] > 0 else None
        r2 = res["r2"]
        perr = [np.linalg.norm(r2[b]["base"]["nav"]["p"] - P.truth[b]["p"]) for b in range(P.B)]
        from vieo_slam_amd import trajectory  # the metric's "ATE vs ref": Horn-aligned RMSE of the step's frame positions
        _, _, ate_err = trajectory.align(np.array([r2[b]["base"]["nav"]["p"] for b in range(P.B)]).T,
                                         np.array([P.truth[b]["p"] for b in range(P.B)]).T)
        out = {
            "metric": "frontend+localBA frames/sec on EuRoC MH05 stereo-VIO; ATE vs ref",
            "value": sharding.aggregate_throughput(B, a.steps, world, dt),
            "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (extract/match) + f64 (pose optimisation)", "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2] 'EuRoC MH05 stereo-VIO + LocalBundleAdjustment' (= configs[1] plus the "
                            "local BA; --lba-every 0 is configs[1] alone): "
                            "synthetic rendered stereo-inertial frames 752x480, 1.2x8 levels, FAST 20/7; per "
                            "frame ORBextractor x2 + ComputeStereoMatches + SearchByProjection(last frame) + "
                            "PoseOptimization(VIO) + SearchByProjection(local map) + PoseOptimization(VIO, marg); " +
                            ("workload r3 (SURVEY 8d): %d distinct cases (8 textures with low-contrast patches x sheet "
                             "layouts: non-planar depth) x noise replicas; Frame::isInFrustum + query construction for "
                             "%d local-map candidates per frame inside the step; plus one LocalBundleAdjustmentNavStatePRV "
                             "(10 local key frames with PR+V+Bias vertices and IMU edges, 40 fixed, ~%d points, ~%d "
                             "observations; every 4th window bLarge with 25 local key frames) per "
                             % (len(cases), int(np.mean(P.ncand_host)), int(nmp_w), int(nobs_w)) if a.workload == "r3" else
                             "workload r2: 8 views of one textured plane, local-map queries precomputed on the host; plus one "
                             "LocalBundleAdjustmentNavStatePRV (10 local key frames with PR+V+Bias vertices and IMU "
                             "pre-integration edges, 6 fixed, ~1500 points, ~12k observations) per ") +
                            "%d frames, the windows of a step advanced in lock step (calls of %s windows%s; %d host threads)"
                            % (a.lba_every, " + ".join(str(len(c)) for c in chunks),
                               "" if (a.lba_mixed or a.lba_batch > 0 or a.workload != "r3") else
                               ": one call per solver class, i.e. ordinary / bLarge windows apart", a.lba_threads),
                "workload_id": a.workload, "lba_every": a.lba_every,
                "local_ba_windows_per_step": n_lba,
                "local_ba_ms_per_window_mean": float(np.mean(lba_ms)) if lba_ms else None,
                "local_ba_lm_iterations_mean": float(np.mean([r["lm_iterations"] for r in lba_res])) if lba_res else None,
                "stereo_frames_per_gpu_per_step": B,
                "hip_streams_per_gpu": S,
                "parallelism": "frames sharded, one batch per GPU split over %d HIP stream(s), no collective" % S,
                "mean_keypoints_per_image": float(res["counts"][:, 0].mean()),
                "mean_pose_inliers": float(np.mean(r2["base"]["n_inliers"])),
                "median_position_error_vs_truth_m": float(np.median(perr)),
                "ate_rmse_vs_truth_m": float(np.sqrt(np.mean(ate_err ** 2))),
            },
            "stage_ms_per_step_stream0": avg,
            "extractor_kernel_ms_per_step_stream0": oavg,
            "kernel_ms_per_step_all": {k: round(v, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])},
            "kernel_ms_note": "orb.* / frontend.*: HIP events of the timed steps on the pipeline stream; lba.*: launches per "
                              "step x the duration of the same launch alone (the bundle-adjustment stream runs at the "
                              "lowest priority; its in-situ event times, mostly waiting, are in lba_elapsed_ms_per_step_in_situ)",
            "lba_elapsed_ms_per_step_in_situ": {k: round(v["ms"] / a.steps, 3) for k, v in lba_k.items()},
            "roofline": roof,
            # the one dense contraction of the path: the Schur complement of the local BA on the FP64 matrix cores
            "roofline_mfma": {"bound": "mfma", "kernel": "lba.schur (k_lba_schur<diagonal tiles> + <off-diagonal tiles>: two launches per LM trial round, v_mfma_f64_16x16x4_f64)",
                              "achieved": sch_tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": sch_tf / FP64_MFMA_PEAK_TFLOPS if sch_tf else None,
                              "flops_per_launch": schur_flops / sch["launches"] if sch["launches"] else None,
                              "flops": "dense 2 np (np + 1) 3 n_mp per window and LM trial (np = 6 x free key frames)",
                              "useful_over_dense": useful_ratio,
                              "frac_useful": sch_tf * useful_ratio / FP64_MFMA_PEAK_TFLOPS if (sch_tf and useful_ratio) else None,
                              "useful_flops_per_launch": (schur_flops / sch["launches"] * useful_ratio) if (sch["launches"] and useful_ratio) else None,
                              "useful_flops": "SURVEY 8d's sparsity-aware count: per landmark with k free observers 60 + 108 k + "
                                              "216 k (k + 1) / 2 + 36 k, summed over the step's windows (one LM trial each)",
                              "avg_launch_ms": sch["ms"] / sch["launches"] if sch["launches"] else None,
                              "launches_per_step": sch["launches"] / a.steps,
                              "mfma_busy": mfma_busy()},
        }
        if schur_alone:
            if useful_ratio:
                schur_alone["frac_useful"] = schur_alone["frac"] * useful_ratio
            out["roofline_mfma"]["alone"] = schur_alone
        if dk:
            # the same kernel with the GPU to itself (no bundle-adjustment stream beside it): a few more launches of the
            # extractor after the timed region, same events.  `achieved` / `frac` above stay the in-situ figures.
            for _ in range(4):
                P.ext.extract_batch_device(P.d_img.ptr, P.n_img, 752, 480, 752, 752 * 480, P.d_kp.ptr, P.d_desc.ptr,
                                           P.cap, P.d_cnt.ptr)
            P.sync()
            alone = P.ext.stage_ms_all()[-3:]
            ms_alone = float(np.mean([m[dk] for m in alone]))
            out["roofline"]["alone"] = {"avg_launch_ms": ms_alone, "achieved": ab[dk] * P.n_img / (ms_alone * 1e-3) / 1e9,
                                        "frac": ab[dk] * P.n_img / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "extractor_kernel_ms": {k: float(np.mean([m[k] for m in alone])) for k in ORB_STAGES},
                                        "note": "same launch without the bundle-adjustment stream beside it (its short "
                                                "high-priority kernels take CUs from the front end in the timed region)"}
        if mg is not None:
            out["multi_gpu"] = mg
        if world == 1 and not a.no_pcie_leg:
            out["pcie_inclusive"] = pcie_leg(P, max(3, min(a.steps, 10)), 2)
        oracle_runs = None
        n_seq = a.sequence_frames
        if world == 1 and not a.no_cpu_baseline:
            # the oracle's replays of the sequence legs (checker + single-stream CPU baseline): worker processes, from now on
            tasks = {}
            if a.single_stream_frames > 1:
                tasks["vio"] = ("vio", 1, a.single_stream_frames, LBA_LAG, None, 0, 0)
            if not a.no_rig_legs and n_seq > 1:
                tasks.update({"radtan2": ("rig", 3, n_seq, LBA_LAG, "radtan", 2, 1200), "kb8_4": ("rig", 5, n_seq, LBA_LAG, "kb8", 4, 1500),
                              "kb8_2": ("rig", 4, n_seq, LBA_LAG, "kb8", 2, 1500), "vision": ("vision", 2, n_seq, LBA_LAG, None, 0, 0)})
            if tasks:
                oracle_runs = OracleReplays(tasks)
        if world == 1 and not a.no_pcie_leg and n_lba:
            # H2D + front end + LocalBA in ONE timed region (the headline has the images resident, pcie_inclusive no LocalBA)
            out["inclusive"] = pcie_leg(P, max(3, min(a.steps, 10)), 2, lba=(pool, run_lba, chunks))
        if world == 1 and not a.no_rig_legs:
            try:  # the other BASELINE configurations on their fast path (a leg must not take the headline down)
                out["rig_batch"] = rig_frontend_batch()
            except Exception as e:
                out["rig_batch"] = {"error": repr(e)}
        if world == 1 and not a.no_cpu_baseline and a.workload == "r3" and a.parity_sample > 0:
            # the headline tied to checked outputs: sampled frames of the timed batch and two of the step's local-BA
            # windows recomputed on the CPU oracle (after the timed region; the oracle is the checker, never the product)
            try:
                from tests import oracle_lib
                from tests.pipeline_check import R3FrameChecker
                orc = oracle_lib.load()
                C = R3FrameChecker(P, res, orc)
                idx = np.unique(np.linspace(0, P.B - 1, a.parity_sample).astype(int))
                chk = [C.check(int(b)) for b in idx]
                ps = {"frames_checked": [int(b) for b in idx],
                      "keypoint_bytes_equal": all(c["keys_equal"] and c["uright_equal"] for c in chk),
                      "matches_equal": all(c["matches_equal"] for c in chk), "inliers_equal": all(c["inliers_equal"] for c in chk),
                      "max_se3_error": max(c["max_se3_error"] for c in chk)}
                lerr = []
                for wi in range(min(2, len(lba_problems))):
                    w = lba_problems[-1] if wi else lba_problems[0]  # one ordinary, one bLarge window (every 4th)
                    on, op, oe, ores = orc.local_ba_vio(*w)
                    hn, hp, he, hres = Optimizer.LocalBundleAdjustmentNavStatePRV(*w)
                    lerr.append({"key_frames": len(w[1]), "observations": len(w[4]), "large": int(w[0][0]["large"]),
                                 "max_se3_error": float(max(max(synth_ba.pose_error(on[k], hn[k])) for k in range(len(on)))),
                                 "erase_flags_equal": bool(np.array_equal(oe, he)),
                                 "lm_trials": [int(ores["lm_trials"]), int(hres["lm_trials"])]})
                ps["local_ba_windows"] = lerr
                ps["note"] = ("frames of the timed device-resident batch re-evaluated stage by stage on the CPU oracle from the "
                              "same inputs (tests/pipeline_check.py); local-BA windows of the timed step against the oracle's "
                              "LocalBundleAdjustmentNavStatePRV; bar: keypoint bytes / matches equal, SE(3) <= 1e-4")
                out["parity_sample"] = ps
            except Exception as e:
                out["parity_sample"] = {"error": repr(e)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, lba_problems, a.lba_every)
        if world == 1 and a.single_stream_frames > 1:
            from vieo_slam_amd import replay
            seq = replay.Sequence(1, a.single_stream_frames)
            orc = oracle_runs.get("vio") if oracle_runs else None
            if orc:
                for k, im in enumerate(orc["images"]):
                    seq._img[k] = im
            out["single_stream"], th, td = single_stream_leg(seq, a.single_stream_frames)
            if orc:  # the oracle's run of the same replay: baseline and checker
                to = orc["traj"]
                cb = {"value": (a.single_stream_frames - 1) / orc["seconds"], "unit": "frames/s", "cores": 1, "kind": "port",
                      "sample": "the %d-frame sequential replay of the single_stream leg (tracking + one local BA per 10 frames) "
                                "on the CPU oracle, one thread (a worker process beside the other legs)" % a.single_stream_frames}
                out["cpu_baseline"]["single_stream"] = cb
                ss = out["single_stream"]
                ss["vs_cpu_single_stream"] = ss["frames_per_s"] / cb["value"]
                # against the oracle threaded like the reference (one thread per camera + LocalMapping: cpu_baseline.value)
                ss["vs_cpu_threaded"] = ss["frames_per_s"] / out["cpu_baseline"]["value"]
                ss["ate_vs_oracle_m"] = replay.ate_between(th, to)
                ss["max_position_difference_vs_oracle_m"] = float(np.linalg.norm(th["p"] - to["p"], axis=1).max())
                ss["drop_in"]["ate_vs_oracle_m"] = replay.ate_between(td, to)
                ss["drop_in"]["max_position_difference_vs_oracle_m"] = float(np.linalg.norm(td["p"] - to["p"], axis=1).max())
                ss["drop_in"]["vs_cpu_single_stream"] = ss["drop_in"]["frames_per_s"] / cb["value"]
                n2 = a.single_stream_frames
                ss["ate_vs_oracle_last_200_m"] = replay.ate_between(th[max(0, n2 - 200):], to[max(0, n2 - 200):])
        if world == 1 and not a.no_rig_legs and oracle_runs and n_seq > 1:
            try:  # the other BASELINE configurations as SEQUENCES with their local BA in the loop
                rigs = {"default_mh05_distorted_stereo": ("radtan2", "radtan", 2, 1200, 3), "configs3_4cam_kb8": ("kb8_4", "kb8", 4, 1500, 5),
                        "configs4_tumvi_2cam_kb8": ("kb8_2", "kb8", 2, 1500, 4)}
                ssr = {}
                for name, (key, rig, nc, nf, seed) in rigs.items():
                    leg = sequence_leg("rig", oracle_runs.get(key), n_seq, LBA_LAG, rig, nc, nf, seed)
                    leg["config"] = "%d-camera %s rig, %d features per camera, %s" % (nc, rig, nf, "752x480" if rig == "radtan" else "512x512")
                    ssr[name] = leg
                ssr["reference_readme_anchor"] = dict(
                    README_FRONTEND_DIST_MS, ratio_to_default_mh05_ms_per_frame_tracking_call=
                    README_FRONTEND_DIST_MS["value"] / ssr["default_mh05_distorted_stereo"]["ms_per_frame_tracking_call"])
                out["single_stream_rig"] = ssr
                leg = sequence_leg("vision", oracle_runs.get("vision"), n_seq, LBA_LAG, seed=2)
                leg["config"] = ("BASELINE configs[0]: rectified stereo WITHOUT IMU, 1000 features, 752x480; TrackWithMotionModel + "
                                 "TrackLocalMap as one vieo_track_frame call (vision-only PoseOptimization x 2), LocalBundleAdjustment "
                                 "per key frame")
                out["single_stream_vision_only"] = leg
            except Exception as e:
                import traceback
                out["single_stream_rig"] = {"error": repr(e), "trace": traceback.format_exc()[-1500:]}
        if oracle_runs:
            oracle_runs.close()
        emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
