"""Host-side mirror of VIEO_SLAM::Optimizer (reference include/Optimizer.h:55-101) for the
functions on the hot path, over flattened problems (ba_types.py) instead of Frame/MapPoint
pointer graphs."""
import numpy as np

from ._lib import check, lib
from .ba_types import (LBA_RESULT_DTYPE, NAVSTATE_DTYPE, POSE_FRAME_DTYPE, POSE_OBS_DTYPE, POSE_RESULT_DTYPE, VIO_FRAME_DTYPE,
                       VIO_RESULT_DTYPE)


class Optimizer:
    @staticmethod
    def PoseOptimization(frame, obs):
        """int Optimizer::PoseOptimization(Frame*, Frame* = NULL) (Optimizer.cc:1611-1874).
        frame: POSE_FRAME_DTYPE[1]; obs: POSE_OBS_DTYPE[n].
        returns (result POSE_RESULT_DTYPE record, outlier uint8[n]); result['n_inliers'] is the
        reference's return value."""
        fr = np.ascontiguousarray(frame, POSE_FRAME_DTYPE).reshape(1)
        ob = np.ascontiguousarray(obs, POSE_OBS_DTYPE)
        outl = np.zeros(max(len(ob), 1), np.uint8)
        res = np.zeros(1, POSE_RESULT_DTYPE)
        check(lib().vieo_pose_optimization(fr.ctypes.data, ob.ctypes.data, outl.ctypes.data,
                                           res.ctypes.data), "vieo_pose_optimization")
        return res[0], outl[:len(ob)]

    @staticmethod
    def PoseOptimizationVIO(vio_frame, obs):
        """template<class KeyFrame> int Optimizer::PoseOptimization(Frame*, KeyFrame* pLastKF, gw,
        bComputeMarg, bNoMPs) (include/Optimizer.h:208-816).  vio_frame: VIO_FRAME_DTYPE[1].
        returns (VIO_RESULT_DTYPE record, outlier uint8[n])."""
        fr = np.ascontiguousarray(vio_frame, VIO_FRAME_DTYPE).reshape(1)
        ob = np.ascontiguousarray(obs, POSE_OBS_DTYPE)
        outl = np.zeros(max(len(ob), 1), np.uint8)
        res = np.zeros(1, VIO_RESULT_DTYPE)
        check(lib().vieo_pose_optimization_vio(fr.ctypes.data, ob.ctypes.data, outl.ctypes.data,
                                               res.ctypes.data), "vieo_pose_optimization_vio")
        return res[0], outl[:len(ob)]

    @staticmethod
    def LocalBundleAdjustment(params, kfs, points, obs, stop=None, enc=None):
        """void Optimizer::LocalBundleAdjustment(KeyFrame*, bool* pbStopFlag, Map*, int Nlocal)
        (src/Optimizer.cc:1876-2307) on a flattened window (ba_types.LBA_*).  enc: LBA_ENC_DTYPE[1] with the
        EdgeEncNavStatePR pairs (:2008-2042) or None.
        returns (navs[n_kf], points float32[n_mp,3], erase uint8[n_obs], result record)."""
        params, kfs = np.ascontiguousarray(params), np.ascontiguousarray(kfs)
        points, obs = np.ascontiguousarray(points, np.float32), np.ascontiguousarray(obs)
        navs = np.zeros(len(kfs), NAVSTATE_DTYPE)
        pts = np.zeros_like(points)
        erase = np.zeros(max(len(obs), 1), np.uint8)
        res = np.zeros(1, LBA_RESULT_DTYPE)
        st = None if stop is None else np.ascontiguousarray(stop, np.int32)
        if enc is not None:
            enc = np.ascontiguousarray(enc)
            check(lib().vieo_local_bundle_adjustment_enc(
                params.ctypes.data, kfs.ctypes.data, len(kfs), points.ctypes.data, len(points), obs.ctypes.data,
                len(obs), enc.ctypes.data, None if st is None else st.ctypes.data, navs.ctypes.data, pts.ctypes.data,
                erase.ctypes.data, res.ctypes.data), "vieo_local_bundle_adjustment_enc")
            return navs, pts, erase[:len(obs)], res[0]
        check(lib().vieo_local_bundle_adjustment(params.ctypes.data, kfs.ctypes.data, len(kfs),
                                                 points.ctypes.data, len(points), obs.ctypes.data,
                                                 len(obs), None if st is None else st.ctypes.data,
                                                 navs.ctypes.data, pts.ctypes.data, erase.ctypes.data,
                                                 res.ctypes.data), "vieo_local_bundle_adjustment")
        return navs, pts, erase[:len(obs)], res[0]

    @staticmethod
    def LocalBundleAdjustmentBatch(windows, stop=None, encs=None):
        """Several independent LocalBundleAdjustment windows in lock step (one launch sequence for
        all of them).  windows: list of (params, kfs, points, obs) as for LocalBundleAdjustment; encs: None or
        one LBA_ENC_DTYPE[1] / None per window.
        returns a list of (navs, points, erase, result) in the same order."""
        W = len(windows)
        keep, outs = [], []
        ptrs = [np.zeros(W, np.uint64) for _ in range(7)]  # params kfs points obs navs pts erase
        cnt = [np.zeros(W, np.int32) for _ in range(3)]    # n_kf n_mp n_obs
        res = np.zeros(W, LBA_RESULT_DTYPE)
        for w, (params, kfs, points, obs) in enumerate(windows):
            params, kfs = np.ascontiguousarray(params), np.ascontiguousarray(kfs)
            points, obs = np.ascontiguousarray(points, np.float32), np.ascontiguousarray(obs)
            navs = np.zeros(len(kfs), NAVSTATE_DTYPE)
            pts = np.zeros_like(points)
            erase = np.zeros(max(len(obs), 1), np.uint8)
            keep.append((params, kfs, points, obs))
            outs.append((navs, pts, erase, len(obs)))
            for a, arr in zip(ptrs, (params, kfs, points, obs, navs, pts, erase)):
                a[w] = arr.ctypes.data
            cnt[0][w], cnt[1][w], cnt[2][w] = len(kfs), len(points), len(obs)
        st = None if stop is None else np.ascontiguousarray(stop, np.int32)
        if encs is not None:
            encs = [None if e is None else np.ascontiguousarray(e) for e in encs]
            pe = np.array([0 if e is None else e.ctypes.data for e in encs], np.uint64)
            check(lib().vieo_local_bundle_adjustment_batch_enc(
                W, ptrs[0].ctypes.data, ptrs[1].ctypes.data, cnt[0].ctypes.data, ptrs[2].ctypes.data,
                cnt[1].ctypes.data, ptrs[3].ctypes.data, cnt[2].ctypes.data, pe.ctypes.data,
                None if st is None else st.ctypes.data, ptrs[4].ctypes.data, ptrs[5].ctypes.data,
                ptrs[6].ctypes.data, res.ctypes.data), "vieo_local_bundle_adjustment_batch_enc")
            return [(n, p, e[:k], res[w]) for w, (n, p, e, k) in enumerate(outs)]
        check(lib().vieo_local_bundle_adjustment_batch(
            W, ptrs[0].ctypes.data, ptrs[1].ctypes.data, cnt[0].ctypes.data, ptrs[2].ctypes.data,
            cnt[1].ctypes.data, ptrs[3].ctypes.data, cnt[2].ctypes.data,
            None if st is None else st.ctypes.data, ptrs[4].ctypes.data, ptrs[5].ctypes.data,
            ptrs[6].ctypes.data, res.ctypes.data), "vieo_local_bundle_adjustment_batch")
        return [(n, p, e[:k], res[w]) for w, (n, p, e, k) in enumerate(outs)]

    @staticmethod
    def LocalBundleAdjustmentNavStatePRV(params, kfs, points, close, obs, imu, stop=None):
        """void Optimizer::LocalBundleAdjustmentNavStatePRV(KeyFrame*, int Nlocal, bool* pbStopFlag, Map*,
        cv::Mat gw, bool bLarge, bool bRecInit, float th_dist_far) (src/Optimizer.cc:21-769) on a flattened
        window (ba_types.LBA_VIO_PARAMS / LBA_KEYFRAME / LBA_OBS / LBA_IMU_EDGE).
        returns (navs[n_kf], points float32[n_mp,3], erase uint8[n_obs], result record)."""
        return Optimizer.LocalBundleAdjustmentNavStatePRVBatch([(params, kfs, points, close, obs, imu)], stop)[0]

    @staticmethod
    def LocalBundleAdjustmentNavStatePRVBatch(windows, stop=None):
        """Several independent visual-inertial windows in lock step.
        windows: list of (params, kfs, points, close, obs, imu)."""
        W = len(windows)
        keep, outs = [], []
        ptrs = [np.zeros(W, np.uint64) for _ in range(9)]  # params kfs points close obs imu navs pts erase
        cnt = [np.zeros(W, np.int32) for _ in range(4)]    # n_kf n_mp n_obs n_imu
        res = np.zeros(W, LBA_RESULT_DTYPE)
        for w, (params, kfs, points, close, obs, imu) in enumerate(windows):
            params, kfs, imu = np.ascontiguousarray(params), np.ascontiguousarray(kfs), np.ascontiguousarray(imu)
            points, obs = np.ascontiguousarray(points, np.float32), np.ascontiguousarray(obs)
            close = np.ascontiguousarray(close, np.uint8)
            navs = np.zeros(len(kfs), NAVSTATE_DTYPE)
            pts = np.zeros_like(points)
            erase = np.zeros(max(len(obs), 1), np.uint8)
            keep.append((params, kfs, points, close, obs, imu))
            outs.append((navs, pts, erase, len(obs)))
            for a, arr in zip(ptrs, (params, kfs, points, close, obs, imu, navs, pts, erase)):
                a[w] = arr.ctypes.data
            cnt[0][w], cnt[1][w], cnt[2][w], cnt[3][w] = len(kfs), len(points), len(obs), len(imu)
        st = None if stop is None else np.ascontiguousarray(stop, np.int32)
        check(lib().vieo_local_bundle_adjustment_vio_batch(
            W, ptrs[0].ctypes.data, ptrs[1].ctypes.data, cnt[0].ctypes.data, ptrs[2].ctypes.data,
            ptrs[3].ctypes.data, cnt[1].ctypes.data, ptrs[4].ctypes.data, cnt[2].ctypes.data,
            ptrs[5].ctypes.data, cnt[3].ctypes.data, None if st is None else st.ctypes.data,
            ptrs[6].ctypes.data, ptrs[7].ctypes.data, ptrs[8].ctypes.data, res.ctypes.data),
            "vieo_local_bundle_adjustment_vio_batch")
        return [(n, p, e[:k], res[w]) for w, (n, p, e, k) in enumerate(outs)]

    @staticmethod
    def enable_kernel_timing(on=True):
        """HIP-event timing of every bundle-adjustment kernel launch, per kernel class (vieo_lba_enable_timing);
        also clears the totals.  on == 2: count the launches per class only (no events on the stream)."""
        lib().vieo_lba_enable_timing(2 if on == 2 else (1 if on else 0))

    @staticmethod
    def kernel_times():
        """({class: {"ms": total, "launches": n}}, dense FLOPs of the timed k_lba_schur launches) since
        enable_kernel_timing()."""
        n = lib().vieo_lba_kernel_classes()
        ms, cnt, fl = np.zeros(n), np.zeros(n, np.int64), np.zeros(1)
        lib().vieo_lba_kernel_times(ms.ctypes.data, cnt.ctypes.data, fl.ctypes.data)
        names = [lib().vieo_lba_kernel_class_name(i).decode() for i in range(n)]
        return {k: {"ms": float(ms[i]), "launches": int(cnt[i])} for i, k in enumerate(names)}, float(fl[0])

    @staticmethod
    def sharded_buffer_doubles(windows):
        """Doubles the reduction buffer of LocalBundleAdjustmentNavStatePRVSharded needs."""
        nf = np.array([int((np.asarray(w[1]["fixed"]) == 0).sum()) for w in windows], np.int32)
        return int(lib().vieo_lba_sharded_buffer_doubles(len(windows), nf.ctypes.data))

    @staticmethod
    def LocalBundleAdjustmentNavStatePRVSharded(windows, reduce_ptr, reduce_doubles, allreduce=None, comm=None, stop=None):
        """This rank's part of landmark-sharded visual-inertial windows.  windows: the
        rank's shards (sharding.shard_window); reduce_ptr: device pointer of a float64 buffer of
        reduce_doubles entries; allreduce(offset, n): in-place sum over the ranks of entries
        [offset, offset + n) of that buffer, complete on return (sharding.torch_allreduce) -- or comm: the in-library
        RCCL communicator of sharding.RcclComm (the library then issues ncclAllReduce on its own stream).  stop: int32[1],
        this rank's pbStopFlag (the ranks' flags are summed inside the run: one raised flag aborts all ranks together).
        Returns per window (navs, points, erase, result)."""
        import ctypes
        W = len(windows)
        keep, outs = [], []
        ptrs = [np.zeros(W, np.uint64) for _ in range(9)]
        cnt = [np.zeros(W, np.int32) for _ in range(4)]
        res = np.zeros(W, LBA_RESULT_DTYPE)
        for w, (params, kfs, points, close, obs, imu) in enumerate(windows):
            params, kfs, imu = np.ascontiguousarray(params), np.ascontiguousarray(kfs), np.ascontiguousarray(imu)
            points, obs = np.ascontiguousarray(points, np.float32), np.ascontiguousarray(obs)
            close = np.ascontiguousarray(close, np.uint8)
            navs = np.zeros(len(kfs), NAVSTATE_DTYPE)
            pts = np.zeros_like(points)
            erase = np.zeros(max(len(obs), 1), np.uint8)
            keep.append((params, kfs, points, close, obs, imu))
            outs.append((navs, pts, erase, len(obs)))
            for a, arr in zip(ptrs, (params, kfs, points, close, obs, imu, navs, pts, erase)):
                a[w] = arr.ctypes.data
            cnt[0][w], cnt[1][w], cnt[2][w], cnt[3][w] = len(kfs), len(points), len(obs), len(imu)
        CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
        base = int(reduce_ptr)

        def _cb(ctx, d_buf, n):
            try:
                return int(allreduce((int(d_buf) - base) // 8, int(n)))
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        cb = CB(_cb)
        fn_ptr, ctx = (ctypes.cast(cb, ctypes.c_void_p), None) if comm is None else (None, ctypes.c_void_p(int(comm)))
        check(lib().vieo_local_bundle_adjustment_vio_sharded_stop(
            W, ptrs[0].ctypes.data, ptrs[1].ctypes.data, cnt[0].ctypes.data, ptrs[2].ctypes.data,
            ptrs[3].ctypes.data, cnt[1].ctypes.data, ptrs[4].ctypes.data, cnt[2].ctypes.data,
            ptrs[5].ctypes.data, cnt[3].ctypes.data, ctypes.c_void_p(base), reduce_doubles,
            fn_ptr, ctx, None if stop is None else stop.ctypes.data, ptrs[6].ctypes.data, ptrs[7].ctypes.data,
            ptrs[8].ctypes.data, res.ctypes.data), "vieo_local_bundle_adjustment_vio_sharded")
        return [(n, p, e[:k], res[w]) for w, (n, p, e, k) in enumerate(outs)]

    @staticmethod
    def BundleAdjustment(params, kfs, points, obs, nIterations=5, bRobust=True, stop=None, enc=None):
        """void Optimizer::BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust, bEnc=false)
        (src/Optimizer.cc:1353-1609) -- GlobalBundleAdjustment passes the whole map -- on the flattened layout of
        LocalBundleAdjustment.  returns (navs[n_kf], points float32[n_mp,3], result record)."""
        params, kfs = np.ascontiguousarray(params), np.ascontiguousarray(kfs)
        points, obs = np.ascontiguousarray(points, np.float32), np.ascontiguousarray(obs)
        navs, pts, res = np.zeros(len(kfs), NAVSTATE_DTYPE), np.zeros_like(points), np.zeros(1, LBA_RESULT_DTYPE)
        st = None if stop is None else np.ascontiguousarray(stop, np.int32)
        if enc is not None:  # bEnc = true: EdgeEncNavStatePR pairs (Optimizer.cc:1401-1438), LBA_ENC_DTYPE[1]
            enc = np.ascontiguousarray(enc)
            check(lib().vieo_bundle_adjustment_enc(
                params.ctypes.data, int(nIterations), int(bool(bRobust)), kfs.ctypes.data, len(kfs), points.ctypes.data,
                len(points), obs.ctypes.data, len(obs), enc.ctypes.data, None if st is None else st.ctypes.data,
                navs.ctypes.data, pts.ctypes.data, res.ctypes.data), "vieo_bundle_adjustment_enc")
            return navs, pts, res[0]
        check(lib().vieo_bundle_adjustment(params.ctypes.data, int(nIterations), int(bool(bRobust)), kfs.ctypes.data,
                                           len(kfs), points.ctypes.data, len(points), obs.ctypes.data, len(obs),
                                           None if st is None else st.ctypes.data, navs.ctypes.data,
                                           pts.ctypes.data, res.ctypes.data), "vieo_bundle_adjustment")
        return navs, pts, res[0]

    @staticmethod
    def GlobalBundleAdjustmentNavStatePRV(params, kfs, points, obs, imu, nIterations=5, bRobust=True, stop=None,
                                          bScaleOpt=None):
        """int Optimizer::GlobalBundleAdjustmentNavStatePRV(pMap, gw, nIterations, pbStopFlag, nLoopKF, bRobust,
        bScaleOpt=false, pimu_initiator=nullptr) (src/Optimizer.cc:771-1345) on the flattened layout of
        LocalBundleAdjustmentNavStatePRV.  returns (navs[n_kf], points float32[n_mp,3], result record); with
        bScaleOpt given (System::FinalGBA passes true, src/System.cc:24-33) the recovered scale as a fourth value."""
        params, kfs = np.ascontiguousarray(params), np.ascontiguousarray(kfs)
        points, obs, imu = np.ascontiguousarray(points, np.float32), np.ascontiguousarray(obs), np.ascontiguousarray(imu)
        navs, pts, res = np.zeros(len(kfs), NAVSTATE_DTYPE), np.zeros_like(points), np.zeros(1, LBA_RESULT_DTYPE)
        st = None if stop is None else np.ascontiguousarray(stop, np.int32)
        if bScaleOpt is not None:
            scale = np.ones(1)
            check(lib().vieo_global_bundle_adjustment_vio_scale(
                params.ctypes.data, int(nIterations), int(bool(bRobust)), int(bool(bScaleOpt)), kfs.ctypes.data, len(kfs),
                points.ctypes.data, len(points), obs.ctypes.data, len(obs), imu.ctypes.data, len(imu),
                None if st is None else st.ctypes.data, navs.ctypes.data, pts.ctypes.data, res.ctypes.data,
                scale.ctypes.data), "vieo_global_bundle_adjustment_vio_scale")
            return navs, pts, res[0], float(scale[0])
        check(lib().vieo_global_bundle_adjustment_vio(
            params.ctypes.data, int(nIterations), int(bool(bRobust)), kfs.ctypes.data, len(kfs), points.ctypes.data,
            len(points), obs.ctypes.data, len(obs), imu.ctypes.data, len(imu),
            None if st is None else st.ctypes.data, navs.ctypes.data, pts.ctypes.data, res.ctypes.data),
            "vieo_global_bundle_adjustment_vio")
        return navs, pts, res[0]

    @staticmethod
    def GlobalBundleAdjustmentNavStatePRVSharded(shard, reduce_ptr, reduce_doubles, allreduce=None, nIterations=5,
                                                 bRobust=True, comm=None, bScaleOpt=None):
        """This rank's landmark shard (sharding.shard_window of (params, kfs, points, close, obs, imu)) of a full
        BA; reduce_ptr / allreduce as in LocalBundleAdjustmentNavStatePRVSharded.  returns (navs, points, result)
        (+ the recovered scale when bScaleOpt is given)."""
        import ctypes
        params, kfs, points, close, obs, imu = shard
        params, kfs, imu = np.ascontiguousarray(params), np.ascontiguousarray(kfs), np.ascontiguousarray(imu)
        points, obs = np.ascontiguousarray(points, np.float32), np.ascontiguousarray(obs)
        navs, pts, res = np.zeros(len(kfs), NAVSTATE_DTYPE), np.zeros_like(points), np.zeros(1, LBA_RESULT_DTYPE)
        CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
        base = int(reduce_ptr)

        def _cb(ctx, d_buf, n):
            try:
                return int(allreduce((int(d_buf) - base) // 8, int(n)))
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        cb = CB(_cb)
        if bScaleOpt is not None:
            scale = np.ones(1)
            check(lib().vieo_global_bundle_adjustment_vio_sharded_scale(
                params.ctypes.data, int(nIterations), int(bool(bRobust)), int(bool(bScaleOpt)), kfs.ctypes.data, len(kfs),
                points.ctypes.data, len(points), obs.ctypes.data, len(obs), imu.ctypes.data, len(imu),
                ctypes.c_void_p(base), reduce_doubles, None if comm is not None else ctypes.cast(cb, ctypes.c_void_p),
                None if comm is None else ctypes.c_void_p(int(comm)), navs.ctypes.data, pts.ctypes.data,
                res.ctypes.data, scale.ctypes.data), "vieo_global_bundle_adjustment_vio_sharded_scale")
            return navs, pts, res[0], float(scale[0])
        check(lib().vieo_global_bundle_adjustment_vio_sharded(
            params.ctypes.data, int(nIterations), int(bool(bRobust)), kfs.ctypes.data, len(kfs), points.ctypes.data,
            len(points), obs.ctypes.data, len(obs), imu.ctypes.data, len(imu), ctypes.c_void_p(base),
            reduce_doubles, None if comm is not None else ctypes.cast(cb, ctypes.c_void_p),
            None if comm is None else ctypes.c_void_p(int(comm)), navs.ctypes.data, pts.ctypes.data,
            res.ctypes.data), "vieo_global_bundle_adjustment_vio_sharded")
        return navs, pts, res[0]

