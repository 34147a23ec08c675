"""ctypes wrapper of the CPU oracle (oracle/_build/liboracle.so).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


def build(native=False):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"),
                           "NATIVE=%d" % int(native)])
    name = "liboracle_native.so" if native else "liboracle.so"
    return os.path.join(ROOT, "oracle", "_build", name)


class Oracle:
    def __init__(self, path):
        L = ctypes.CDLL(path)
        L.vo_orb_create.restype = P
        L.vo_orb_create.argtypes = [I, F, I, I, I]
        L.vo_orb_destroy.argtypes = [P]
        L.vo_orb_extract.argtypes = [P, P, I, I, I, P, P, P, I, P]
        L.vo_orb_features_per_level.argtypes = [P, I]
        L.vo_orb_scale_factor.argtypes = [P, I]
        L.vo_orb_scale_factor.restype = F
        L.vo_orb_umax.argtypes = [P, I]
        L.vo_orb_tie_count.argtypes = [P]
        L.vo_orb_tie_count.restype = ctypes.c_long
        L.vo_orb_level_size.argtypes = [P, I, P, P]
        L.vo_orb_get_plane.argtypes = [P, I, I, P]
        L.vo_orb_get_candidates.argtypes = [P, I, P, I]
        L.vo_orb_get_level_keys.argtypes = [P, I, P, I]
        L.vo_resize_linear_u8.argtypes = [P, I, I, P, I, I]
        L.vo_gaussian_blur7.argtypes = [P, I, I, P]
        L.vo_fast.argtypes = [P, I, I, I, P, I]
        L.vo_fast_atan2.argtypes = [F, F]
        L.vo_fast_atan2.restype = F
        L.vo_cv_round_f.argtypes = [F]
        L.vo_sincos_ref.argtypes = [F, P, P]
        L.vo_distribute_octtree.argtypes = [P, I, I, I, I, I, I, P, I]
        L.vo_descriptor_distance.argtypes = [P, P]
        L.vo_knn2_hamming.argtypes = [P, I, P, I, P, P]
        L.vo_stereo_match_rectified.argtypes = [P, P, I, P, I, P, P, I, P, P, P, F, F, P, P]
        L.vo_sbp_project_last_frame.argtypes = [P, I, P, P]
        L.vo_search_by_projection.argtypes = [I, P, I, P, P, P, P, I, P, F, I, P]
        L.vo_local_bundle_adjustment.argtypes = [P, P, I, P, I, P, I, P, P, P, P, P]
        L.vo_pose_optimization.argtypes = [P, P, P, P]
        L.vo_pose_optimization_vio.argtypes = [P, P, P, P]
        L.vo_imu_edge_eval.argtypes = [P, P, P, P, P, P, P]
        L.vo_navstate_inc.argtypes = [P, P, P]
        L.vo_pose_edge_eval.argtypes = [P, P, P, P, P]
        L.vo_so3_exp.argtypes = [P, P]
        L.vo_so3_log.argtypes = [P, P]
        L.vo_so3_jr.argtypes = [P, P, I]
        self.L = L

    # ---- projection search
    def sbp_project_last_frame(self, pts, cam, rig=None):
        from vieo_slam_amd.ba_types import PROJ_QUERY_DTYPE
        pts = np.ascontiguousarray(pts)
        cam = np.ascontiguousarray(cam)
        if rig is None:
            q = np.zeros(len(pts), PROJ_QUERY_DTYPE)
            self.L.vo_sbp_project_last_frame(pts.ctypes.data, len(pts), cam.ctypes.data, q.ctypes.data)
            return q
        rig = np.ascontiguousarray(rig)
        q = np.zeros(len(pts) * int(rig[0]["n_cams"]), PROJ_QUERY_DTYPE)
        self.L.vo_sbp_project_last_frame_rig.argtypes = [P, I, P, P, P]
        self.L.vo_sbp_project_last_frame_rig(pts.ctypes.data, len(pts), cam.ctypes.data, rig.ctypes.data, q.ctypes.data)
        return q

    def sbp_project_keyframe(self, pts, cam, rig, log_scale_factor):
        from vieo_slam_amd.ba_types import PROJ_QUERY_DTYPE
        pts = np.ascontiguousarray(pts)
        cam = np.ascontiguousarray(cam)
        nc = 1 if rig is None else int(rig[0]["n_cams"])
        rig = None if rig is None else np.ascontiguousarray(rig)
        q = np.zeros(len(pts) * nc, PROJ_QUERY_DTYPE)
        self.L.vo_sbp_project_keyframe.argtypes = [P, I, P, P, F, P]
        self.L.vo_sbp_project_keyframe(pts.ctypes.data, len(pts), cam.ctypes.data,
                                       None if rig is None else rig.ctypes.data, float(log_scale_factor), q.ctypes.data)
        return q

    def search_by_projection(self, mode, queries, keys, uright, desc, taken, bounds, nn_ratio=0.6,
                             check_ori=True, cam_first=None):
        queries = np.ascontiguousarray(queries)
        keys = np.ascontiguousarray(keys)
        uright = np.ascontiguousarray(uright, np.float32)
        desc = np.ascontiguousarray(desc, np.uint8)
        tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
        b = np.ascontiguousarray(bounds, np.float32)
        assign = np.zeros(max(len(keys), 1), np.int32)
        if cam_first is None:
            n = self.L.vo_search_by_projection(mode, queries.ctypes.data, len(queries), keys.ctypes.data,
                                               uright.ctypes.data, desc.ctypes.data,
                                               None if tk is None else tk.ctypes.data, len(keys),
                                               b.ctypes.data, nn_ratio, int(check_ori), assign.ctypes.data)
        else:
            cf = np.ascontiguousarray(cam_first, np.int32)
            self.L.vo_search_by_projection_rig.argtypes = [I, P, I, P, P, P, P, I, P, P, I, F, I, P]
            n = self.L.vo_search_by_projection_rig(mode, queries.ctypes.data, len(queries), keys.ctypes.data,
                                                   uright.ctypes.data, desc.ctypes.data,
                                                   None if tk is None else tk.ctypes.data, len(keys), cf.ctypes.data,
                                                   b.ctypes.data, len(cf) - 1, nn_ratio, int(check_ori),
                                                   assign.ctypes.data)
        return n, assign[:len(keys)]

    # ---- local bundle adjustment
    def search_for_triangulation(self, kf1, kf2s, only_stereo=False, check_orientation=True, pair_capacity=None):
        from vieo_slam_amd.tri_search import tri_call
        P, I = ctypes.c_void_p, ctypes.c_int
        self.L.vo_search_for_triangulation.argtypes = [P, P, I, I, I, I, I, P, P, P]
        self.L.vo_search_for_triangulation.restype = I
        rc, out = tri_call(self.L.vo_search_for_triangulation, kf1, kf2s, only_stereo, check_orientation, pair_capacity)
        return out

    def tri_gates(self, kf1, kf2, idx1, idx2):
        P, I = ctypes.c_void_p, ctypes.c_int
        self.L.vo_tri_gates.argtypes = [P, P, I, I, P]
        self.L.vo_tri_gates.restype = I
        ep = np.zeros(2, np.float32)
        d = self.L.vo_tri_gates(kf1.rec.ctypes.data, kf2.rec.ctypes.data, int(idx1), int(idx2), ep.ctypes.data)
        return d, ep

    def local_ba(self, params, kfs, points, obs, stop=None, enc=None):
        from vieo_slam_amd.ba_types import LBA_RESULT_DTYPE, NAVSTATE_DTYPE
        params, kfs = np.ascontiguousarray(params), np.ascontiguousarray(kfs)
        points, obs = np.ascontiguousarray(points, np.float32), np.ascontiguousarray(obs)
        navs = np.zeros(len(kfs), NAVSTATE_DTYPE)
        pts = np.zeros_like(points)
        erase = np.zeros(max(len(obs), 1), np.uint8)
        res = np.zeros(1, LBA_RESULT_DTYPE)
        st = None if stop is None else np.ascontiguousarray(stop, np.int32)
        if enc is not None:
            P, I = ctypes.c_void_p, ctypes.c_int
            self.L.vo_local_bundle_adjustment_enc.argtypes = [P, P, I, P, I, P, I, P, P, P, P, P, P]
            self.L.vo_local_bundle_adjustment_enc(params.ctypes.data, kfs.ctypes.data, len(kfs), points.ctypes.data,
                                                  len(points), obs.ctypes.data, len(obs),
                                                  np.ascontiguousarray(enc).ctypes.data,
                                                  None if st is None else st.ctypes.data, navs.ctypes.data,
                                                  pts.ctypes.data, erase.ctypes.data, res.ctypes.data)
            return navs, pts, erase[:len(obs)], res[0]
        self.L.vo_local_bundle_adjustment(params.ctypes.data, kfs.ctypes.data, len(kfs),
                                          points.ctypes.data, len(points), obs.ctypes.data, len(obs),
                                          None if st is None else st.ctypes.data, navs.ctypes.data,
                                          pts.ctypes.data, erase.ctypes.data, res.ctypes.data)
        return navs, pts, erase[:len(obs)], res[0]

    def bundle_adjustment(self, params, kfs, points, obs, n_iterations=5, robust=True, stop=None, enc=None):
        from vieo_slam_amd.ba_types import LBA_RESULT_DTYPE, NAVSTATE_DTYPE
        params, kfs = np.ascontiguousarray(params), np.ascontiguousarray(kfs)
        points, obs = np.ascontiguousarray(points, np.float32), np.ascontiguousarray(obs)
        navs, pts, res = np.zeros(len(kfs), NAVSTATE_DTYPE), np.zeros_like(points), np.zeros(1, LBA_RESULT_DTYPE)
        st = None if stop is None else np.ascontiguousarray(stop, np.int32)
        P, I = ctypes.c_void_p, ctypes.c_int
        if enc is not None:
            self.L.vo_bundle_adjustment_enc.argtypes = [P, I, I, P, I, P, I, P, I, P, P, P, P, P]
            self.L.vo_bundle_adjustment_enc(params.ctypes.data, int(n_iterations), int(bool(robust)), kfs.ctypes.data,
                                            len(kfs), points.ctypes.data, len(points), obs.ctypes.data, len(obs),
                                            np.ascontiguousarray(enc).ctypes.data,
                                            None if st is None else st.ctypes.data, navs.ctypes.data, pts.ctypes.data,
                                            res.ctypes.data)
            return navs, pts, res[0]
        self.L.vo_bundle_adjustment.argtypes = [P, I, I, P, I, P, I, P, I, P, P, P, P]
        self.L.vo_bundle_adjustment(params.ctypes.data, int(n_iterations), int(bool(robust)), kfs.ctypes.data,
                                    len(kfs), points.ctypes.data, len(points), obs.ctypes.data, len(obs),
                                    None if st is None else st.ctypes.data, navs.ctypes.data, pts.ctypes.data,
                                    res.ctypes.data)
        return navs, pts, res[0]

    def global_ba_vio(self, params, kfs, points, obs, imu, n_iterations=5, robust=True, stop=None, scale_opt=None):
        """scale_opt is None: (navs, points, result); otherwise bScaleOpt = scale_opt (System::FinalGBA passes true)
        and the recovered VertexScale estimate comes back as a fourth value."""
        from vieo_slam_amd.ba_types import LBA_RESULT_DTYPE, NAVSTATE_DTYPE
        params, kfs, imu = np.ascontiguousarray(params), np.ascontiguousarray(kfs), np.ascontiguousarray(imu)
        points, obs = np.ascontiguousarray(points, np.float32), np.ascontiguousarray(obs)
        navs, pts, res = np.zeros(len(kfs), NAVSTATE_DTYPE), np.zeros_like(points), np.zeros(1, LBA_RESULT_DTYPE)
        st = None if stop is None else np.ascontiguousarray(stop, np.int32)
        P, I = ctypes.c_void_p, ctypes.c_int
        if scale_opt is not None:
            scale = np.ones(1)
            self.L.vo_global_bundle_adjustment_vio_scale.argtypes = [P, I, I, I, P, I, P, I, P, I, P, I, P, P, P, P, P]
            self.L.vo_global_bundle_adjustment_vio_scale(
                params.ctypes.data, int(n_iterations), int(bool(robust)), int(bool(scale_opt)), kfs.ctypes.data, len(kfs),
                points.ctypes.data, len(points), obs.ctypes.data, len(obs), imu.ctypes.data, len(imu),
                None if st is None else st.ctypes.data, navs.ctypes.data, pts.ctypes.data, res.ctypes.data,
                scale.ctypes.data)
            return navs, pts, res[0], float(scale[0])
        self.L.vo_global_bundle_adjustment_vio.argtypes = [P, I, I, P, I, P, I, P, I, P, I, P, P, P, P]
        self.L.vo_global_bundle_adjustment_vio(params.ctypes.data, int(n_iterations), int(bool(robust)),
                                               kfs.ctypes.data, len(kfs), points.ctypes.data, len(points),
                                               obs.ctypes.data, len(obs), imu.ctypes.data, len(imu),
                                               None if st is None else st.ctypes.data, navs.ctypes.data,
                                               pts.ctypes.data, res.ctypes.data)
        return navs, pts, res[0]

    def local_ba_vio(self, params, kfs, points, close, obs, imu, stop=None):
        from vieo_slam_amd.ba_types import LBA_RESULT_DTYPE, NAVSTATE_DTYPE
        params, kfs, imu = np.ascontiguousarray(params), np.ascontiguousarray(kfs), np.ascontiguousarray(imu)
        points, obs = np.ascontiguousarray(points, np.float32), np.ascontiguousarray(obs)
        close = np.ascontiguousarray(close, np.uint8)
        navs = np.zeros(len(kfs), NAVSTATE_DTYPE)
        pts = np.zeros_like(points)
        erase = np.zeros(max(len(obs), 1), np.uint8)
        res = np.zeros(1, LBA_RESULT_DTYPE)
        st = None if stop is None else np.ascontiguousarray(stop, np.int32)
        P = ctypes.c_void_p
        self.L.vo_local_bundle_adjustment_vio.argtypes = [P, P, ctypes.c_int, P, P, ctypes.c_int, P, ctypes.c_int,
                                                          P, ctypes.c_int, P, P, P, P, P]
        self.L.vo_local_bundle_adjustment_vio(params.ctypes.data, kfs.ctypes.data, len(kfs), points.ctypes.data,
                                              close.ctypes.data, len(points), obs.ctypes.data, len(obs),
                                              imu.ctypes.data, len(imu), None if st is None else st.ctypes.data,
                                              navs.ctypes.data, pts.ctypes.data, erase.ctypes.data, res.ctypes.data)
        return navs, pts, erase[:len(obs)], res[0]

    def lba_prs_edge_eval(self, params, ns, Xh, scale, ob, jac=True):
        """EdgeReprojectPRS / PRSStereo at (key-frame state, unscaled point, scale): err[3], Jp[3,6], Jx[3,3], Js[3]."""
        err, Jp, Jx, Js = np.zeros(3), np.zeros((3, 6)), np.zeros((3, 3)), np.zeros(3)
        P = ctypes.c_void_p
        self.L.vo_lba_prs_edge_eval.argtypes = [P, P, P, ctypes.c_double, P, P, P, P, P]
        params = np.ascontiguousarray(params)
        a, o = np.zeros(1, ns.dtype), np.zeros(1, ob.dtype)
        a[0], o[0] = ns, ob
        X = np.ascontiguousarray(Xh, np.float64)
        self.L.vo_lba_prs_edge_eval(params.ctypes.data, a.ctypes.data, X.ctypes.data, float(scale), o.ctypes.data,
                                    err.ctypes.data, Jp.ctypes.data if jac else None, Jx.ctypes.data, Js.ctypes.data)
        return err, Jp, Jx, Js

    def lba_imu_edge_eval(self, params, edge, nsi, nsj, jac=True):
        err = np.zeros(15)
        J = np.zeros((9, 24))
        P = ctypes.c_void_p
        self.L.vo_lba_imu_edge_eval.argtypes = [P] * 6
        a, b = np.zeros(1, nsi.dtype), np.zeros(1, nsj.dtype)
        a[0], b[0] = nsi, nsj
        nsi, nsj = a, b
        params, edge = np.ascontiguousarray(params), np.ascontiguousarray(edge)
        self.L.vo_lba_imu_edge_eval(params.ctypes.data, edge.ctypes.data, nsi.ctypes.data, nsj.ctypes.data,
                                    err.ctypes.data, J.ctypes.data if jac else None)
        return err, J

    def lba_navstate_inc(self, ns, d15):
        out = np.zeros(1, ns.dtype)  # (np.array(np.void) would alias the caller's record)
        out[0] = ns
        d = np.ascontiguousarray(d15, np.float64)
        P = ctypes.c_void_p
        self.L.vo_lba_navstate_inc.argtypes = [P, P]
        self.L.vo_lba_navstate_inc(out.ctypes.data, d.ctypes.data)
        return out[0]

    def cam_project(self, cam, P, jac=True):
        cam = np.ascontiguousarray(cam).reshape(1)
        P = np.ascontiguousarray(P, np.float64)
        uv = np.zeros(2, np.float32)
        J = np.zeros((2, 3))
        V = ctypes.c_void_p
        self.L.vo_cam_project.argtypes = [V, V, V, V]
        self.L.vo_cam_project(cam.ctypes.data, P.ctypes.data, uv.ctypes.data, J.ctypes.data if jac else None)
        return uv, J

    # ---- pose optimisation
    def pose_optimization(self, frame, obs):
        from vieo_slam_amd.ba_types import POSE_RESULT_DTYPE
        frame = np.ascontiguousarray(frame)
        obs = np.ascontiguousarray(obs)
        outl = np.zeros(len(obs), np.uint8)
        res = np.zeros(1, POSE_RESULT_DTYPE)
        self.L.vo_pose_optimization(frame.ctypes.data, obs.ctypes.data, outl.ctypes.data,
                                    res.ctypes.data)
        return res[0], outl

    def pose_optimization_vio(self, frame, obs):
        from vieo_slam_amd.ba_types import VIO_RESULT_DTYPE
        frame = np.ascontiguousarray(frame)
        obs = np.ascontiguousarray(obs)
        outl = np.zeros(max(len(obs), 1), np.uint8)
        res = np.zeros(1, VIO_RESULT_DTYPE)
        self.L.vo_pose_optimization_vio(frame.ctypes.data, obs.ctypes.data, outl.ctypes.data,
                                        res.ctypes.data)
        return res[0], outl[:len(obs)]

    def imu_edge_eval(self, frame, nsi, nsj, want_jac=True):
        err = np.zeros(9)
        Ji, Jj, JB = np.zeros((9, 9)), np.zeros((9, 9)), np.zeros((9, 6))
        nsi, nsj = np.ascontiguousarray(nsi), np.ascontiguousarray(nsj)
        self.L.vo_imu_edge_eval(np.ascontiguousarray(frame).ctypes.data, nsi.ctypes.data,
                                nsj.ctypes.data, err.ctypes.data,
                                Ji.ctypes.data if want_jac else None, Jj.ctypes.data, JB.ctypes.data)
        return err, Ji, Jj, JB

    def navstate_inc(self, ns, dpvr=None, dbias=None):
        out = np.ascontiguousarray(ns).copy()
        a = None if dpvr is None else np.ascontiguousarray(dpvr, np.float64)
        b = None if dbias is None else np.ascontiguousarray(dbias, np.float64)
        self.L.vo_navstate_inc(out.ctypes.data, None if a is None else a.ctypes.data,
                               None if b is None else b.ctypes.data)
        return out

    def pose_edge_eval(self, frame, ob, delta=None, want_jac=True):
        err = np.zeros(3)
        J = np.zeros((3, 6))
        d = None if delta is None else np.ascontiguousarray(delta, np.float64)
        self.L.vo_pose_edge_eval(np.ascontiguousarray(frame).ctypes.data,
                                 np.ascontiguousarray(ob).ctypes.data,
                                 None if d is None else d.ctypes.data, err.ctypes.data,
                                 J.ctypes.data if want_jac else None)
        return err, J

    def so3_exp(self, w):
        w = np.ascontiguousarray(w, np.float64)
        q = np.zeros(4)
        self.L.vo_so3_exp(w.ctypes.data, q.ctypes.data)
        return q

    def so3_log(self, q):
        q = np.ascontiguousarray(q, np.float64)
        w = np.zeros(3)
        self.L.vo_so3_log(q.ctypes.data, w.ctypes.data)
        return w

    def so3_jr(self, w, inverse=False):
        w = np.ascontiguousarray(w, np.float64)
        J = np.zeros((3, 3))
        self.L.vo_so3_jr(w.ctypes.data, J.ctypes.data, int(inverse))
        return J

    # ---- matching
    def descriptor_distance(self, a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return self.L.vo_descriptor_distance(a.ctypes.data, b.ctypes.data)

    def stereo_fisheye(self, params, keys, descs, num_mono, group_capacity=None):
        from vieo_slam_amd.matching import fisheye_call
        P = ctypes.c_void_p
        self.L.vo_stereo_fisheye_match.argtypes = [P, P, P, P, P, ctypes.c_int, P, P, P, P, P, P, P]
        rc, out = fisheye_call(self.L.vo_stereo_fisheye_match, params, keys, descs, num_mono, group_capacity)
        assert rc == 0, rc
        return out

    def is_in_frustum(self, frame, points):
        from vieo_slam_amd.map_point import frustum_call
        P = ctypes.c_void_p
        self.L.vo_is_in_frustum_batch.argtypes = [P, P, ctypes.c_int, P]
        self.L.vo_is_in_frustum_batch.restype = None
        return frustum_call(self.L.vo_is_in_frustum_batch, frame, points)[1]

    def distinctive_descriptors(self, descriptors, first):
        from vieo_slam_amd.map_point import distinctive_call
        P = ctypes.c_void_p
        self.L.vo_distinctive_descriptors_batch.argtypes = [P, P, ctypes.c_int, P]
        self.L.vo_distinctive_descriptors_batch.restype = None
        return distinctive_call(self.L.vo_distinctive_descriptors_batch, descriptors, first)[1]

    def update_normal_and_depth(self, points, first, obs_centre, centres, ref_centre, ref_scale, scale_last):
        from vieo_slam_amd.map_point import normal_depth_call
        P = ctypes.c_void_p
        self.L.vo_update_normal_and_depth_batch.argtypes = [P, P, P, P, P, P, ctypes.c_float, ctypes.c_int, P, P, P]
        self.L.vo_update_normal_and_depth_batch.restype = None
        return normal_depth_call(self.L.vo_update_normal_and_depth_batch, points, first, obs_centre, centres,
                                 ref_centre, ref_scale, scale_last, oracle=True)[1:]

    def fuse_search(self, frame, keys, uright, descs, points):
        from vieo_slam_amd.map_point import fuse_call
        P = ctypes.c_void_p
        self.L.vo_fuse_search.argtypes = [P, P, P, P, P, P, ctypes.c_int, P, P]
        self.L.vo_fuse_search.restype = None
        return fuse_call(self.L.vo_fuse_search, frame, keys, uright, descs, points)[1:]

    def imu_preintegrate(self, noise, sample_lists, ti, tj, bg, ba):
        from vieo_slam_amd.imu import preint_call
        P = ctypes.c_void_p
        self.L.vo_imu_preintegrate_batch.argtypes = [P, P, P, P, P, P, P, ctypes.c_int, P, P, P]
        self.L.vo_imu_preintegrate_batch.restype = None
        return preint_call(self.L.vo_imu_preintegrate_batch, noise, sample_lists, ti, tj, bg, ba)[1:]

    def enc_edge_eval(self, nsi, nsj, meas, qRbe, pbe, jac=True):
        a, b = np.zeros(1, nsi.dtype), np.zeros(1, nsj.dtype)
        a[0], b[0] = nsi, nsj
        meas, qRbe, pbe = (np.ascontiguousarray(x, np.float64) for x in (meas, qRbe, pbe))
        err, Ji, Jj = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
        P = ctypes.c_void_p
        self.L.vo_enc_edge_eval.argtypes = [P] * 8
        self.L.vo_enc_edge_eval.restype = None
        self.L.vo_enc_edge_eval(a.ctypes.data, b.ctypes.data, meas.ctypes.data, qRbe.ctypes.data, pbe.ctypes.data,
                                err.ctypes.data, Ji.ctypes.data if jac else None, Jj.ctypes.data if jac else None)
        return err, Ji, Jj

    def fisheye_branch_counts(self, reset=True):
        """(new group, extension, member replaced, contradiction kept, contradiction swapped) since the last reset"""
        out = (ctypes.c_long * 5)()
        self.L.vo_fisheye_branch_counts.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.L.vo_fisheye_branch_counts(out, int(reset))
        return tuple(out)

    def cam_unproject(self, cam, uv):
        cam = np.ascontiguousarray(cam).reshape(1)
        uv = np.ascontiguousarray(uv, np.float32)
        out = np.zeros(3, np.float64)
        self.L.vo_cam_unproject.argtypes = [ctypes.c_void_p] * 3
        self.L.vo_cam_unproject(cam.ctypes.data, uv.ctypes.data, out.ctypes.data)
        return out

    def null_vector4(self, A):
        A = np.ascontiguousarray(A, np.float64)
        out = np.zeros(4, np.float64)
        self.L.vo_null_vector4.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self.L.vo_null_vector4(A.ctypes.data, A.shape[0], out.ctypes.data)
        return out

    def knn2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.zeros((len(q), 2), np.int32)
        dist = np.zeros((len(q), 2), np.int32)
        self.L.vo_knn2_hamming(q.ctypes.data, len(q), t.ctypes.data, len(t), idx.ctypes.data,
                               dist.ctypes.data)
        return idx, dist

    def stereo_match(self, extL, extR, kl, dl, kr, dr, baseline, bf):
        """extL/extR: OracleExtractor objects that just processed the left/right image."""
        kl, kr = np.ascontiguousarray(kl), np.ascontiguousarray(kr)
        dl, dr = np.ascontiguousarray(dl, np.uint8), np.ascontiguousarray(dr, np.uint8)
        sc = np.array(extL.scale_factors(), np.float32)
        inv = (np.float32(1.0) / sc).astype(np.float32)
        ur = np.zeros(len(kl), np.float32)
        dp = np.zeros(len(kl), np.float32)
        self.L.vo_stereo_match_rectified(extL.h, extR.h, extL.nlevels, kl.ctypes.data, len(kl),
                                         dl.ctypes.data, kr.ctypes.data, len(kr), dr.ctypes.data,
                                         sc.ctypes.data, inv.ctypes.data, baseline, bf,
                                         ur.ctypes.data, dp.ctypes.data)
        return ur, dp

    # ---- primitives
    def resize(self, src, dw, dh):
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros((dh, dw), np.uint8)
        self.L.vo_resize_linear_u8(src.ctypes.data, src.shape[1], src.shape[0], dst.ctypes.data, dw, dh)
        return dst

    def blur(self, src):
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.zeros_like(src)
        self.L.vo_gaussian_blur7(src.ctypes.data, src.shape[1], src.shape[0], dst.ctypes.data)
        return dst

    def fast(self, src, threshold, cap=100000):
        src = np.ascontiguousarray(src, np.uint8)
        out = np.zeros((cap, 3), np.int32)
        n = self.L.vo_fast(src.ctypes.data, src.shape[1], src.shape[0], threshold, out.ctypes.data, cap)
        assert n >= 0
        return out[:n].copy()

    def fast_atan2(self, y, x):
        return self.L.vo_fast_atan2(y, x)

    def cv_round(self, v):
        return self.L.vo_cv_round_f(v)

    def sincos(self, angle_deg):
        c, s = F(), F()
        self.L.vo_sincos_ref(angle_deg, ctypes.byref(c), ctypes.byref(s))
        return c.value, s.value

    def distribute(self, xyr, minX, maxX, minY, maxY, N):
        xyr = np.ascontiguousarray(xyr, np.int32)
        out = np.zeros((N + 64, 3), np.int32)
        n = self.L.vo_distribute_octtree(xyr.ctypes.data, len(xyr), minX, maxX, minY, maxY, N,
                                         out.ctypes.data, N + 64)
        assert n >= 0, n
        return out[:n].copy()

    def extractor(self, nfeatures=1200, scale=1.2, nlevels=8, ini=20, mn=7):
        return OracleExtractor(self.L, nfeatures, scale, nlevels, ini, mn)


class OracleExtractor:
    def __init__(self, L, nfeatures, scale, nlevels, ini, mn):
        self.L = L
        self.nlevels = nlevels
        self.h = P(L.vo_orb_create(nfeatures, scale, nlevels, ini, mn))

    def __del__(self):
        try:
            self.L.vo_orb_destroy(self.h)
        except Exception:
            pass

    def __call__(self, image, lapping=None, cap=20000):
        img = np.ascontiguousarray(image, np.uint8)
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = I()
        lap = None
        if lapping is not None:
            lap = (I * 2)(int(lapping[0]), int(lapping[1]))
        r = self.L.vo_orb_extract(self.h, img.ctypes.data, img.shape[1], img.shape[0], img.strides[0],
                                  lap, kps.ctypes.data, desc.ctypes.data, cap, ctypes.byref(n))
        assert r != -2, "oracle capacity"
        if r == -1:
            return -1, kps[:0], desc[:0]
        return r, kps[:n.value].copy(), desc[:n.value].copy()

    def features_per_level(self):
        return [self.L.vo_orb_features_per_level(self.h, l) for l in range(self.nlevels)]

    def scale_factors(self):
        return [self.L.vo_orb_scale_factor(self.h, l) for l in range(self.nlevels)]

    def tie_count(self):
        return self.L.vo_orb_tie_count(self.h)

    def level_size(self, l):
        w, h = I(), I()
        self.L.vo_orb_level_size(self.h, l, ctypes.byref(w), ctypes.byref(h))
        return w.value, h.value

    def plane(self, l, which=0):
        w, h = self.level_size(l)
        if which == 2:
            w, h = w + 38, h + 38
        out = np.zeros((h, w), np.uint8)
        self.L.vo_orb_get_plane(self.h, l, which, out.ctypes.data)
        return out

    def candidates(self, l, cap=200000):
        out = np.zeros((cap, 3), np.int32)
        n = self.L.vo_orb_get_candidates(self.h, l, out.ctypes.data, cap)
        assert n >= 0
        return out[:n].copy()

    def level_keys(self, l, cap=20000):
        out = np.zeros(cap, KEYPOINT_DTYPE)
        n = self.L.vo_orb_get_level_keys(self.h, l, out.ctypes.data, cap)
        assert n >= 0
        return out[:n].copy()


_cached = None


def load():
    global _cached
    if _cached is None:
        _cached = Oracle(build())
    return _cached
