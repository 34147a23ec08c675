"""Host-side mirrors of the reference's stereo descriptor searches on top of the C-ABI:
  knn_match2(query, train)             cv::BFMatcher(NORM_HAMMING).knnMatch(k=2)  (Frame.cc:620-628)
  compute_stereo_matches(extL, extR..) Frame::ComputeStereoMatches                (Frame.cc:451-611)
  compute_stereo_fisheye_matches(..)   Frame::ComputeStereoFishEyeMatches         (Frame.cc:613-779)
"""
import numpy as np

from ._lib import check, lib


def knn_match2(query, train):
    q = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
    idx = np.full((len(q), 2), -1, np.int32)
    dist = np.full((len(q), 2), np.iinfo(np.int32).max, np.int32)
    check(lib().vieo_hamming_knn2(q.ctypes.data, len(q), t.ctypes.data, len(t), idx.ctypes.data,
                                  dist.ctypes.data), "vieo_hamming_knn2")
    return idx, dist


def compute_stereo_matches(ext_left, ext_right, kps_l, desc_l, kps_r, desc_r, baseline, bf):
    """returns (vuright, vdepth) float32[len(kps_l)], -1 where unmatched."""
    kl, kr = np.ascontiguousarray(kps_l), np.ascontiguousarray(kps_r)
    dl, dr = np.ascontiguousarray(desc_l, np.uint8), np.ascontiguousarray(desc_r, np.uint8)
    ur = np.full(len(kl), -1, np.float32)
    dp = np.full(len(kl), -1, np.float32)
    check(lib().vieo_stereo_match_rectified(ext_left._h, ext_right._h, kl.ctypes.data,
                                            dl.ctypes.data, len(kl), kr.ctypes.data, dr.ctypes.data,
                                            len(kr), baseline, bf, ur.ctypes.data, dp.ctypes.data),
          "vieo_stereo_match_rectified")
    return ur, dp


def compute_stereo_matches_resident(ext_left, ext_right, baseline, bf):
    """Frame::ComputeStereoMatches of the frame the two extractors have just processed: keys, descriptors and pyramids
    are read in HBM where ORBextractor.__call__ left them (vieo_stereo_match_rectified_resident)."""
    n = lib().vieo_orb_resident_keys(ext_left._h)
    assert n >= 0, "the left extractor holds no frame"
    ur = np.full(max(n, 1), -1, np.float32)
    dp = np.full(max(n, 1), -1, np.float32)
    check(lib().vieo_stereo_match_rectified_resident(ext_left._h, ext_right._h, baseline, bf, ur.ctypes.data,
                                                     dp.ctypes.data), "vieo_stereo_match_rectified_resident")
    return ur[:n], dp[:n]


def fisheye_call(fn, params, keys, descs, num_mono, group_capacity=None):
    """Marshals one ComputeStereoFishEyeMatches call for `fn` (the C-ABI entry, or the test oracle's function of
    the same signature).  params: FISHEYE_PARAMS_DTYPE[1]; keys[c]: KEYPOINT_DTYPE[n_c]; descs[c]: uint8[n_c, 32].
    returns dict(depth float32[N], key_group int32[N], group_idx int32[G, n_cams], group_good bool[G],
    group_p3d float64[G, 3], n_matches) with N = sum n_c in mvKeys (camera-major) order."""
    import ctypes
    nc = int(params[0]["n_cams"])
    keys = [np.ascontiguousarray(k) for k in keys]
    descs = [np.ascontiguousarray(d, np.uint8).reshape(-1, 32) for d in descs]
    n_keys = np.array([len(k) for k in keys], np.int32)
    mono = np.ascontiguousarray(num_mono, np.int32)
    kp = (ctypes.c_void_p * nc)(*[k.ctypes.data for k in keys])
    dp = (ctypes.c_void_p * nc)(*[d.ctypes.data for d in descs])
    N = int(n_keys.sum())
    cap = int(group_capacity if group_capacity is not None else max(N, 1))
    depth, kg = np.zeros(max(N, 1), np.float32), np.zeros(max(N, 1), np.int32)
    gidx, good = np.zeros((cap, nc), np.int32), np.zeros(cap, np.uint8)
    p3d, ng, nm = np.zeros((cap, 3), np.float64), np.zeros(1, np.int32), np.zeros(1, np.int32)
    rc = fn(params.ctypes.data, ctypes.cast(kp, ctypes.c_void_p), ctypes.cast(dp, ctypes.c_void_p),
            n_keys.ctypes.data, mono.ctypes.data, cap, depth.ctypes.data, kg.ctypes.data, gidx.ctypes.data,
            good.ctypes.data, p3d.ctypes.data, ng.ctypes.data, nm.ctypes.data)
    g = int(ng[0])
    return rc, dict(depth=depth[:N], key_group=kg[:N], group_idx=gidx[:g], group_good=good[:g].astype(bool),
                    group_p3d=p3d[:g], n_matches=int(nm[0]))


def compute_stereo_fisheye_matches(params, keys, descs, num_mono, group_capacity=None):
    """void Frame::ComputeStereoFishEyeMatches(th_far_pts) (Frame.cc:613-779); see fisheye_call for the result."""
    rc, out = fisheye_call(lib().vieo_stereo_fisheye_match, params, keys, descs, num_mono, group_capacity)
    check(rc, "vieo_stereo_fisheye_match")
    return out


class FisheyeStereoDevice:
    """Handle of the device-resident stereo stage of camera-rig frames (vieo_fisheye_*): `match_batch` takes the
    extractor-layout arrays of n_frames rig frames ([frame][camera][cap]) already in HBM (DeviceBuffer) or as numpy
    arrays (uploaded here, test convenience) and returns the per-frame outputs downloaded as numpy arrays."""

    def __init__(self, params, key_cap, max_frames=1):
        import ctypes
        self.params = np.ascontiguousarray(params)
        self.nc, self.cap, self.max_frames = int(params[0]["n_cams"]), int(key_cap), int(max_frames)
        h = ctypes.c_void_p()
        check(lib().vieo_fisheye_create(ctypes.byref(h), self.params.ctypes.data, self.cap, self.max_frames),
              "vieo_fisheye_create")
        self.h = h
        self.gcap = lib().vieo_fisheye_group_capacity(self.h)

    def close(self):
        if self.h:
            lib().vieo_fisheye_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def match_batch(self, frames):
        """frames: list of (keys [n_cams] KEYPOINT_DTYPE arrays, descs [n_cams], num_mono).  returns a list of dicts
        like fisheye_call's plus keys / desc (mvKeys order), cam_first, hdr."""
        from ._lib import DeviceBuffer
        from .orb_extractor import KEYPOINT_DTYPE
        nf, nc, cap, gcap = len(frames), self.nc, self.cap, self.gcap
        kc = nc * cap
        K = np.zeros((nf, nc, cap), KEYPOINT_DTYPE)
        D = np.zeros((nf, nc, cap, 32), np.uint8)
        C = np.zeros((nf, nc, 2), np.int32)
        for f, (keys, descs, mono) in enumerate(frames):
            for c in range(nc):
                n = len(keys[c])
                K[f, c, :n], D[f, c, :n] = keys[c], np.asarray(descs[c]).reshape(-1, 32)
                C[f, c] = (n, mono[c])
        bufs = {}
        for name, arr in (("K", K), ("D", D), ("C", C)):
            bufs[name] = DeviceBuffer(max(arr.nbytes, 16))
            bufs[name].upload(arr)
        outs = dict(kcat=(KEYPOINT_DTYPE, (nf, kc)), dcat=(np.uint8, (nf, kc, 32)), first=(np.int32, (nf, nc + 1)),
                    fcnt=(np.int32, (nf, 2)),
                    depth=(np.float32, (nf, kc)), ur=(np.float32, (nf, kc)), kg=(np.int32, (nf, kc)),
                    gidx=(np.int32, (nf, gcap, nc)), good=(np.uint8, (nf, gcap)), p3d=(np.float64, (nf, gcap, 3)),
                    hdr=(np.int32, (nf, 8)))
        for name, (dt, shp) in outs.items():
            bufs[name] = DeviceBuffer(max(int(np.prod(shp)) * np.dtype(dt).itemsize, 16))
        check(lib().vieo_stereo_fisheye_match_batch_device(
            self.h, bufs["K"].ptr, bufs["D"].ptr, bufs["C"].ptr, nf, bufs["kcat"].ptr, bufs["dcat"].ptr, bufs["first"].ptr,
            bufs["fcnt"].ptr, bufs["depth"].ptr, bufs["ur"].ptr, bufs["kg"].ptr, bufs["gidx"].ptr, bufs["good"].ptr, bufs["p3d"].ptr,
            bufs["hdr"].ptr, None), "vieo_stereo_fisheye_match_batch_device")
        check(lib().vieo_device_synchronize(), "sync")
        got = {name: bufs[name].download(dt, shp) for name, (dt, shp) in outs.items()}
        res = []
        for f in range(nf):
            N, g = int(got["first"][f, nc]), int(got["hdr"][f, 0])
            res.append(dict(depth=got["depth"][f, :N], key_group=got["kg"][f, :N], group_idx=got["gidx"][f, :g],
                            group_good=got["good"][f, :g].astype(bool), group_p3d=got["p3d"][f, :g],
                            n_matches=int(got["hdr"][f, 1]), keys=got["kcat"][f, :N], desc=got["dcat"][f, :N],
                            uright=got["ur"][f, :N], cam_first=got["first"][f], hdr=got["hdr"][f]))
        for b in bufs.values():
            b.free()
        return res


def fisheye_last_walk():
    """(rows walked, wavefront steps) of this thread's last compute_stereo_fisheye_matches (test tap)."""
    import ctypes
    a, b = ctypes.c_int32(), ctypes.c_int32()
    lib().vieo_fisheye_last_walk(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


class ORBmatcher:
    """ORBmatcher(nnratio=0.6, checkOri=True) (reference include/ORBmatcher.h:25-101), tracking-side
    searches on flattened inputs (ba_types.PROJ_QUERY_DTYPE etc.)."""
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30
    SBP_LAST_FRAME, SBP_LOCAL_MAP, SBP_RELOC = 0, 1, 2

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)

    @staticmethod
    def project_last_frame(points, cam, rig=None):
        """ORBmatcher.cc:1313-1378: rig = None -> one query per point (the rectified camera of `cam`);
        rig = SBP_RIG_DTYPE[1] -> queries[i * n_cams + camj]."""
        from .ba_types import PROJ_QUERY_DTYPE
        pts = np.ascontiguousarray(points)
        cam = np.ascontiguousarray(cam)
        if rig is None:
            q = np.zeros(len(pts), PROJ_QUERY_DTYPE)
            check(lib().vieo_sbp_project_last_frame(pts.ctypes.data, len(pts), cam.ctypes.data,
                                                    q.ctypes.data), "vieo_sbp_project_last_frame")
            return q
        rig = np.ascontiguousarray(rig)
        q = np.zeros(len(pts) * int(rig[0]["n_cams"]), PROJ_QUERY_DTYPE)
        check(lib().vieo_sbp_project_last_frame_rig(pts.ctypes.data, len(pts), cam.ctypes.data, rig.ctypes.data,
                                                    q.ctypes.data), "vieo_sbp_project_last_frame_rig")
        return q

    @staticmethod
    def project_keyframe(points, cam, rig, log_scale_factor):
        """Projection of SearchByProjection(Frame&, KeyFrame*, ...) (ORBmatcher.cc:1487-1543);
        points: KEYFRAME_POINT_DTYPE, rig may be None (the rectified camera of `cam`)."""
        from .ba_types import PROJ_QUERY_DTYPE
        pts = np.ascontiguousarray(points)
        cam = np.ascontiguousarray(cam)
        nc = 1 if rig is None else int(rig[0]["n_cams"])
        rig = None if rig is None else np.ascontiguousarray(rig)
        q = np.zeros(len(pts) * nc, PROJ_QUERY_DTYPE)
        check(lib().vieo_sbp_project_keyframe(pts.ctypes.data, len(pts), cam.ctypes.data,
                                              None if rig is None else rig.ctypes.data, float(log_scale_factor),
                                              q.ctypes.data), "vieo_sbp_project_keyframe")
        return q

    def _search(self, mode, queries, keys, uright, desc, taken, bounds, ratio=None, cam_first=None):
        """bounds: float32[4], or [n_cams][4] with cam_first int32[n_cams + 1] for the key list of a rig frame."""
        import ctypes
        queries = np.ascontiguousarray(queries)
        keys = np.ascontiguousarray(keys)
        uright = np.ascontiguousarray(uright, np.float32)
        desc = np.ascontiguousarray(desc, np.uint8)
        tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
        b = np.ascontiguousarray(bounds, np.float32)
        assign = np.zeros(max(len(keys), 1), np.int32)
        n = ctypes.c_int32()
        r = self.mfNNratio if ratio is None else float(ratio)
        if cam_first is None:
            check(lib().vieo_search_by_projection(mode, queries.ctypes.data, len(queries),
                                                  keys.ctypes.data, uright.ctypes.data, desc.ctypes.data,
                                                  None if tk is None else tk.ctypes.data, len(keys),
                                                  b.ctypes.data, r, int(self.mbCheckOrientation), assign.ctypes.data,
                                                  ctypes.byref(n)), "vieo_search_by_projection")
        else:
            cf = np.ascontiguousarray(cam_first, np.int32)
            check(lib().vieo_search_by_projection_rig(mode, queries.ctypes.data, len(queries), keys.ctypes.data,
                                                      uright.ctypes.data, desc.ctypes.data,
                                                      None if tk is None else tk.ctypes.data, len(keys),
                                                      cf.ctypes.data, b.ctypes.data, len(cf) - 1, r,
                                                      int(self.mbCheckOrientation), assign.ctypes.data,
                                                      ctypes.byref(n)), "vieo_search_by_projection_rig")
        return n.value, assign[:len(keys)]

    # ---- the same searches on a frame that is still resident in its extractor handle (include/vieo_hot.h "the resident
    # frame"): nothing of the frame goes up
    def search_last_frame_resident(self, ext, points, cam, uright=None):
        """SearchByProjection(Frame&, const Frame&, ...) as one call: projection + search.  returns (nmatches, assign)."""
        import ctypes
        n_keys = lib().vieo_orb_resident_keys(ext._h)
        assert n_keys >= 0, "the extractor holds no frame"
        pts, cam = np.ascontiguousarray(points), np.ascontiguousarray(cam)
        ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
        assign = np.zeros(max(n_keys, 1), np.int32)
        n = ctypes.c_int32()
        check(lib().vieo_search_by_projection_last_frame_resident(
            ext._h, pts.ctypes.data, len(pts), cam.ctypes.data, None if ur is None else ur.ctypes.data, self.mfNNratio,
            int(self.mbCheckOrientation), assign.ctypes.data, ctypes.byref(n)), "vieo_search_by_projection_last_frame_resident")
        return n.value, assign[:n_keys]

    def search_resident(self, mode, ext, queries, taken, bounds, uright=None, ratio=None):
        import ctypes
        n_keys = lib().vieo_orb_resident_keys(ext._h)
        assert n_keys >= 0, "the extractor holds no frame"
        q = np.ascontiguousarray(queries)
        tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
        ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
        b = np.ascontiguousarray(bounds, np.float32)
        assign = np.zeros(max(n_keys, 1), np.int32)
        n = ctypes.c_int32()
        check(lib().vieo_search_by_projection_resident(
            mode, ext._h, q.ctypes.data, len(q), None if ur is None else ur.ctypes.data, None if tk is None else tk.ctypes.data,
            b.ctypes.data, self.mfNNratio if ratio is None else float(ratio), int(self.mbCheckOrientation), assign.ctypes.data,
            ctypes.byref(n)), "vieo_search_by_projection_resident")
        return n.value, assign[:n_keys]

    def SearchByProjectionLastFrame(self, queries, keys, uright, desc, taken, bounds, cam_first=None):
        """SearchByProjection(Frame&, const Frame&, th, bMono, th_far) (ORBmatcher.cc:1303-1467)
        after project_last_frame(); returns (nmatches, assign[n_keys])."""
        return self._search(self.SBP_LAST_FRAME, queries, keys, uright, desc, taken, bounds, cam_first=cam_first)

    def SearchByProjectionKeyFrame(self, queries, keys, uright, desc, taken, bounds, ORBdist, cam_first=None):
        """SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist, th_far) (ORBmatcher.cc:1471-1606,
        relocalisation) on the key frame's projected map points; `taken` marks the keys that already hold
        a map point.  returns (nmatches, assign[n_keys])."""
        return self._search(self.SBP_RELOC, queries, keys, uright, desc, taken, bounds, ratio=ORBdist,
                            cam_first=cam_first)

    def SearchByProjectionLocalMap(self, queries, keys, uright, desc, taken, bounds, cam_first=None):
        """SearchByProjection(Frame&, vector<MapPoint*>&, th, th_far) (ORBmatcher.cc:230-335) on
        queries prepared by Frame::isInFrustum."""
        return self._search(self.SBP_LOCAL_MAP, queries, keys, uright, desc, taken, bounds, cam_first=cam_first)
