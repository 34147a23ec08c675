"""Host-side mirrors of the reference's stereo descriptor searches on top of the C-ABI:
  knn_match2(query, train)             cv::BFMatcher(NORM_HAMMING).knnMatch(k=2)  (Frame.cc:620-628)
  compute_stereo_matches(extL, extR..) Frame::ComputeStereoMatches                (Frame.cc:451-611)
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


class ORBmatcher:
    """ORBmatcher(nnratio=0.6, checkOri=True) (reference include/ORBmatcher.h:25-101), tracking-side
    searches on flattened inputs (ba_types.PROJ_QUERY_DTYPE etc.)."""
    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30
    SBP_LAST_FRAME, SBP_LOCAL_MAP, SBP_RELOC = 0, 1, 2

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)

    @staticmethod
    def project_last_frame(points, cam):
        from .ba_types import PROJ_QUERY_DTYPE
        pts = np.ascontiguousarray(points)
        cam = np.ascontiguousarray(cam)
        q = np.zeros(len(pts), PROJ_QUERY_DTYPE)
        check(lib().vieo_sbp_project_last_frame(pts.ctypes.data, len(pts), cam.ctypes.data,
                                                q.ctypes.data), "vieo_sbp_project_last_frame")
        return q

    def _search(self, mode, queries, keys, uright, desc, taken, bounds, ratio=None):
        import ctypes
        queries = np.ascontiguousarray(queries)
        keys = np.ascontiguousarray(keys)
        uright = np.ascontiguousarray(uright, np.float32)
        desc = np.ascontiguousarray(desc, np.uint8)
        tk = None if taken is None else np.ascontiguousarray(taken, np.uint8)
        b = np.ascontiguousarray(bounds, np.float32)
        assign = np.zeros(max(len(keys), 1), np.int32)
        n = ctypes.c_int32()
        check(lib().vieo_search_by_projection(mode, queries.ctypes.data, len(queries),
                                              keys.ctypes.data, uright.ctypes.data, desc.ctypes.data,
                                              None if tk is None else tk.ctypes.data, len(keys),
                                              b.ctypes.data, self.mfNNratio if ratio is None else float(ratio),
                                              int(self.mbCheckOrientation), assign.ctypes.data,
                                              ctypes.byref(n)), "vieo_search_by_projection")
        return n.value, assign[:len(keys)]

    def SearchByProjectionLastFrame(self, queries, keys, uright, desc, taken, bounds):
        """SearchByProjection(Frame&, const Frame&, th, bMono, th_far) (ORBmatcher.cc:1303-1467)
        after project_last_frame(); returns (nmatches, assign[n_keys])."""
        return self._search(self.SBP_LAST_FRAME, queries, keys, uright, desc, taken, bounds)

    def SearchByProjectionKeyFrame(self, queries, keys, uright, desc, taken, bounds, ORBdist):
        """SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist, th_far) (ORBmatcher.cc:1471-1606,
        relocalisation) on the key frame's projected map points; `taken` marks the keys that already hold
        a map point.  returns (nmatches, assign[n_keys])."""
        return self._search(self.SBP_RELOC, queries, keys, uright, desc, taken, bounds, ratio=ORBdist)

    def SearchByProjectionLocalMap(self, queries, keys, uright, desc, taken, bounds):
        """SearchByProjection(Frame&, vector<MapPoint*>&, th, th_far) (ORBmatcher.cc:230-335) on
        queries prepared by Frame::isInFrustum."""
        return self._search(self.SBP_LOCAL_MAP, queries, keys, uright, desc, taken, bounds)
