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
