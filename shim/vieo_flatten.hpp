// vieo_flatten.hpp -- reference objects <-> the POD structs of include/vieo_hot.h.
//
// Included by the two shim translation units (ORBmatcher_hot.cc, Optimizer_hot.cc), which are compiled INSIDE the
// reference tree against the reference's own headers (FrameBase.h, MapPoint.h, NavState.h, camera_base.h ...).  Only
// element access / trivial constructors of the third-party types are used (Eigen: operator(), Quaterniond(w,x,y,z),
// .w()..; Sophus: rotationMatrix(), translation(), unit_quaternion(); cv::Mat: at<float>, ptr, data, rows), so the
// header does not depend on a particular Eigen / Sophus / OpenCV version.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

#include "vieo_hot.h"

namespace vieo_shim {

// NavState (src/Odom/NavState.h:18-85) -> vieo_navstate
inline void to_pod(const VIEO_SLAM::NavState& ns, vieo_navstate& o) {
  const auto& q = ns.mRwb.unit_quaternion();
  o.q[0] = q.w(), o.q[1] = q.x(), o.q[2] = q.y(), o.q[3] = q.z();
  for (int i = 0; i < 3; ++i) {
    o.p[i] = ns.mpwb(i), o.v[i] = ns.mvwb(i);
    o.bg[i] = ns.mbg(i), o.ba[i] = ns.mba(i), o.dbg[i] = ns.mdbg(i), o.dba[i] = ns.mdba(i);
  }
}
// what the optimisers write back: p, R, v and the bias corrections (bg, ba stay as they are)
inline void from_pod(const vieo_navstate& o, VIEO_SLAM::NavState& ns) {
  ns.mRwb = Sophus::SO3exd(Eigen::Quaterniond(o.q[0], o.q[1], o.q[2], o.q[3]));
  for (int i = 0; i < 3; ++i) ns.mpwb(i) = o.p[i], ns.mvwb(i) = o.v[i], ns.mdbg(i) = o.dbg[i], ns.mdba(i) = o.dba[i];
}

// Sophus::SE3<T> -> row-major 3x4 doubles
template <class SE3>
inline void se3_to_3x4(const SE3& T, double* o) {
  const auto R = T.rotationMatrix();
  const auto t = T.translation();
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) o[r * 4 + c] = (double)R(r, c);
    o[r * 4 + 3] = (double)t(r);
  }
}

// camm::Camera (common/camera_models/camera_{pinhole,radtan,kb8}.h) -> vieo_camera; Rcrb / tcrb = the frame's
// body -> reference-camera extrinsics (FrameBase::meigRcb / meigtcb), composed with the camera's Tcr exactly as
// EdgeReproject::SetParams does (src/Odom/g2otypes.h:409-416).  false: unknown model / parameter count.
inline bool to_pod(const VIEO_SLAM::camm::Camera* cam, const Eigen::Matrix3d& Rcrb, const Eigen::Vector3d& tcrb,
                   vieo_camera& o) {
  std::memset(&o, 0, sizeof(o));
  const std::vector<float>& p = cam->GetParameters();
  if (p.size() < 4) return false;
  o.fx = p[0], o.fy = p[1], o.cx = p[2], o.cy = p[3];
  switch (cam->camera_model()) {
    case VIEO_SLAM::camm::Camera::kPinhole:
      o.model = VIEO_CAM_PINHOLE;
      break;
    case VIEO_SLAM::camm::Camera::kRadtan:
      o.model = VIEO_CAM_RADTAN;
      if (p.size() < 8 || p.size() > 12) return false;
      o.num_k = (int)p.size() - 6;
      for (size_t i = 4; i < p.size(); ++i) o.dist[i - 4] = p[i];
      break;
    case VIEO_SLAM::camm::Camera::kKB8:
      o.model = VIEO_CAM_KB8;
      if (p.size() < 8) return false;
      for (int i = 0; i < 4; ++i) o.dist[i] = p[4 + i];
      break;
    default:
      return false;
  }
  const auto Rccr = cam->GetTcr().rotationMatrix();  // float
  const auto tcr = cam->GetTcr().translation();
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += (double)Rccr(r, k) * Rcrb(k, c);
      o.Rcb[r * 3 + c] = s;
    }
    double s = 0;
    for (int k = 0; k < 3; ++k) s += (double)Rccr(r, k) * tcrb(k);
    o.tcb[r] = s + (double)tcr(r);
  }
  return true;
}

// IMUPreIntegratorBase (src/Odom/OdomPreIntegrator.h:108-147) -> vieo_imu_preint; prv: Sigma = mSigmaijPRV (p, Phi, v)
// for the local / full BA, else mSigmaij (p, v, Phi) for the pose optimisation
template <class Preint>
inline void to_pod(const Preint& m, bool prv, vieo_imu_preint& o) {
  o.dt = m.mdeltatij;
  for (int r = 0; r < 3; ++r) {
    o.vij[r] = m.mvij(r), o.pij[r] = m.mpij(r);
    for (int c = 0; c < 3; ++c) {
      o.Rij[r * 3 + c] = m.mRij(r, c);
      o.JgR[r * 3 + c] = m.mJgRij(r, c), o.Jgv[r * 3 + c] = m.mJgvij(r, c), o.Jav[r * 3 + c] = m.mJavij(r, c);
      o.Jgp[r * 3 + c] = m.mJgpij(r, c), o.Jap[r * 3 + c] = m.mJapij(r, c);
    }
  }
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 9; ++c) o.Sigma[r * 9 + c] = prv ? m.mSigmaijPRV(r, c) : m.mSigmaij(r, c);
}
// EncPreIntegrator (OdomPreIntegrator.h:66-100) -> vieo_enc_preint
template <class EncPreint>
inline void to_pod(const EncPreint& m, vieo_enc_preint& o) {
  o.dt = m.mdeltatij;
  for (int r = 0; r < 6; ++r) {
    o.delx[r] = m.mdelxEij(r);
    for (int c = 0; c < 6; ++c) o.Sigma[r * 6 + c] = m.mSigmaEij(r, c);
  }
}

// Tbe = Frame::mTbc * Frame::mTce (CV_32F 4x4) as quaternion (w, x, y, z) + translation, the way the reference forms
// qRbe = Quaterniond(Converter::toMatrix3d(Tbe.R)) (Optimizer.cc:226-228)
inline void tbe_to_pod(const cv::Mat& Tbc, const cv::Mat& Tce, double* qRbe, double* pbe) {
  float T[12];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      float s = 0;
      for (int k = 0; k < 4; ++k) s += Tbc.at<float>(r, k) * Tce.at<float>(k, c);
      T[r * 4 + c] = s;
    }
  Eigen::Matrix3d R;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R(r, c) = T[r * 4 + c];
  const Eigen::Quaterniond q(R);
  qRbe[0] = q.w(), qRbe[1] = q.x(), qRbe[2] = q.y(), qRbe[3] = q.z();
  for (int r = 0; r < 3; ++r) pbe[r] = T[r * 4 + 3];
}

// protected static FrameBase::gridinfo_ (image bounds per camera) read through a derived class
struct GridAccess : public VIEO_SLAM::FrameBase {
  static const std::vector<std::array<float, 4>>& bounds() { return gridinfo_.minmax_xy_; }
};

// the cameras of a frame as vieo_sbp_rig (ORBmatcher.cc:1339-1366); false: more than 4 cameras / unknown model
template <class FrameT>
inline bool rig_to_pod(const FrameT& F, vieo_sbp_rig& R) {
  std::memset(&R, 0, sizeof(R));
  const size_t nc = F.mpCameras.size();
  if (nc < 1 || nc > 4 || GridAccess::bounds().size() < nc) return false;
  R.n_cams = (int)nc, R.use_distort = FrameT::usedistort_ ? 1 : 0;
  const Eigen::Matrix3d I = Eigen::Matrix3d::Identity();
  const Eigen::Vector3d z = Eigen::Vector3d::Zero();
  for (size_t c = 0; c < nc; ++c) {
    if (!to_pod(F.mpCameras[c].get(), I, z, R.cams[c])) return false;
    se3_to_3x4(F.mpCameras[c]->GetTcr(), R.Tcr[c]);
    const auto t = F.mpCameras[c]->GetTrc().translation();
    for (int k = 0; k < 3; ++k) R.trc[c][k] = (double)t(k);
    for (int k = 0; k < 4; ++k) R.bounds[c][k] = GridAccess::bounds()[c][k];
  }
  return true;
}

// key ranges of the cameras in mvKeys (camera-major, Frame.cc:738-764): cam_first[c] = first key of camera c
template <class FrameT>
inline void cam_first_of(const FrameT& F, int n_cams, int32_t* cam_first) {
  int count[4] = {0, 0, 0, 0};
  if (F.mapn2in_.empty())
    count[0] = F.N;
  else
    for (int i = 0; i < F.N; ++i) ++count[(int)std::get<0>(F.mapn2in_[i]) & 3];
  cam_first[0] = 0;
  for (int c = 0; c < n_cams; ++c) cam_first[c + 1] = cam_first[c] + count[c];
}

}  // namespace vieo_shim
