// TEST INFRASTRUCTURE -- part of the pinning recipe (oracle/ref_build/CMakeLists.txt, -DVIEO_REF_WITH_BA=ON).
// Single-edge hooks on the REFERENCE's own g2o types (src/Odom/g2otypes.h): error and Jacobians of
// EdgeReprojectPR / PRStereo / PRS / PRSStereo at given states, with the names and signatures of the restated oracle's
// hooks (oracle/local_ba_vio.cc: vo_lba_prs_edge_eval), so tests/test_global_ba_scale.py's Jacobian checks can be run
// against the reference's arithmetic.  The graph-level functions (Optimizer.cc) need the whole Frame / KeyFrame / Map
// object model and are compared end to end through the reference's own binaries instead.
// Not compiled in the authoring image (no Eigen / Sophus / OpenCV).
#include <cstring>
#include <vector>

#include "../../include/vieo_hot.h"
#include "common/camera_models/camera_pinhole.h"
#include "g2otypes.h"

namespace {
// BaseMultiEdge keeps _jacobianOplus protected (optimizer/g2o/g2o/core/base_multi_edge.h:101)
template <class E>
struct Open : E {
  using E::_jacobianOplus;
};

template <class Edge>
void eval_edge(Edge& e, g2o::VertexSBAPointXYZ& vX, g2o::VertexNavStatePR& vPR, g2o::VertexScale* vS, int de,
               double* err3, double* Jp, double* Jx, double* Js) {
  e.setVertex(0, &vX);
  e.setVertex(1, &vPR);
  if (vS) e.setVertex(2, vS);
  e.computeError();
  for (int r = 0; r < 3; r++) err3[r] = r < de ? e.error()[r] : 0.0;
  if (!Jp) return;
  e.linearizeOplus();
  std::memset(Jp, 0, 18 * sizeof(double)), std::memset(Jx, 0, 9 * sizeof(double)), std::memset(Js, 0, 3 * sizeof(double));
  for (int r = 0; r < de; r++) {
    for (int c = 0; c < 3; c++) Jx[r * 3 + c] = e._jacobianOplus[0](r, c);
    for (int c = 0; c < 6; c++) Jp[r * 6 + c] = e._jacobianOplus[1](r, c);
    if (vS) Js[r] = e._jacobianOplus[2](r, 0);
  }
}
}  // namespace

extern "C" void vo_lba_prs_edge_eval(const vieo_lba_vio_params* P, const vieo_navstate* ns, const double* Xh, double scale,
                                     const vieo_lba_obs* obs, double* err3, double* Jp, double* Jx, double* Js) {
  using namespace VIEO_SLAM;
  const vieo_lba_params& B = P->base;
  using Tdata = camm::Camera::Tdata;
  const std::vector<Tdata> intr = {(Tdata)B.fx, (Tdata)B.fy, (Tdata)B.cx, (Tdata)B.cy};
  camm::PinholeCamera cam(0, 752, 480, intr);
  Eigen::Matrix3d Rcb = Eigen::Map<const Eigen::Matrix<double, 3, 3, Eigen::RowMajor>>(B.Rcb);
  Eigen::Vector3d tcb(B.tcb[0], B.tcb[1], B.tcb[2]);
  NavStated s;
  s.mpwb = Eigen::Vector3d(ns->p[0], ns->p[1], ns->p[2]);
  s.mRwb = Sophus::SO3exd(Eigen::Quaterniond(ns->q[0], ns->q[1], ns->q[2], ns->q[3]));
  g2o::VertexSBAPointXYZ vX;
  vX.setEstimate(Eigen::Vector3d(Xh[0], Xh[1], Xh[2]));
  g2o::VertexNavStatePR vPR;
  vPR.setEstimate(s);
  g2o::VertexScale vS;
  vS.setEstimate(scale);
  const float bf = B.bf;
  if (obs->ur < 0) {
    Open<g2o::EdgeReprojectPRS> e;
    e.setMeasurement(Eigen::Vector2d(obs->u, obs->v));
    e.setInformation(Eigen::Matrix2d::Identity() * obs->inv_sigma2);
    e.SetParams(&cam, Rcb, tcb);
    eval_edge(e, vX, vPR, &vS, 2, err3, Jp, Jx, Js);
  } else {
    Open<g2o::EdgeReprojectPRSStereo> e;
    e.setMeasurement(Eigen::Vector3d(obs->u, obs->v, obs->ur));
    e.setInformation(Eigen::Matrix3d::Identity() * obs->inv_sigma2);
    e.SetParams(&cam, Rcb, tcb, &bf);
    eval_edge(e, vX, vPR, &vS, 3, err3, Jp, Jx, Js);
  }
}
