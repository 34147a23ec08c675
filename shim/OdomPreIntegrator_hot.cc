// OdomPreIntegrator_hot.cc -- IMU pre-integration of one interval behind the reference's own member,
//
//   template <class IMUDataBase>
//   int IMUPreIntegratorBase<IMUDataBase>::PreIntegration(const double& timeStampi, const double& timeStampj, const Vector3d& bgi_bar,
//                                                         const Vector3d& bai_bar, const listeig(IMUDataBase)::const_iterator& iterBegin,
//                                                         const listeig(IMUDataBase)::const_iterator& iterEnd, bool breset = true)
//                                                                                       src/Odom/OdomPreIntegrator.h:226-328 (+ update(), :330-506)
//
// as an explicit specialisation for IMUDataBase (the only instantiation of the reference: OdomPreIntegrator.h:550), on
// vieo_imu_preintegrate_batch.  The member is a template in a header: INTEGRATION.md section 4 puts `#ifndef VIEO_HOT` around
// its body in the header (the declaration inside the class stays) and compiles this file.  Its callers -- FrameBase::
// PreIntegration<...> (src/FrameBase.cpp:69-71: Tracking::PreIntegration, KeyFrame's constructor) and the two-argument
// overloads -- are untouched.
//
// The call hands over the whole list [iterBegin, iterEnd); the selection of the samples around [timeStampi, timeStampj],
// the interpolation of the two partial intervals, the backward order of map reuse, the 1.5 s gap check and the mid-point
// update are the kernel's (imu_preint.hip; bit-compared with the oracle's restatement of the lines above).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "Frame.h"  // (pulls in src/Odom/OdomPreIntegrator.h in the reference tree)
#include "vieo_hot.h"

namespace VIEO_SLAM {

template <>
int IMUPreIntegratorBase<IMUDataBase>::PreIntegration(const double& timeStampi, const double& timeStampj, const Eigen::Vector3d& bgi_bar,
                                                      const Eigen::Vector3d& bai_bar,
                                                      const typename listeig(IMUDataBase)::const_iterator& iterBegin,
                                                      const typename listeig(IMUDataBase)::const_iterator& iterEnd, bool breset) {
  if (iterBegin == iterEnd) return 0;  // :232: nothing happens, the members keep their values
  if (!breset) {
    // (continuing an integration: no caller in the reference passes false -- FrameBase.cpp:69-71 forwards its default true)
    std::fprintf(stderr, "vieo_hot: IMUPreIntegratorBase::PreIntegration(breset = false) is not built\n");
    std::abort();
  }
  std::vector<vieo_imu_sample> samples;
  for (auto it = iterBegin; it != iterEnd; ++it) {
    vieo_imu_sample s;
    s.t = it->mtm;
    for (int r = 0; r < 3; ++r) s.w[r] = it->mw(r), s.a[r] = it->ma(r);
    samples.push_back(s);
  }
  vieo_imu_noise noise;
  std::memset(&noise, 0, sizeof(noise));
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) noise.sigma_g[r * 3 + c] = IMUDataBase::mSigmag(r, c), noise.sigma_a[r * 3 + c] = IMUDataBase::mSigmaa(r, c);
  noise.freq_ref = IMUDataBase::mFreqRef, noise.dt_cov_noise_fixed = IMUDataBase::mdt_cov_noise_fixed;
  const int32_t first[2] = {0, (int32_t)samples.size()};
  const double bg[3] = {bgi_bar(0), bgi_bar(1), bgi_bar(2)}, ba[3] = {bai_bar(0), bai_bar(1), bai_bar(2)};
  vieo_imu_preint out;
  double sigma_prv[81];
  int32_t status = 0;
  const int rc = vieo_imu_preintegrate_batch(&noise, samples.data(), first, &timeStampi, &timeStampj, bg, ba, 1, &out, sigma_prv, &status);
  if (rc != VIEO_OK) {
    std::fprintf(stderr, "vieo_hot: vieo_imu_preintegrate_batch failed (%d): %s\n", rc, vieo_last_error());
    std::abort();
  }
  // reset() + the accumulated members (:234, update())
  this->mdeltatij = out.dt;
  for (int r = 0; r < 3; ++r) {
    mvij(r) = out.vij[r], mpij(r) = out.pij[r];
    for (int c = 0; c < 3; ++c) {
      mRij(r, c) = out.Rij[r * 3 + c];
      mJgpij(r, c) = out.Jgp[r * 3 + c], mJapij(r, c) = out.Jap[r * 3 + c];
      mJgvij(r, c) = out.Jgv[r * 3 + c], mJavij(r, c) = out.Jav[r * 3 + c], mJgRij(r, c) = out.JgR[r * 3 + c];
    }
  }
  for (int r = 0; r < 9; ++r)
    for (int c = 0; c < 9; ++c) mSigmaij(r, c) = out.Sigma[r * 9 + c], mSigmaijPRV(r, c) = sigma_prv[r * 9 + c];
  if (status == VIEO_PREINT_GAP) {  // :290-294 "CheckIMU!!!": mdeltatij = 0, -1
    this->mdeltatij = 0;
    return -1;
  }
  return 0;
}

}  // namespace VIEO_SLAM
