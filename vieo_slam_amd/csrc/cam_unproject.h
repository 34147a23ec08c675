// {Pinhole,Radtan,KB8}Camera::UnProject on the device, shared by the fisheye stereo matcher and SearchForTriangulation.
#pragma once
#include "ba_device.h"

namespace vieo {

// camm::PinholeCamera::UnProject to the plane z = 1 (camera_pinhole.h:108-125)
__device__ __forceinline__ void unproject_pinhole(const CamD& c, double u, double v, double* P) {
  P[0] = (u - c.cx) / c.fx;
  P[1] = (v - c.cy) / c.fy;
  P[2] = 1.0;
}

// {Pinhole,Radtan,KB8}Camera::UnProject, kUnProject2Plane, num_max_iteration_ = 10, precision_ = 1e-8f
// (camera_base.h:121-123, camera_radtan.h:132-178, camera_kb8.h:159-195 + SolveTheta :278-312)
static __device__ void cam_unproject(const CamD& c, float u, float v, double* P) {
  const double precision = (double)1e-8f;
  double t[3];
  unproject_pinhole(c, (double)u, (double)v, t);
  if (c.model == VIEO_CAM_RADTAN) {
    const double y0 = t[0], y1 = t[1];
    double yb0 = y0, yb1 = y1;
    const double precision2 = precision * precision;
    for (int i = 0; i < 10; ++i) {
      const double Pn[3] = {yb0, yb1, 1.};
      double uv[2], Jc[6];
      cam_project(c, Pn, uv, Jc);  // uv already rounded to float
      unproject_pinhole(c, uv[0], uv[1], t);
      const double F00 = Jc[0] / c.fx, F01 = Jc[1] / c.fx, F10 = F01, F11 = Jc[4] / c.fy;
      const double e0 = y0 - t[0], e1 = y1 - t[1];
      const double A00 = F00 * F00 + F10 * F10, A01 = F00 * F01 + F10 * F11, A11 = F01 * F01 + F11 * F11;
      const double det = A00 * A11 - A01 * A01, inv = 1. / det;
      const double I00 = A11 * inv, I01 = -A01 * inv, I11 = A00 * inv;
      const double M00 = I00 * F00 + I01 * F01, M01 = I00 * F10 + I01 * F11;
      const double M10 = I01 * F00 + I11 * F01, M11 = I01 * F10 + I11 * F11;
      yb0 += M00 * e0 + M01 * e1;
      yb1 += M10 * e0 + M11 * e1;
      if (e0 * e0 + e1 * e1 < precision2) break;
    }
    P[0] = (double)(float)yb0, P[1] = (double)(float)yb1, P[2] = 1.0;
    return;
  }
  if (c.model == VIEO_CAM_KB8) {
    const double mx = t[0], my = t[1];
    double scaling = 1.0;
    double thetad = sqrt(mx * mx + my * my);
    thetad = fmin(fmax(-M_PI / 2., thetad), M_PI / 2.);
    if (thetad > precision) {
      double theta = thetad;
      for (int i = 0; i < 10; ++i) {
        const double theta2 = theta * theta;
        double func = c.k[3] * theta2;
        func += c.k[2], func *= theta2, func += c.k[1], func *= theta2, func += c.k[0], func *= theta2;
        func += 1, func *= theta;
        double d = 9 * c.k[3] * theta2;
        d += 7 * c.k[2], d *= theta2, d += 5 * c.k[1], d *= theta2, d += 3 * c.k[0], d *= theta2, d += 1;
        const double fix = (thetad - func) / d;
        theta += fix;
        if (fabs(fix) < precision) break;
      }
      scaling = tan(theta) / thetad;
    }
    P[0] = mx * scaling, P[1] = my * scaling, P[2] = 1.0;
    return;
  }
  P[0] = t[0], P[1] = t[1], P[2] = t[2];
}

}  // namespace vieo
