// ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.  camm::{Pinhole,Radtan,KB8}Camera::Project restated
// (common/camera_models/camera_pinhole.h:70-106, camera_radtan.h:61-129, camera_kb8.h:68-157):
// float parameters (Tdata), double arithmetic (Tcalc), image point returned as float (Vec2data).
#pragma once
#include <cmath>

#include "../include/vieo_hot.h"

namespace vo {

struct OCam {  // one camera as an EdgeReproject sees it
  int model = 0, num_k = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0, bf = 0;
  float dist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double Rcb[9], tcb[3];
};

static inline void ocam_pinhole(const OCam& c, const double* P, float* uv, double* J) {
  const double x = P[0], y = P[1], z = P[2];
  const double invz = 1. / z;
  if (uv) {
    uv[0] = (float)((double)c.fx * x * invz + c.cx);
    uv[1] = (float)((double)c.fy * y * invz + c.cy);
  }
  if (J) {
    const double invz2 = invz * invz;
    J[0] = c.fx * invz, J[1] = 0, J[2] = -c.fx * x * invz2;
    J[3] = 0, J[4] = c.fy * invz, J[5] = -c.fy * y * invz2;
  }
}

// uv (2 floats) and/or J (2x3 row-major), either may be null
static inline void ocam_project(const OCam& c, const double* P, float* uv, double* J) {
  if (c.model == VIEO_CAM_RADTAN) {
    const float* k = c.dist;
    const float* p = k + c.num_k;
    const double invz = 1 / P[2];
    double x = P[0] * invz, y = P[1] * invz;
    const double x2 = x * x, y2 = y * y, xy = x * y, r2 = x2 + y2;
    double fd = 1, term_r = 1;
    for (int i = 0; i < c.num_k; ++i) {
      term_r *= r2;
      fd += k[i] * term_r;
    }
    if (J) {
      double fd2 = 0, coeff2 = 0;
      term_r = 1;
      for (int i = 2; i < c.num_k; ++i) {
        coeff2 += 2;
        fd2 += coeff2 * k[i] * term_r;
        term_r *= r2;
      }
      const double du_dx = c.fx * invz * (fd + fd2 * x2 + 2 * (p[0] * y + 3 * p[1] * x));
      const double du_dy = c.fx * invz * (fd2 * xy + 2 * (p[0] * x + p[1] * y));
      const double du_dz = -(x * du_dx + y * du_dy);
      const double dv_dx = du_dy * c.fy / c.fx;
      const double dv_dy = c.fy * invz * (fd + fd2 * y2 + 2 * (p[1] * x + 3 * p[0] * y));
      const double dv_dz = -(x * dv_dx + y * dv_dy);
      J[0] = du_dx, J[1] = du_dy, J[2] = du_dz, J[3] = dv_dx, J[4] = dv_dy, J[5] = dv_dz;
    }
    const double xd = x * fd + 2 * p[0] * xy + p[1] * (r2 + 2 * x2);
    const double yd = y * fd + 2 * p[1] * xy + p[0] * (r2 + 2 * y2);
    if (uv) {
      const double Pn[3] = {xd, yd, 1.};
      ocam_pinhole(c, Pn, uv, nullptr);
    }
    return;
  }
  if (c.model == VIEO_CAM_KB8) {
    const double x = P[0], y = P[1];
    const double x2 = x * x, y2 = y * y, r2 = x2 + y2, r = std::sqrt(r2);
    const float precision_r = 1e-5;
    if (r > precision_r) {
      const float fx = c.fx, fy = c.fy, k1 = c.dist[0], k2 = c.dist[1], k3 = c.dist[2], k4 = c.dist[3];
      const double z = P[2];
      const double theta = std::atan2(r, z), theta2 = theta * theta;
      double thetad = k4 * theta2;
      thetad += k3, thetad *= theta2, thetad += k2, thetad *= theta2, thetad += k1, thetad *= theta2;
      thetad += 1, thetad *= theta;
      const double mx = x * thetad / r, my = y * thetad / r;
      if (uv) {
        const double Pn[3] = {mx, my, 1.};
        ocam_pinhole(c, Pn, uv, nullptr);
      }
      if (J) {
        const double invr = 1. / r, d_r_d_x = x * invr, d_r_d_y = y * invr;
        const double tmp = 1. / (z * z + r2);
        const double d_thetad_x = d_r_d_x * z * tmp, d_thetad_y = d_r_d_y * z * tmp;
        double dd = double(9) * k4 * theta2;
        dd += double(7) * k3, dd *= theta2, dd += double(5) * k2, dd *= theta2, dd += double(3) * k1;
        dd *= theta2, dd += double(1);
        const double invr2 = invr * invr;
        J[0] = fx * (x * r * dd * d_thetad_x + y2 * thetad / r) * invr2;
        J[1] = fx * x * (dd * d_thetad_y * r - y * thetad / r) * invr2;
        J[2] = -fx * x * dd * tmp;
        J[3] = J[1] * fy / fx;
        J[4] = fy * (y * r * dd * d_thetad_y + x2 * thetad / r) * invr2;
        J[5] = -fy * y * dd * tmp;
      }
      return;
    }
  }
  ocam_pinhole(c, P, uv, J);
}

// cameras of a window: n_cams == 0 -> the single pinhole of vieo_lba_params
static inline int ocams_from_params(const vieo_lba_params& P, OCam* out) {
  if (P.n_cams == 0) {
    OCam& c = out[0];
    c.model = VIEO_CAM_PINHOLE, c.fx = P.fx, c.fy = P.fy, c.cx = P.cx, c.cy = P.cy, c.bf = P.bf;
    for (int i = 0; i < 9; i++) c.Rcb[i] = P.Rcb[i];
    for (int i = 0; i < 3; i++) c.tcb[i] = P.tcb[i];
    return 1;
  }
  for (int k = 0; k < P.n_cams; k++) {
    OCam& c = out[k];
    const vieo_camera& s = P.cams[k];
    c.model = s.model, c.num_k = s.model == VIEO_CAM_RADTAN ? s.num_k : 0;
    c.fx = s.fx, c.fy = s.fy, c.cx = s.cx, c.cy = s.cy, c.bf = 0;
    for (int i = 0; i < 8; i++) c.dist[i] = s.dist[i];
    for (int i = 0; i < 9; i++) c.Rcb[i] = s.Rcb[i];
    for (int i = 0; i < 3; i++) c.tcb[i] = s.tcb[i];
  }
  return P.n_cams;
}

// Base (pinhole) UnProject to the plane z = 1
static inline void unproject_pinhole(const OCam& c, const float* uv, double* P) {
  P[0] = ((double)uv[0] - c.cx) / c.fx;
  P[1] = ((double)uv[1] - c.cy) / c.fy;
  P[2] = 1.0;
}

// {Pinhole,Radtan,KB8}Camera::UnProject, kUnProject2Plane (camera_pinhole.h:108-125, camera_radtan.h:132-178,
// camera_kb8.h:159-195 + SolveTheta :278-312; num_max_iteration_ = 10, precision_ = 1e-8f: camera_base.h:121-123)
static inline void ocam_unproject(const OCam& c, const float* uv, double* P) {
  const int max_iter = 10;
  const float precision = 1e-8f;
  if (c.model == VIEO_CAM_RADTAN) {
    double t[3];
    unproject_pinhole(c, uv, t);
    const double y0 = t[0], y1 = t[1];
    double yb0 = y0, yb1 = y1;
    const double precision2 = precision * precision;
    for (int i = 0; i < max_iter; ++i) {
      const double Pn[3] = {yb0, yb1, 1.};
      float y2f[2];
      double Jc[6];
      ocam_project(c, Pn, y2f, Jc);
      unproject_pinhole(c, y2f, t);
      const double F00 = Jc[0] / c.fx, F01 = Jc[1] / c.fx, F10 = F01, F11 = Jc[4] / c.fy;
      const double e0 = y0 - t[0], e1 = y1 - t[1];
      // du = (F^T F)^-1 F^T e, 2x2 closed-form inverse
      const double A00 = F00 * F00 + F10 * F10, A01 = F00 * F01 + F10 * F11, A11 = F01 * F01 + F11 * F11;
      const double det = A00 * A11 - A01 * A01, inv = 1. / det;
      const double I00 = A11 * inv, I01 = -A01 * inv, I11 = A00 * inv;
      // (A^-1 F^T) e, evaluated as Eigen does: ((A^-1 * F^T) * e)
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
    double t[3];
    unproject_pinhole(c, uv, t);
    const double mx = t[0], my = t[1];
    double theta = 0, sin_theta = 0, cos_theta = 1, scaling = 1.0;
    double thetad = std::sqrt(mx * mx + my * my);
    thetad = std::min(std::max(-M_PI / 2., thetad), M_PI / 2.);
    if (thetad > precision) {
      const float k1 = c.dist[0], k2 = c.dist[1], k3 = c.dist[2], k4 = c.dist[3];
      theta = thetad;
      for (int i = 0; i < max_iter; ++i) {  // SolveTheta
        const double theta2 = theta * theta;
        double func = k4 * theta2;
        func += k3, func *= theta2, func += k2, func *= theta2, func += k1, func *= theta2, func += 1, func *= theta;
        double d = 9 * k4 * theta2;
        d += 7 * k3, d *= theta2, d += 5 * k2, d *= theta2, d += 3 * k1, d *= theta2, d += 1;
        const double fix = (thetad - func) / d;
        theta += fix;
        if (std::fabs(fix) < precision) break;
      }
      sin_theta = std::tan(theta);
      cos_theta = 1.;
      scaling = sin_theta / thetad;
    }
    P[0] = mx * scaling, P[1] = my * scaling, P[2] = cos_theta;
    return;
  }
  unproject_pinhole(c, uv, P);
}

}  // namespace vo
