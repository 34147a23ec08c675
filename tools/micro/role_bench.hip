// One lane's cost of the single-lane edge evaluations of k_pose_opt_vio (s_memtime around each, 64 repetitions):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I vieo_slam_amd/csrc tools/micro/role_bench.hip -o tools/micro/role_bench
// The kernel's own functions are used (pose_opt_vio.hip is included as a translation unit).
#include "../../vieo_slam_amd/csrc/pose_opt_vio.hip"
#include <cstdio>
namespace vieo {
// (host-side entry points of the included translation unit: not used here)
void set_error(const char*, ...) {}
int require_device() { return 0; }
int pose_rig_launches() { return 1; }
int pose_enc_launches() { return 1; }
int pose_launch_mask(int) { return 1; }
struct RB {
  NSd si, sj, pr;
  vieo_imu_preint imu;
  double gw[4], qRij[4], errI[9], errP[15], JI[216], JP[225];
  RotCache rc;
};
__global__ void __launch_bounds__(256) k_role(const vieo_imu_preint* M, unsigned long long* t, double* out, int reps) {
  __shared__ RB S;
  const int tid = threadIdx.x;
  for (int e = tid; e < (int)(sizeof(vieo_imu_preint) / 8); e += 256) ((double*)&S.imu)[e] = ((const double*)M)[e];
  if (tid == 0) {
    S.si = NSd{{0.1, 0.2, 0.3}, {0.5, 0.1, -0.2}, 0.999, 0.02, -0.03, 0.01, {1e-3, 2e-3, -1e-3}, {0.01, 0.02, 0.03}, {1e-4, -2e-4, 1e-4}, {1e-3, 1e-3, -1e-3}};
    S.sj = NSd{{0.13, 0.21, 0.29}, {0.52, 0.08, -0.21}, 0.9985, 0.025, -0.035, 0.012, {1e-3, 2e-3, -1e-3}, {0.01, 0.02, 0.03}, {0, 0, 0}, {0, 0, 0}};
    S.pr = S.si;
    S.pr.qx += 1e-3;
    const double n = 1.0 / sqrt(S.si.qw * S.si.qw + S.si.qx * S.si.qx + S.si.qy * S.si.qy + S.si.qz * S.si.qz);
    S.si.qw *= n, S.si.qx *= n, S.si.qy *= n, S.si.qz *= n;
    const double m = 1.0 / sqrt(S.sj.qw * S.sj.qw + S.sj.qx * S.sj.qx + S.sj.qy * S.sj.qy + S.sj.qz * S.sj.qz);
    S.sj.qw *= m, S.sj.qx *= m, S.sj.qy *= m, S.sj.qz *= m;
    const double k = 1.0 / sqrt(S.pr.qw * S.pr.qw + S.pr.qx * S.pr.qx + S.pr.qy * S.pr.qy + S.pr.qz * S.pr.qz);
    S.pr.qw *= k, S.pr.qx *= k, S.pr.qy *= k, S.pr.qz *= k;
    S.gw[0] = 0, S.gw[1] = 0, S.gw[2] = -9.81;
    const Qd q = R_to_q(S.imu.Rij);
    S.qRij[0] = q.w, S.qRij[1] = q.x, S.qRij[2] = q.y, S.qRij[3] = q.z;
  }
  prior_jacobian_init(S.JP, tid, 256);
  imu_jacobian_init(*M, S.JI, tid, 256);
  __syncthreads();
  if (tid != 0) return;
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < reps; r++) {
    unsigned long long a = __builtin_amdgcn_s_memtime();
    imu_error(S.imu, S.gw, S.si, S.sj, S.errI, 6, 1);
    unsigned long long b = __builtin_amdgcn_s_memtime();
    acc[0] += b - a;
    imu_error_rot(S.imu, Qd{S.qRij[0], S.qRij[1], S.qRij[2], S.qRij[3]}, S.si, S.sj, S.errI, S.rc);
    a = __builtin_amdgcn_s_memtime();
    acc[1] += a - b;
    {
      double e9[9];
      RotCache C = S.rc;
      imu_error(S.imu, S.gw, S.si, S.sj, e9, 6, 1);
      imu_error_rot(S.imu, Qd{S.qRij[0], S.qRij[1], S.qRij[2], S.qRij[3]}, S.si, S.sj, e9, C);
      for (int k = 0; k < 9; k++) S.errI[k] = e9[k];
      S.rc = C;
    }
    b = __builtin_amdgcn_s_memtime();
    acc[2] += b - a;
    prior_error(S.pr, S.si, S.errP, S.rc);
    a = __builtin_amdgcn_s_memtime();
    acc[3] += a - b;
    imu_linearize_pv(S.imu, S.gw, S.si, S.sj, S.JI);
    imu_linearize_rot(S.imu, S.errI, S.JI, S.rc);
    b = __builtin_amdgcn_s_memtime();
    acc[4] += b - a;
    prior_linearize(S.errP, S.JP, S.rc);
    a = __builtin_amdgcn_s_memtime();
    acc[5] += a - b;
    double x[15] = {1e-3, -2e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-4, 2e-4, -1e-4, 0, 0, 0, 0, 0, 0};
    ns_inc_unit(S.sj, x, x + 9);
    b = __builtin_amdgcn_s_memtime();
    acc[6] += b - a;
    a = __builtin_amdgcn_s_memtime();
    acc[7] += a - b;
  }
  for (int i = 0; i < 8; i++) t[i] = acc[i];
  out[0] = S.errI[0] + S.errI[7] + S.errP[7] + S.JI[100] + S.JP[100] + S.sj.qw;
}
}  // namespace vieo
int main() {
  vieo_imu_preint M;
  memset(&M, 0, sizeof(M));
  M.dt = 0.05;
  for (int i = 0; i < 3; i++) M.Rij[i * 4] = 1.0;
  M.Rij[1] = -0.01, M.Rij[3] = 0.01;
  for (int i = 0; i < 9; i++) M.JgR[i] = 0.01 * (i + 1), M.Jgp[i] = 1e-3 * i, M.Jap[i] = 1e-3 * (9 - i), M.Jgv[i] = 2e-3 * i, M.Jav[i] = 1e-2;
  M.pij[0] = 0.03, M.vij[0] = 0.02;
  vieo_imu_preint* dM;
  unsigned long long* dt;
  double* dout;
  hipMalloc(&dM, sizeof(M)), hipMalloc(&dt, 64), hipMalloc(&dout, 8);
  hipMemcpy(dM, &M, sizeof(M), hipMemcpyHostToDevice);
  const int reps = 64;
  for (int pass = 0; pass < 2; pass++) hipLaunchKernelGGL(vieo::k_role, dim3(1), dim3(256), 0, 0, dM, dt, dout, reps);
  hipDeviceSynchronize();
  unsigned long long h[8];
  double o;
  hipMemcpy(h, dt, 64, hipMemcpyDeviceToHost), hipMemcpy(&o, dout, 8, hipMemcpyDeviceToHost);
  const char* names[8] = {"imu_error p, v rows", "imu_error_rot", "imu_error both (one block)", "prior_error", "imu_linearize pv + rot", "prior_linearize", "ns_inc_unit", "empty (two s_memtime)"};
  for (int i = 0; i < 8; i++) printf("%-32s %8.0f cycles\n", names[i], (double)h[i] / reps);
  printf("(checksum %g)\n", o);
  return 0;
}
