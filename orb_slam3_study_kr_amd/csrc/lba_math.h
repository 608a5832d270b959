// lba_math.h -- device-side SE3 / projection / robust-kernel arithmetic (FP64, gfx950).
//
// Numerical contract (what the CPU reference path computes; paths relative to
// /root/reference, "g2o/" = Thirdparty/g2o/g2o/):
//   * poses are unit quaternion (x,y,z,w, w>=0) + translation, renormalised after
//     every product                      g2o/types/se3quat.h:104-110,280-285
//   * point transform q*X+t with Eigen's two-cross-product form   se3quat.h:217-221
//   * stereo residual uses a FLOAT 1/z and FLOAT bf                g2o/types/types_six_dof_expmap.cpp:190-197
//   * Jacobians in double                                         types_six_dof_expmap.cpp:228-273,
//                                                                  src/OptimizableTypes.cpp:139-160
//   * exp() small-angle branch R = I + W + W*W (sic)               se3quat.h:238-242
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/orbslam3_hip.h"

namespace osh {
namespace dev {

__device__ __forceinline__ void quat_rotate(const double* q, const double* v, double* o) {
  double uv0 = q[1] * v[2] - q[2] * v[1];
  double uv1 = q[2] * v[0] - q[0] * v[2];
  double uv2 = q[0] * v[1] - q[1] * v[0];
  uv0 += uv0; uv1 += uv1; uv2 += uv2;
  o[0] = v[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
  o[1] = v[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
  o[2] = v[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
}

__device__ __forceinline__ void quat_to_R(const double* q, double* R) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Eigen's Quaterniond(Matrix3d), branch order as in upstream Eigen.
__device__ __forceinline__ void R_to_quat(const double* R, double* q) {
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    // largest diagonal element i, cyclic (j,k); written without dynamic indexing
    if (R[0] >= R[4] && R[0] >= R[8]) {          // i=0,j=1,k=2  (ties resolve to the lower index like Eigen)
      t = sqrt(R[0] - R[4] - R[8] + 1.0);
      q[0] = 0.5 * t; t = 0.5 / t;
      q[3] = (R[7] - R[5]) * t; q[1] = (R[3] + R[1]) * t; q[2] = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) {    // i=1,j=2,k=0
      t = sqrt(R[4] - R[8] - R[0] + 1.0);
      q[1] = 0.5 * t; t = 0.5 / t;
      q[3] = (R[2] - R[6]) * t; q[2] = (R[7] + R[5]) * t; q[0] = (R[1] + R[3]) * t;
    } else {                                      // i=2,j=0,k=1
      t = sqrt(R[8] - R[0] - R[4] + 1.0);
      q[2] = 0.5 * t; t = 0.5 / t;
      q[3] = (R[3] - R[1]) * t; q[0] = (R[2] + R[6]) * t; q[1] = (R[5] + R[7]) * t;
    }
  }
}

__device__ __forceinline__ void quat_normalize_rotation(double* q) {
  if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

// T <- exp(u) * T   (VertexSE3Expmap::oplusImpl, types_six_dof_expmap.h:73-76)
__device__ inline void pose_oplus(const double* u, const double* qt_in, double* qt_out) {
  const double w0 = u[0], w1 = u[1], w2 = u[2];
  const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
  const double Om[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0};
  double Om2[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += Om[i * 3 + k] * Om[k * 3 + j];
      Om2[i * 3 + j] = s;
    }
  double R[9], V[9];
  if (theta < 0.00001) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Om[i] + Om2[i]; V[i] = R[i]; }
  } else {
    const double a = sin(theta) / theta;
    const double b = (1 - cos(theta)) / (theta * theta);
    const double c = (theta - sin(theta)) / (theta * theta * theta);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const double I = (i % 4 == 0) ? 1.0 : 0.0;
      R[i] = I + a * Om[i] + b * Om2[i];
      V[i] = I + b * Om[i] + c * Om2[i];
    }
  }
  double eq[4], et[3], rt[3], nq[4];
  R_to_quat(R, eq);
#pragma unroll
  for (int i = 0; i < 3; ++i) et[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
  quat_normalize_rotation(eq);
  quat_rotate(eq, qt_in + 4, rt);
  // Hamilton product eq * q_in
  nq[3] = eq[3] * qt_in[3] - eq[0] * qt_in[0] - eq[1] * qt_in[1] - eq[2] * qt_in[2];
  nq[0] = eq[3] * qt_in[0] + eq[0] * qt_in[3] + eq[1] * qt_in[2] - eq[2] * qt_in[1];
  nq[1] = eq[3] * qt_in[1] + eq[1] * qt_in[3] + eq[2] * qt_in[0] - eq[0] * qt_in[2];
  nq[2] = eq[3] * qt_in[2] + eq[2] * qt_in[3] + eq[0] * qt_in[1] - eq[1] * qt_in[0];
  quat_normalize_rotation(nq);
  qt_out[0] = nq[0]; qt_out[1] = nq[1]; qt_out[2] = nq[2]; qt_out[3] = nq[3];
  qt_out[4] = et[0] + rt[0]; qt_out[5] = et[1] + rt[1]; qt_out[6] = et[2] + rt[2];
}

// Inverse of the symmetric 3x3 (h00 h01 h02 / . h11 h12 / . . h22) by cofactors * (1/det)
// (Eigen Matrix3d::inverse(), used at g2o/core/block_solver.hpp:389).  Output full row-major.
__device__ __forceinline__ void inv3_sym(double h00, double h01, double h02, double h11, double h12, double h22,
                                         double* inv) {
  const double c00 = h11 * h22 - h12 * h12;
  const double c10 = h12 * h02 - h01 * h22;
  const double c20 = h01 * h12 - h11 * h02;
  const double det = h00 * c00 + h01 * c10 + h02 * c20;
  const double invdet = 1.0 / det;
  inv[0] = c00 * invdet;
  inv[3] = c10 * invdet;
  inv[6] = c20 * invdet;
  inv[1] = (h02 * h12 - h01 * h22) * invdet;
  inv[4] = (h00 * h22 - h02 * h02) * invdet;
  inv[7] = (h01 * h02 - h00 * h12) * invdet;
  inv[2] = (h01 * h12 - h02 * h11) * invdet;
  inv[5] = (h02 * h01 - h00 * h12) * invdet;
  inv[8] = (h00 * h11 - h01 * h01) * invdet;
}

// RobustKernelHuber::robustify, g2o/core/robust_kernel_impl.cpp:78-91
__device__ __forceinline__ void huber(double e, double delta, double& rho0, double& rho1) {
  const double dsqr = delta * delta;
  if (e <= dsqr) { rho0 = e; rho1 = 1.0; }
  else { const double s = sqrt(e); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
}

// Residual of one visual edge; returns chi2 = r^T (info I) r.  kind: 0 mono, 1 stereo.
__device__ __forceinline__ double edge_residual(int kind, const double* qt, const double* cam, const double* X,
                                                const double* obs, double info, double* r, double* Xc) {
  double rot[3];
  quat_rotate(qt, X, rot);
  Xc[0] = rot[0] + qt[4]; Xc[1] = rot[1] + qt[5]; Xc[2] = rot[2] + qt[6];
  if (kind == OSH_EDGE_MONO) {
    r[0] = obs[0] - (cam[0] * Xc[0] / Xc[2] + cam[2]);
    r[1] = obs[1] - (cam[1] * Xc[1] / Xc[2] + cam[3]);
    r[2] = 0.0;
    return r[0] * (info * r[0]) + r[1] * (info * r[1]);
  }
  const float invz = (float)(1.0 / Xc[2]);   // `const float invz = 1.0f/trans_xyz[2];`
  const float bf = (float)cam[4];            // `const float &bf`
  const double u = Xc[0] * (double)invz * cam[0] + cam[2];
  const double v = Xc[1] * (double)invz * cam[1] + cam[3];
  const float bfz = __fmul_rn(bf, invz);     // float product, never fused
  r[0] = obs[0] - u;
  r[1] = obs[1] - v;
  r[2] = obs[2] - (u - (double)bfz);
  return r[0] * (info * r[0]) + r[1] * (info * r[1]) + r[2] * (info * r[2]);
}

// Jacobians: JX[d][3] (d err / d point), Jp[d][6] (d err / d pose, rotation columns first).
// Row 2 is zero for mono so callers can always sum k = 0..2.  R = rotation matrix of the pose.
// The reference divides by z / z^2 term by term (types_six_dof_expmap.cpp:228-273,
// src/OptimizableTypes.cpp:139-160); here 1/z is formed once and multiplied through (the FP64
// divide is ~12 instructions on gfx950): entries differ from the term-by-term form by <= 2 ulp.
__device__ __forceinline__ void edge_jacobians(int kind, const double* R, const double* cam, const double* Xc,
                                               double* JX, double* Jp) {
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  const double iz = 1.0 / z, iz2 = iz * iz;
  if (kind == OSH_EDGE_MONO) {
    const double p00 = -(cam[0] * iz), p02 = cam[0] * x * iz2;
    const double p11 = -(cam[1] * iz), p12 = cam[1] * y * iz2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      JX[j] = p00 * R[j] + p02 * R[6 + j];
      JX[3 + j] = p11 * R[3 + j] + p12 * R[6 + j];
      JX[6 + j] = 0.0;
    }
    // SE3deriv = [0 z -y 1 0 0; -z 0 x 0 1 0; y -x 0 0 0 1]
    Jp[0] = p02 * y;            Jp[1] = p00 * z - p02 * x;  Jp[2] = -p00 * y;  Jp[3] = p00; Jp[4] = 0.0; Jp[5] = p02;
    Jp[6] = -p11 * z + p12 * y; Jp[7] = -p12 * x;           Jp[8] = p11 * x;   Jp[9] = 0.0; Jp[10] = p11; Jp[11] = p12;
#pragma unroll
    for (int j = 0; j < 6; ++j) Jp[12 + j] = 0.0;
  } else {
    const double fx = cam[0], fy = cam[1], bf = cam[4];
    const double fxz = fx * iz, fyz = fy * iz, fxx = fx * x * iz2, fyy = fy * y * iz2, bz2 = bf * iz2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      JX[j] = fxx * R[6 + j] - fxz * R[j];
      JX[3 + j] = fyy * R[6 + j] - fyz * R[3 + j];
      JX[6 + j] = JX[j] - bz2 * R[6 + j];
    }
    Jp[0] = fxx * y;
    Jp[1] = -(fx + fxx * x);
    Jp[2] = fxz * y;
    Jp[3] = -fxz;
    Jp[4] = 0;
    Jp[5] = fxx;
    Jp[6] = fy + fyy * y;
    Jp[7] = -(fyy * x);
    Jp[8] = -(fyz * x);
    Jp[9] = 0;
    Jp[10] = -fyz;
    Jp[11] = fyy;
    Jp[12] = Jp[0] - bz2 * y;
    Jp[13] = Jp[1] + bz2 * x;
    Jp[14] = Jp[2];
    Jp[15] = Jp[3];
    Jp[16] = 0;
    Jp[17] = Jp[5] - bz2;
  }
}

// ---- KannalaBrandt8 (fisheye) monocular edge: ORB_SLAM3::EdgeSE3ProjectXYZ through KannalaBrandt8::project(Vector3d)
// (src/CameraModels/KannalaBrandt8.cpp:45-63) and ::projectJac (:147-175).  cam = fx fy cx cy (pose_cam), kb = k1..k4
// (mvParameters[4..7]).  The reference rounds theta and psi to float32 (atan2f / sqrtf on double arguments); float32
// atan2 is taken as its correctly rounded value (FP64 atan2 rounded once), which is libm independent and differs from
// glibc's atan2f only in rare 1-ulp cases (the parity tests' CPU restatement uses the same convention).
__device__ __forceinline__ float atan2f_rn(float y, float x) { return (float)atan2((double)y, (double)x); }
__device__ __forceinline__ float sqrtf_rn(float x) { return (float)sqrt((double)x); }

// KannalaBrandt8::project(Vector3d) (src/CameraModels/KannalaBrandt8.cpp:45-63)
__device__ __forceinline__ void kb8_project(const double* cam, const double* kb, const double* Xc, double& u, double& v) {
  const double x2_plus_y2 = Xc[0] * Xc[0] + Xc[1] * Xc[1];
  const double theta = (double)atan2f_rn(sqrtf_rn((float)x2_plus_y2), (float)Xc[2]);
  const double psi = (double)atan2f_rn((float)Xc[1], (float)Xc[0]);
  const double theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
               theta9 = theta7 * theta2;
  const double rr = theta + kb[0] * theta3 + kb[1] * theta5 + kb[2] * theta7 + kb[3] * theta9;
  u = cam[0] * rr * cos(psi) + cam[2];
  v = cam[1] * rr * sin(psi) + cam[3];
}
// KannalaBrandt8::projectJac (src/CameraModels/KannalaBrandt8.cpp:147-175): J row-major 2x3
__device__ __forceinline__ void kb8_project_jac(const double* cam, const double* kb, const double* Xc, double* J) {
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  const double x2 = x * x, y2 = y * y, z2 = z * z;
  const double r2 = x2 + y2, rr = sqrt(r2), r3 = r2 * rr;
  const double theta = atan2(rr, z);
  const double theta2 = theta * theta, theta3 = theta2 * theta, theta4 = theta2 * theta2, theta5 = theta4 * theta,
               theta6 = theta2 * theta4, theta7 = theta6 * theta, theta8 = theta4 * theta4, theta9 = theta8 * theta;
  const double f = theta + theta3 * kb[0] + theta5 * kb[1] + theta7 * kb[2] + theta9 * kb[3];
  const double fd = 1 + 3 * kb[0] * theta2 + 5 * kb[1] * theta4 + 7 * kb[2] * theta6 + 9 * kb[3] * theta8;
  const double den = r2 * (r2 + z2);
  J[0] = cam[0] * (fd * z * x2 / den + f * y2 / r3);
  J[3] = cam[1] * (fd * z * y * x / den - f * y * x / r3);
  J[1] = cam[0] * (fd * z * y * x / den - f * y * x / r3);
  J[4] = cam[1] * (fd * z * y2 / den + f * x2 / r3);
  J[2] = -cam[0] * fd * x / (r2 + z2);
  J[5] = -cam[1] * fd * y / (r2 + z2);
}

__device__ __forceinline__ double edge_residual_kb8(const double* qt, const double* cam, const double* kb, const double* X,
                                                    const double* obs, double info, double* r, double* Xc) {
  double rot[3];
  quat_rotate(qt, X, rot);
  Xc[0] = rot[0] + qt[4]; Xc[1] = rot[1] + qt[5]; Xc[2] = rot[2] + qt[6];
  double u, v;
  kb8_project(cam, kb, Xc, u, v);
  r[0] = obs[0] - u;
  r[1] = obs[1] - v;
  r[2] = 0.0;
  return r[0] * (info * r[0]) + r[1] * (info * r[1]);
}

__device__ __forceinline__ void edge_jacobians_kb8(const double* R, const double* cam, const double* kb, const double* Xc,
                                                   double* JX, double* Jp) {
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  // Pm = -projectJac(Xc)  (src/OptimizableTypes.cpp:146)
  double Pm[6];
  kb8_project_jac(cam, kb, Xc, Pm);
#pragma unroll
  for (int k = 0; k < 6; ++k) Pm[k] = -Pm[k];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) JX[3 * i + j] = Pm[3 * i] * R[j] + Pm[3 * i + 1] * R[3 + j] + Pm[3 * i + 2] * R[6 + j];
    // SE3deriv = [0 z -y 1 0 0; -z 0 x 0 1 0; y -x 0 0 0 1]
    Jp[6 * i + 0] = Pm[3 * i + 2] * y - Pm[3 * i + 1] * z;
    Jp[6 * i + 1] = Pm[3 * i] * z - Pm[3 * i + 2] * x;
    Jp[6 * i + 2] = Pm[3 * i + 1] * x - Pm[3 * i] * y;
    Jp[6 * i + 3] = Pm[3 * i]; Jp[6 * i + 4] = Pm[3 * i + 1]; Jp[6 * i + 5] = Pm[3 * i + 2];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) JX[6 + j] = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) Jp[12 + j] = 0.0;
}

// 64-lane butterfly sum: every lane ends with the same total (deterministic order).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

}  // namespace dev
}  // namespace osh
