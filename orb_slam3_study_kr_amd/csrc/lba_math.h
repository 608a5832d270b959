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

// 1 / z and sqrt(e), 1 / sqrt(e) from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~26 bits) and two Newton steps: within an ulp or
// two of the IEEE results at about half the instructions of the compiler's divide / square-root expansions.  Used by the
// per-edge cores of the batch kernels, which are FP64-issue bound; never where the reference's rounding is restated on purpose.
__device__ __forceinline__ double rcp_nr(double z) {
  double x = __builtin_amdgcn_rcp(z);
  x = fma(x, fma(-z, x, 1.0), x);
  x = fma(x, fma(-z, x, 1.0), x);
  return x;
}
__device__ __forceinline__ void sqrt_rsqrt_nr(double e, double& s, double& inv_s) {
  const double y = __builtin_amdgcn_rsq(e);
  double g = e * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  s = g; inv_s = h + h;
}
// RobustKernelHuber::robustify (robust_kernel_impl.cpp:78-91) with sqrt / divide replaced as above
__device__ __forceinline__ void huber_fast(double e, double delta, double& rho0, double& rho1) {
  const double dsqr = delta * delta;
  double s, inv_s;
  sqrt_rsqrt_nr(fmax(e, dsqr), s, inv_s);        // branch-free: the inlier lanes compute on dsqr and discard
  const bool in = e <= dsqr;
  rho0 = in ? e : 2 * s * delta - dsqr;
  rho1 = in ? 1.0 : delta * inv_s;
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

// EdgeSE3ProjectXYZOnlyPoseToBody / EdgeSE3ProjectXYZToBody (include/OptimizableTypes.h:62-87,117-144): the right camera of a
// fisheye rig.  computeError maps through the SE3Quat product mTrl * T_lw (normalised quaternion, se3quat.h:104-110),
// linearizeOplus through the two transforms one after the other (src/OptimizableTypes.cpp:91-107): each is restated as it is.
// cam2: fx fy cx cy k1..k4 of the right camera, trl: qx qy qz qw tx ty tz.  Xl: point in the LEFT camera frame.
__device__ __forceinline__ double edge_residual_body(const double* qt, const double* cam2, const double* trl, const double* X,
                                                     const double* obs, double info, double* r, double* Xl, double* Xe) {
  double rot[3], q[4], tr[3], rx[3];
  quat_rotate(qt, X, rot);
  Xl[0] = rot[0] + qt[4]; Xl[1] = rot[1] + qt[5]; Xl[2] = rot[2] + qt[6];
  q[3] = trl[3] * qt[3] - trl[0] * qt[0] - trl[1] * qt[1] - trl[2] * qt[2];
  q[0] = trl[3] * qt[0] + trl[0] * qt[3] + trl[1] * qt[2] - trl[2] * qt[1];
  q[1] = trl[3] * qt[1] + trl[1] * qt[3] + trl[2] * qt[0] - trl[0] * qt[2];
  q[2] = trl[3] * qt[2] + trl[2] * qt[3] + trl[0] * qt[1] - trl[1] * qt[0];
  quat_normalize_rotation(q);
  quat_rotate(trl, qt + 4, tr);
  quat_rotate(q, X, rx);
  Xe[0] = rx[0] + (tr[0] + trl[4]); Xe[1] = rx[1] + (tr[1] + trl[5]); Xe[2] = rx[2] + (tr[2] + trl[6]);
  double u, v;
  kb8_project(cam2, cam2 + 4, Xe, u, v);
  r[0] = obs[0] - u;
  r[1] = obs[1] - v;
  r[2] = 0.0;
  return r[0] * (info * r[0]) + r[1] * (info * r[1]);
}
// pose Jacobian only (the unary edge): Jp = -projectJac(X_r) * Rrl * SE3deriv(X_l), X_r = mTrl.map(X_l); rows 0..1, row 2 zero
__device__ __forceinline__ void edge_jacobian_pose_body(const double* cam2, const double* trl, const double* Xl, double* Jp) {
  double Rrl[9], Xr[3], J2[6], Pm[6];
  quat_to_R(trl, Rrl);
  quat_rotate(trl, Xl, Xr);
  Xr[0] += trl[4]; Xr[1] += trl[5]; Xr[2] += trl[6];
  kb8_project_jac(cam2, cam2 + 4, Xr, J2);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Pm[3 * i + j] = -(J2[3 * i] * Rrl[j] + J2[3 * i + 1] * Rrl[3 + j] + J2[3 * i + 2] * Rrl[6 + j]);
  const double x = Xl[0], y = Xl[1], z = Xl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    Jp[6 * i + 0] = Pm[3 * i + 2] * y - Pm[3 * i + 1] * z;
    Jp[6 * i + 1] = Pm[3 * i] * z - Pm[3 * i + 2] * x;
    Jp[6 * i + 2] = Pm[3 * i + 1] * x - Pm[3 * i] * y;
    Jp[6 * i + 3] = Pm[3 * i]; Jp[6 * i + 4] = Pm[3 * i + 1]; Jp[6 * i + 5] = Pm[3 * i + 2];
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) Jp[12 + j] = 0.0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Compact description of a visual edge at the linearisation point.  Both reference Jacobians factor through the
// camera-frame point Xc = R X + t:   d err / d X = Pm R,   d err / d pose = Pm [ -[Xc]x | I ]   with Pm = -projectJac(Xc)
// (src/OptimizableTypes.cpp:139-160, types_six_dof_expmap.cpp:228-273: SE3deriv IS [ -[Xc]x | I ]; for the body edge of a
// fisheye rig Pm = -projectJac2(Xr) Rrl, src/OptimizableTypes.cpp:192-213).  With the robustified weight w = rho' invSigma2:
//   Q = w Pm^T Pm (3x3 symmetric)      g = Pm^T (-w r)
//   Hll += R^T Q R     b_l += R^T g     Hpl = D^T Q R     Hpp += D^T Q D     b_p += D^T g       D = [ -[Xc]x | I ]
// so an edge is described by Xc, the six entries of Q and g; two edges on one (keyframe, landmark) block simply add their
// Q and g.  Every kernel below forms its blocks from this description; nothing of size 6x3 is ever stored.
// Q is stored as 00 01 02 11 12 22.  For a pinhole camera Q01 == 0 (fx and fy rows do not mix).
// ---------------------------------------------------------------------------------------------------------------------

// Residual of a pinhole mono / rectified-stereo edge from the pose's rotation MATRIX (Xc = R X + t: 9 FMAs instead of the
// quaternion form) and ONE reciprocal of z (rcp_nr) shared by the projection and the Jacobian entries.  The stereo residual
// keeps the reference's float32 1/z and float bf product (types_six_dof_expmap.cpp:190-191).  rec = u v u_r +-invSigma2.
__device__ __forceinline__ double edge_residual_pinhole(const double* R, const double* t, const double* cam, const double* X,
                                                        const double* rec, double* r, double* Xc, double& iz) {
  Xc[0] = fma(R[0], X[0], fma(R[1], X[1], fma(R[2], X[2], t[0])));
  Xc[1] = fma(R[3], X[0], fma(R[4], X[1], fma(R[5], X[2], t[1])));
  Xc[2] = fma(R[6], X[0], fma(R[7], X[1], fma(R[8], X[2], t[2])));
  iz = rcp_nr(Xc[2]);
  const double info = fabs(rec[3]);
  if (!(rec[3] > 0.0)) {   // monocular
    r[0] = rec[0] - (cam[0] * Xc[0] * iz + cam[2]);
    r[1] = rec[1] - (cam[1] * Xc[1] * iz + cam[3]);
    r[2] = 0.0;
    return r[0] * (info * r[0]) + r[1] * (info * r[1]);
  }
  const float invz = (float)iz;              // `const float invz = 1.0f/trans_xyz[2];`
  const float bf = (float)cam[4];            // `const float &bf`
  const double u = Xc[0] * (double)invz * cam[0] + cam[2];
  const double v = Xc[1] * (double)invz * cam[1] + cam[3];
  const float bfz = __fmul_rn(bf, invz);     // float product, never fused
  r[0] = rec[0] - u;
  r[1] = rec[1] - v;
  r[2] = rec[2] - (u - (double)bfz);
  return r[0] * (info * r[0]) + r[1] * (info * r[1]) + r[2] * (info * r[2]);
}

// Pinhole mono / rectified-stereo edge.  rec = u v u_r +-invSigma2 (sign bit set: monocular).  Returns rho(chi2) in rho0.
// R, t: rotation matrix and translation of the pose.
__device__ __forceinline__ void edge_core_pinhole(const double* R, const double* t, const double* cam, const double* X, const double* rec,
                                                  double huber_mono, double huber_stereo, double* Xc, double* Q, double* g,
                                                  double& rho0) {
  const bool stereo = rec[3] > 0.0;
  const double info = fabs(rec[3]);
  double r[3], iz;
  const double chi2 = edge_residual_pinhole(R, t, cam, X, rec, r, Xc, iz);
  double rho1;
  huber_fast(chi2, stereo ? huber_stereo : huber_mono, rho0, rho1);
  const double ww = rho1 * info;                       // robustInformation (first order only, base_edge.h:96-102)
  const double wr0 = -(info * r[0]) * rho1, wr1 = -(info * r[1]) * rho1, wr2 = -(info * r[2]) * rho1;   // r[2] == 0 for mono
  const double iz2 = iz * iz;
  const double pa = cam[0] * iz, pb = cam[1] * iz, pc = cam[0] * Xc[0] * iz2, pd = cam[1] * Xc[1] * iz2;
  const double pce = stereo ? pc - cam[4] * iz2 : 0.0;   // third row of Pm: (-pa, 0, pc - bf/z^2); absent for mono
  // rows of Pm: (-pa, 0, pc)  (0, -pb, pd)  [(-pa, 0, pce)]
  g[0] = -pa * (wr0 + wr2);
  g[1] = -pb * wr1;
  g[2] = pc * wr0 + pd * wr1 + pce * wr2;
  const double wa = ww * pa, wb = ww * pb;
  Q[0] = stereo ? wa * pa + wa * pa : wa * pa;
  Q[1] = 0.0;
  Q[2] = -(wa * (pc + pce));
  Q[3] = wb * pb;
  Q[4] = -(wb * pd);
  Q[5] = ww * (pc * pc + pd * pd + pce * pce);
}

// Adds w Pm^T Pm and Pm^T (-w r) of a two-row edge with dense Pm (2x3 row-major) to Q / g.
__device__ __forceinline__ void edge_core_add_rows2(const double* Pm, double ww, double wr0, double wr1, double* Q, double* g) {
  g[0] += Pm[0] * wr0 + Pm[3] * wr1;
  g[1] += Pm[1] * wr0 + Pm[4] * wr1;
  g[2] += Pm[2] * wr0 + Pm[5] * wr1;
  const double a0 = ww * Pm[0], a1 = ww * Pm[1], a2 = ww * Pm[2], b0 = ww * Pm[3], b1 = ww * Pm[4], b2 = ww * Pm[5];
  Q[0] += a0 * Pm[0] + b0 * Pm[3];
  Q[1] += a0 * Pm[1] + b0 * Pm[4];
  Q[2] += a0 * Pm[2] + b0 * Pm[5];
  Q[3] += a1 * Pm[1] + b1 * Pm[4];
  Q[4] += a1 * Pm[2] + b1 * Pm[5];
  Q[5] += a2 * Pm[2] + b2 * Pm[5];
}

// KannalaBrandt8 window: kind is one of the sorted-edge kinds 0 mono (left camera), 2 body (right camera through Trl),
// 3 both on one block.  rec = left observation record, rec2 = right observation record (u v - invSigma2).
// chi_l / chi_r: chi2 of the left / right edge (0 when absent); rho0: sum of the robustified chi2 of the edges present.
__device__ __forceinline__ void edge_core_kb8(int kind, const double* qt, const double* cam, const double* kb, const double* cam2,
                                              const double* trl, const double* X, const double* rec, const double* rec2,
                                              double huber_mono, double* Xc, double* Q, double* g, double& rho0,
                                              double& chi_l, double& chi_r) {
  double rot[3];
  quat_rotate(qt, X, rot);
  Xc[0] = rot[0] + qt[4]; Xc[1] = rot[1] + qt[5]; Xc[2] = rot[2] + qt[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) Q[k] = 0.0;
  g[0] = g[1] = g[2] = 0.0;
  rho0 = 0.0; chi_l = 0.0; chi_r = 0.0;
  if (kind != 2) {   // left edge: EdgeSE3ProjectXYZ through KannalaBrandt8 (src/OptimizableTypes.cpp:139-160)
    const double info = fabs(rec[3]);
    double u, v;
    kb8_project(cam, kb, Xc, u, v);
    const double r0 = rec[0] - u, r1 = rec[1] - v;
    chi_l = r0 * (info * r0) + r1 * (info * r1);
    double rh0, rh1;
    huber(chi_l, huber_mono, rh0, rh1);
    rho0 += rh0;
    double Pm[6];
    kb8_project_jac(cam, kb, Xc, Pm);
#pragma unroll
    for (int k = 0; k < 6; ++k) Pm[k] = -Pm[k];
    edge_core_add_rows2(Pm, rh1 * info, -(info * r0) * rh1, -(info * r1) * rh1, Q, g);
  }
  if (kind >= 2) {   // right edge: EdgeSE3ProjectXYZToBody (include/OptimizableTypes.h:125-130, src/OptimizableTypes.cpp:192-213)
    // (values, not a pointer, are selected: a pointer into either record would park both in scratch memory)
    const bool solo = kind == 2;
    const double ro[4] = {solo ? rec[0] : rec2[0], solo ? rec[1] : rec2[1], 0.0, solo ? rec[3] : rec2[3]};
    const double info = fabs(ro[3]);
    double Rrl[9], Xr[3], Xe[3];
    quat_to_R(trl, Rrl);
    // linearizeOplus maps through the two transforms one after the other (X_r = mTrl.map(T_lw.map(X_w)), OptimizableTypes.cpp:198),
    // computeError through their SE3Quat product ((mTrl * v1->estimate()).map(X), OptimizableTypes.h:129): with float32 theta /
    // psi inside the projection the two differ by a staircase step now and then, so each is restated as it is
    quat_rotate(trl, Xc, Xr);
    Xr[0] += trl[4]; Xr[1] += trl[5]; Xr[2] += trl[6];
    {
      double q[4], tr[3], rx[3];
      q[3] = trl[3] * qt[3] - trl[0] * qt[0] - trl[1] * qt[1] - trl[2] * qt[2];
      q[0] = trl[3] * qt[0] + trl[0] * qt[3] + trl[1] * qt[2] - trl[2] * qt[1];
      q[1] = trl[3] * qt[1] + trl[1] * qt[3] + trl[2] * qt[0] - trl[0] * qt[2];
      q[2] = trl[3] * qt[2] + trl[2] * qt[3] + trl[0] * qt[1] - trl[1] * qt[0];
      quat_normalize_rotation(q);                     // SE3Quat::operator* normalises (se3quat.h:104-110)
      quat_rotate(trl, qt + 4, tr);
      quat_rotate(q, X, rx);
      Xe[0] = rx[0] + (tr[0] + trl[4]); Xe[1] = rx[1] + (tr[1] + trl[5]); Xe[2] = rx[2] + (tr[2] + trl[6]);
    }
    double u, v;
    kb8_project(cam2, cam2 + 4, Xe, u, v);
    const double r0 = ro[0] - u, r1 = ro[1] - v;
    chi_r = r0 * (info * r0) + r1 * (info * r1);
    double rh0, rh1;
    huber(chi_r, huber_mono, rh0, rh1);    // rk->setDelta(thHuberMono) (src/Optimizer.cc:1386)
    rho0 += rh0;
    double J2[6], Pm[6];
    kb8_project_jac(cam2, cam2 + 4, Xr, J2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Pm[3 * i + j] = -(J2[3 * i] * Rrl[j] + J2[3 * i + 1] * Rrl[3 + j] + J2[3 * i + 2] * Rrl[6 + j]);
    edge_core_add_rows2(Pm, rh1 * info, -(info * r0) * rh1, -(info * r1) * rh1, Q, g);
  }
}

// M = Q R  (R row-major 3x3).  DENSE: Q01 may be non-zero (fisheye); the pinhole kernels skip those terms.
template <bool DENSE>
__device__ __forceinline__ void core_QR(const double* Q, const double* R, double* M) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    M[j] = Q[0] * R[j] + Q[2] * R[6 + j];
    M[3 + j] = Q[3] * R[3 + j] + Q[4] * R[6 + j];
    M[6 + j] = Q[2] * R[j] + Q[4] * R[3 + j] + Q[5] * R[6 + j];
    if (DENSE) { M[j] += Q[1] * R[3 + j]; M[3 + j] += Q[1] * R[j]; }
  }
}

// hl[0..5] = upper(R^T Q R), hl[6..8] = R^T g: the edge's share of Hll and b_l.
template <bool DENSE>
__device__ __forceinline__ void core_landmark_side(const double* Q, const double* g, const double* R, double* hl) {
  double M[9];
  core_QR<DENSE>(Q, R, M);
  hl[0] = R[0] * M[0] + R[3] * M[3] + R[6] * M[6];
  hl[1] = R[0] * M[1] + R[3] * M[4] + R[6] * M[7];
  hl[2] = R[0] * M[2] + R[3] * M[5] + R[6] * M[8];
  hl[3] = R[1] * M[1] + R[4] * M[4] + R[7] * M[7];
  hl[4] = R[1] * M[2] + R[4] * M[5] + R[7] * M[8];
  hl[5] = R[2] * M[2] + R[5] * M[5] + R[8] * M[8];
  hl[6] = R[0] * g[0] + R[3] * g[1] + R[6] * g[2];
  hl[7] = R[1] * g[0] + R[4] * g[1] + R[7] * g[2];
  hl[8] = R[2] * g[0] + R[5] * g[1] + R[8] * g[2];
}

// Rows of the Hpl block W = D^T Q R times the landmark factor F (upper triangular, F F^T = (Hll + lambda I)^-1,
// stored 00 01 02 11 12 22): WF[r*3 + m], r = 0..5 (rotation rows first), m = 0..2.
template <bool DENSE>
__device__ __forceinline__ void core_WF(const double* Xc, const double* Q, const double* R, const double* F, double* WF) {
  double M[9];
  core_QR<DENSE>(Q, R, M);
  double* T = WF + 9;   // rows 3..5 of W F are T = Q R F
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    T[3 * i] = M[3 * i] * F[0];
    T[3 * i + 1] = M[3 * i] * F[1] + M[3 * i + 1] * F[3];
    T[3 * i + 2] = M[3 * i] * F[2] + M[3 * i + 1] * F[4] + M[3 * i + 2] * F[5];
  }
  // rows 0..2: [Xc]x T, column by column: Xc x T[:, m]
  const double x = Xc[0], y = Xc[1], z = Xc[2];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    WF[m] = y * T[6 + m] - z * T[3 + m];
    WF[3 + m] = z * T[m] - x * T[6 + m];
    WF[6 + m] = x * T[3 + m] - y * T[m];
  }
}

// Hpp (upper triangle, 21 entries row-major) += D^T Q D and b_p += D^T g.
template <bool DENSE>
__device__ __forceinline__ void core_pose_side(const double* Xc, const double* Q, const double* g, double* H, double* b) {
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  // Mx = [Xc]x Q : column j = Xc x Q[:, j]
  const double q00 = Q[0], q01 = DENSE ? Q[1] : 0.0, q02 = Q[2], q11 = Q[3], q12 = Q[4], q22 = Q[5];
  double m00, m01, m02, m10, m11, m12, m20, m21, m22;
  if (DENSE) {
    m00 = y * q02 - z * q01; m10 = z * q00 - x * q02; m20 = x * q01 - y * q00;
    m01 = y * q12 - z * q11; m11 = z * q01 - x * q12; m21 = x * q11 - y * q01;
  } else {
    m00 = y * q02;           m10 = z * q00 - x * q02; m20 = -(y * q00);
    m01 = y * q12 - z * q11; m11 = -(x * q12);        m21 = x * q11;
  }
  m02 = y * q22 - z * q12; m12 = z * q02 - x * q22; m22 = x * q12 - y * q02;
  (void)q01;
  // top-left A = Mx [Xc]x^T : A[i][0] = y M[i][2] - z M[i][1], A[i][1] = z M[i][0] - x M[i][2], A[i][2] = x M[i][1] - y M[i][0]
  H[0] += y * m02 - z * m01;  H[1] += z * m00 - x * m02;  H[2] += x * m01 - y * m00;
  H[3] += m00; H[4] += m01; H[5] += m02;                      // top-right Mx, row 0
  H[6] += z * m10 - x * m12;  H[7] += x * m11 - y * m10;
  H[8] += m10; H[9] += m11; H[10] += m12;
  H[11] += x * m21 - y * m20;
  H[12] += m20; H[13] += m21; H[14] += m22;
  H[15] += q00; if (DENSE) H[16] += Q[1]; H[17] += q02;       // bottom-right Q
  H[18] += q11; H[19] += q12;
  H[20] += q22;
  b[0] += y * g[2] - z * g[1];
  b[1] += z * g[0] - x * g[2];
  b[2] += x * g[1] - y * g[0];
  b[3] += g[0]; b[4] += g[1]; b[5] += g[2];
}

// -(Hpl^T x) of one edge for the back-substitution: -R^T Q (D x),  D x = x[3..5] + x[0..2] x Xc
template <bool DENSE>
__device__ __forceinline__ void core_backsub(const double* Xc, const double* Q, const double* R, const double* xp, double* c) {
  const double v0 = xp[3] + (xp[1] * Xc[2] - xp[2] * Xc[1]);
  const double v1 = xp[4] + (xp[2] * Xc[0] - xp[0] * Xc[2]);
  const double v2 = xp[5] + (xp[0] * Xc[1] - xp[1] * Xc[0]);
  double s0 = Q[0] * v0 + Q[2] * v2, s1 = Q[3] * v1 + Q[4] * v2;
  const double s2 = Q[2] * v0 + Q[4] * v1 + Q[5] * v2;
  if (DENSE) { s0 += Q[1] * v1; s1 += Q[1] * v0; }
  c[0] = -(R[0] * s0 + R[3] * s1 + R[6] * s2);
  c[1] = -(R[1] * s0 + R[4] * s1 + R[7] * s2);
  c[2] = -(R[2] * s0 + R[5] * s1 + R[8] * s2);
}

// Landmark factor: D = Hll + lambda I = C C^T (Cholesky), F = C^-T (upper triangular, F F^T = D^-1), u = F^T b_l.
// This is the role of `D->inverse()` at block_solver.hpp:389: the Schur products B_i D^-1 B_j^T are formed as
// (B_i F)(B_j F)^T and the solution D^-1 c as F (F^T c).  Output dl[0..5] = F 00 01 02 11 12 22, dl[6..8] = u.
__device__ __forceinline__ void landmark_factor(const double* hl, const double* bl, double lambda, double* dl) {
  const double d00 = hl[0] + lambda, d01 = hl[1], d02 = hl[2], d11 = hl[3] + lambda, d12 = hl[4], d22 = hl[5] + lambda;
  const double c00 = sqrt(d00), i00 = 1.0 / c00;
  const double c10 = d01 * i00, c20 = d02 * i00;
  const double c11 = sqrt(d11 - c10 * c10), i11 = 1.0 / c11;
  const double c21 = (d12 - c20 * c10) * i11;
  const double c22 = sqrt(d22 - c20 * c20 - c21 * c21), i22 = 1.0 / c22;
  const double n10 = -(c10 * i00) * i11;                 // C^-1 (lower): n10, n20, n21 below the diagonal i00, i11, i22
  const double n21 = -(c21 * i11) * i22;
  const double n20 = -(c20 * i00 + c21 * n10) * i22;
  dl[0] = i00; dl[1] = n10; dl[2] = n20; dl[3] = i11; dl[4] = n21; dl[5] = i22;
  dl[6] = i00 * bl[0];
  dl[7] = n10 * bl[0] + i11 * bl[1];
  dl[8] = n20 * bl[0] + n21 * bl[1] + i22 * bl[2];
}

// 64-lane butterfly sum: every lane ends with the same total (deterministic order).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// Sum over the wavefront without the LDS crossbar: four DPP steps leave every lane with the sum of its row of 16, the four row
// sums are then read as scalars and added in row order.  (wave_sum's xor butterfly is 12 ds_bpermute per value.)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);   // row_half_mirror
  v += dpp_f64<0x140>(v);   // row_mirror
  double r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 16 * k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 16 * k);
    r[k] = __hiloint2double(hi, lo);
  }
  return ((r[0] + r[1]) + r[2]) + r[3];
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

}  // namespace dev
}  // namespace osh
