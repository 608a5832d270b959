// liba_math.h -- device arithmetic of the stereo-inertial edges (gfx950).
//
// Numerical contract = what the reference computes on the CPU (paths relative to /root/reference):
//   ExpSO3 / LogSO3 / RightJacobianSO3 / InverseRightJacobianSO3 in double            src/G2oTypes.cc:777-861
//     (ExpSO3 re-orthonormalises with a JacobiSVD U V^T; here: the same orthogonal polar factor by Newton iteration)
//   IMU::Preintegrated::GetDeltaRotation / Velocity / Position in FLOAT32              src/ImuTypes.cc:277-309
//     with the bias estimates rounded to float first (IMU::Bias, src/G2oTypes.cc:522) and Sophus::SO3f::exp
//     (Thirdparty/Sophus/sophus/so3.hpp:583-619)
//   EdgeInertial residual / Jacobians                                                   src/G2oTypes.cc:513-594
//   EdgeMono / EdgeStereo with the ImuCamPose parameterisation                           src/G2oTypes.cc:170-220,349-427
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/orbslam3_hip.h"

namespace osh {
namespace imu {

__device__ __forceinline__ void m3_mul(const double* A, const double* B, double* C) {
  double T[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = T[i];
}
__device__ __forceinline__ void m3_tmul(const double* A, const double* B, double* C) {  // A^T B
  double T[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = T[i];
}
__device__ __forceinline__ void m3_vec(const double* A, const double* v, double* o) {
  const double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2], t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
  const double t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  o[0] = t0; o[1] = t1; o[2] = t2;
}
__device__ __forceinline__ void m3_tvec(const double* A, const double* v, double* o) {  // A^T v
  const double t0 = A[0] * v[0] + A[3] * v[1] + A[6] * v[2], t1 = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
  const double t2 = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
  o[0] = t0; o[1] = t1; o[2] = t2;
}
__device__ __forceinline__ void m3_hat(const double* v, double* W) {
  W[0] = 0; W[1] = -v[2]; W[2] = v[1]; W[3] = v[2]; W[4] = 0; W[5] = -v[0]; W[6] = -v[1]; W[7] = v[0]; W[8] = 0;
}
__device__ __forceinline__ void m3_inv(const double* m, double* inv) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c10 = m[5] * m[6] - m[3] * m[8], c20 = m[3] * m[7] - m[4] * m[6];
  const double id = 1.0 / (m[0] * c00 + m[1] * c10 + m[2] * c20);
  const double T[9] = {c00 * id, (m[2] * m[7] - m[1] * m[8]) * id, (m[1] * m[5] - m[2] * m[4]) * id,
                       c10 * id, (m[0] * m[8] - m[2] * m[6]) * id, (m[2] * m[3] - m[0] * m[5]) * id,
                       c20 * id, (m[1] * m[6] - m[0] * m[7]) * id, (m[0] * m[4] - m[1] * m[3]) * id};
#pragma unroll
  for (int i = 0; i < 9; ++i) inv[i] = T[i];
}
// NormalizeRotation (include/G2oTypes.h:67-71): U V^T == orthogonal polar factor
__device__ inline void normalize_rotation(double* R) {
  for (int it = 0; it < 12; ++it) {
    double Ri[9], d = 0;
    m3_inv(R, Ri);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double nv = 0.5 * (R[i * 3 + j] + Ri[j * 3 + i]);
        d = fmax(d, fabs(nv - R[i * 3 + j]));
        R[i * 3 + j] = nv;
      }
    if (d < 1e-16) break;
  }
}
__device__ inline void normalize_rotation_f(float* R) {  // IMU::NormalizeRotation, src/ImuTypes.cc:34-37
  for (int it = 0; it < 12; ++it) {
    const float c00 = __fsub_rn(__fmul_rn(R[4], R[8]), __fmul_rn(R[5], R[7]));
    const float c10 = __fsub_rn(__fmul_rn(R[5], R[6]), __fmul_rn(R[3], R[8]));
    const float c20 = __fsub_rn(__fmul_rn(R[3], R[7]), __fmul_rn(R[4], R[6]));
    const float det = __fadd_rn(__fadd_rn(__fmul_rn(R[0], c00), __fmul_rn(R[1], c10)), __fmul_rn(R[2], c20));
    const float id = 1.0f / det;
    const float Ri[9] = {c00 * id, __fsub_rn(__fmul_rn(R[2], R[7]), __fmul_rn(R[1], R[8])) * id, __fsub_rn(__fmul_rn(R[1], R[5]), __fmul_rn(R[2], R[4])) * id,
                         c10 * id, __fsub_rn(__fmul_rn(R[0], R[8]), __fmul_rn(R[2], R[6])) * id, __fsub_rn(__fmul_rn(R[2], R[3]), __fmul_rn(R[0], R[5])) * id,
                         c20 * id, __fsub_rn(__fmul_rn(R[1], R[6]), __fmul_rn(R[0], R[7])) * id, __fsub_rn(__fmul_rn(R[0], R[4]), __fmul_rn(R[1], R[3])) * id};
    float d = 0, N[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        N[i * 3 + j] = 0.5f * __fadd_rn(R[i * 3 + j], Ri[j * 3 + i]);
        d = fmaxf(d, fabsf(N[i * 3 + j] - R[i * 3 + j]));
      }
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = N[i];
    if (d < 1e-7f) break;
  }
}
__device__ inline void exp_so3(const double* w, double* R) {  // src/G2oTypes.cc:782-798
  const double d2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], d = sqrt(d2);
  double W[9], W2[9];
  m3_hat(w, W);
  m3_mul(W, W, W2);
  const double a = (d < 1e-5) ? 1.0 : sin(d) / d, b = (d < 1e-5) ? 0.5 : (1.0 - cos(d)) / d2;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    R[i] = (d < 1e-5) ? I + W[i] + 0.5 * W2[i] : I + W[i] * sin(d) / d + W2[i] * (1.0 - cos(d)) / d2;
  }
  (void)a; (void)b;
  normalize_rotation(R);
}
__device__ inline void log_so3(const double* R, double* w) {  // src/G2oTypes.cc:800-813
  const double tr = R[0] + R[4] + R[8];
  w[0] = (R[7] - R[5]) / 2; w[1] = (R[2] - R[6]) / 2; w[2] = (R[3] - R[1]) / 2;
  const double costheta = (tr - 1.0) * 0.5f;
  if (costheta > 1 || costheta < -1) return;
  const double theta = acos(costheta), s = sin(theta);
  if (fabs(s) < 1e-5) return;
  w[0] = theta * w[0] / s; w[1] = theta * w[1] / s; w[2] = theta * w[2] / s;
}
__device__ inline void inv_right_jac(const double* v, double* J) {  // :820-833
  const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrt(d2);
  double W[9], W2[9];
  m3_hat(v, W); m3_mul(W, W, W2);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    J[i] = (d < 1e-5) ? I : I + W[i] / 2 + W2[i] * (1.0 / d2 - (1.0 + cos(d)) / (2.0 * d * sin(d)));
  }
}
__device__ inline void right_jac(const double* v, double* J) {  // :840-853
  const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrt(d2);
  double W[9], W2[9];
  m3_hat(v, W); m3_mul(W, W, W2);
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    J[i] = (d < 1e-5) ? I : I - W[i] * (1.0 - cos(d)) / d2 + W2[i] * (d - sin(d)) / (d2 * d);
  }
}

// ---- float32 preintegration getters (every product/sum rounded to float, no contraction)
__device__ __forceinline__ float dot3f(const float* a, const float* b) {
  return __fadd_rn(__fadd_rn(__fmul_rn(a[0], b[0]), __fmul_rn(a[1], b[1])), __fmul_rn(a[2], b[2]));
}
__device__ inline void so3f_exp_matrix(const float* v, float* R) {
  const float theta_sq = dot3f(v, v);
  float imag, real;
  if (theta_sq < 1e-5f * 1e-5f) {
    const float theta_po4 = __fmul_rn(theta_sq, theta_sq);
    imag = __fadd_rn(__fsub_rn(0.5f, __fmul_rn((float)(1.0 / 48.0), theta_sq)), __fmul_rn((float)(1.0 / 3840.0), theta_po4));
    real = __fadd_rn(__fsub_rn(1.0f, __fmul_rn((float)(1.0 / 8.0), theta_sq)), __fmul_rn((float)(1.0 / 384.0), theta_po4));
  } else {
    const float theta = sqrtf(theta_sq), half = 0.5f * theta;
    imag = sinf(half) / theta;
    real = cosf(half);
  }
  const float x = __fmul_rn(imag, v[0]), y = __fmul_rn(imag, v[1]), z = __fmul_rn(imag, v[2]), w = real;
  const float tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const float twx = __fmul_rn(tx, w), twy = __fmul_rn(ty, w), twz = __fmul_rn(tz, w);
  const float txx = __fmul_rn(tx, x), txy = __fmul_rn(ty, x), txz = __fmul_rn(tz, x);
  const float tyy = __fmul_rn(ty, y), tyz = __fmul_rn(tz, y), tzz = __fmul_rn(tz, z);
  R[0] = 1 - __fadd_rn(tyy, tzz); R[1] = __fsub_rn(txy, twz); R[2] = __fadd_rn(txz, twy);
  R[3] = __fadd_rn(txy, twz); R[4] = 1 - __fadd_rn(txx, tzz); R[5] = __fsub_rn(tyz, twx);
  R[6] = __fsub_rn(txz, twy); R[7] = __fadd_rn(tyz, twx); R[8] = 1 - __fadd_rn(txx, tyy);
}
// p = one OSH_PREINT_FLOATS record; bg/ba = current DOUBLE estimates of the earlier keyframe's biases
__device__ inline void preint_deltas(const float* p, const double* bg, const double* ba, double* dR, double* dV, double* dP,
                                     double* dbg_out) {
  const float* pdR = p + 1; const float* pdV = p + 10; const float* pdP = p + 13;
  const float* JRg = p + 16; const float* JVg = p + 25; const float* JVa = p + 34; const float* JPg = p + 43; const float* JPa = p + 52;
  const float* b = p + 61;
  const float dbg[3] = {__fsub_rn((float)bg[0], b[3]), __fsub_rn((float)bg[1], b[4]), __fsub_rn((float)bg[2], b[5])};
  const float dba[3] = {__fsub_rn((float)ba[0], b[0]), __fsub_rn((float)ba[1], b[1]), __fsub_rn((float)ba[2], b[2])};
  float w[3], E[9], M[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) w[i] = dot3f(JRg + 3 * i, dbg);
  so3f_exp_matrix(w, E);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      M[i * 3 + j] = __fadd_rn(__fadd_rn(__fmul_rn(pdR[i * 3], E[j]), __fmul_rn(pdR[i * 3 + 1], E[3 + j])), __fmul_rn(pdR[i * 3 + 2], E[6 + j]));
  normalize_rotation_f(M);
#pragma unroll
  for (int i = 0; i < 9; ++i) dR[i] = (double)M[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dV[i] = (double)__fadd_rn(__fadd_rn(pdV[i], dot3f(JVg + 3 * i, dbg)), dot3f(JVa + 3 * i, dba));
    dP[i] = (double)__fadd_rn(__fadd_rn(pdP[i], dot3f(JPg + 3 * i, dbg)), dot3f(JPa + 3 * i, dba));
    if (dbg_out) dbg_out[i] = (double)dbg[i];
  }
}

}  // namespace imu
}  // namespace osh
