// ImuTypes.cc -- IMU::Calib and the FLOAT32 preintegration recursion IMU::Preintegrated::IntegrateNewMeasurement
// (reference src/ImuTypes.cc:86-108,147-237,265-283,398-410).  Written without Eigen: explicit float loops, the
// JacobiSVD-based NormalizeRotation replaced by the orthogonal polar factor (Newton iteration).
#include "ImuTypes.h"

#include <cmath>

namespace ORB_SLAM3 {
namespace IMU {

namespace {
typedef float M3[9];
inline void mul(const float* A, const float* B, float* C) {
  float T[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  for (int i = 0; i < 9; ++i) C[i] = T[i];
}
inline void hat(const float* v, float* W) { W[0] = 0; W[1] = -v[2]; W[2] = v[1]; W[3] = v[2]; W[4] = 0; W[5] = -v[0]; W[6] = -v[1]; W[7] = v[0]; W[8] = 0; }
void normalize_rotation(float* R) {   // src/ImuTypes.cc:34-37
  for (int it = 0; it < 12; ++it) {
    const float c00 = R[4] * R[8] - R[5] * R[7], c10 = R[5] * R[6] - R[3] * R[8], c20 = R[3] * R[7] - R[4] * R[6];
    const float id = 1.0f / (R[0] * c00 + R[1] * c10 + R[2] * c20);
    const float Ri[9] = {c00 * id, (R[2] * R[7] - R[1] * R[8]) * id, (R[1] * R[5] - R[2] * R[4]) * id,
                         c10 * id, (R[0] * R[8] - R[2] * R[6]) * id, (R[2] * R[3] - R[0] * R[5]) * id,
                         c20 * id, (R[1] * R[6] - R[0] * R[7]) * id, (R[0] * R[4] - R[1] * R[3]) * id};
    float d = 0, N[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { N[i * 3 + j] = 0.5f * (R[i * 3 + j] + Ri[j * 3 + i]); d = std::fmax(d, std::fabs(N[i * 3 + j] - R[i * 3 + j])); }
    for (int i = 0; i < 9; ++i) R[i] = N[i];
    if (d < 1e-7f) break;
  }
}
}  // namespace

void Calib::Set(const Sophus::SE3f& sophTbc, const float& ng, const float& na, const float& ngw, const float& naw) {
  mbIsSet = true;
  mTbc = sophTbc;
  mTcb = mTbc.inverse();
  for (int i = 0; i < 3; ++i) { Cov[i] = ng * ng; Cov[3 + i] = na * na; CovWalk[i] = ngw * ngw; CovWalk[3 + i] = naw * naw; }
}

Preintegrated::Preintegrated(const Bias& b_, const Calib& calib) {
  for (int i = 0; i < 6; ++i) { Nga[i] = calib.Cov[i]; NgaWalk[i] = calib.CovWalk[i]; }
  Initialize(b_);
}

void Preintegrated::Initialize(const Bias& b_) {
  dR = Eigen::Matrix3f(); dR(0, 0) = dR(1, 1) = dR(2, 2) = 1.f;
  dV = Eigen::Vector3f(); dP = Eigen::Vector3f();
  JRg = JVg = JVa = JPg = JPa = Eigen::Matrix3f();
  C = Eigen::Matrix<float, 15, 15>();
  for (int i = 0; i < 6; ++i) db[i] = 0.f;
  b = b_; bu = b_;
  dT = 0.0f;
}

void Preintegrated::IntegrateNewMeasurement(const Eigen::Vector3f& acceleration, const Eigen::Vector3f& angVel, const float& dt) {
  float A[81], B[54];
  for (int i = 0; i < 81; ++i) A[i] = (i % 10 == 0) ? 1.f : 0.f;
  for (int i = 0; i < 54; ++i) B[i] = 0.f;
  const float acc[3] = {acceleration(0) - b.bax, acceleration(1) - b.bay, acceleration(2) - b.baz};
  float* R = dR.v;
  float Racc[3];
  for (int i = 0; i < 3; ++i) Racc[i] = R[i * 3] * acc[0] + R[i * 3 + 1] * acc[1] + R[i * 3 + 2] * acc[2];
  // position first (uses the old velocity and rotation), then velocity (:192-194)
  for (int i = 0; i < 3; ++i) dP(i) = dP(i) + dV(i) * dt + 0.5f * Racc[i] * dt * dt;
  for (int i = 0; i < 3; ++i) dV(i) = dV(i) + Racc[i] * dt;
  M3 Wacc, T, T2;
  hat(acc, Wacc);
  // A / B blocks that rely on the non-updated rotation (:197-203)
  M3 Rdt; for (int i = 0; i < 9; ++i) Rdt[i] = R[i] * dt;
  M3 mRdt; for (int i = 0; i < 9; ++i) mRdt[i] = -Rdt[i];
  mul(mRdt, Wacc, T);                                             // -dR*dt*Wacc
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[(3 + i) * 9 + j] = T[i * 3 + j];
  M3 hRdt2; for (int i = 0; i < 9; ++i) hRdt2[i] = -0.5f * R[i] * dt * dt;
  mul(hRdt2, Wacc, T2);                                           // -0.5*dR*dt*dt*Wacc
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[(6 + i) * 9 + j] = T2[i * 3 + j];
  for (int i = 0; i < 3; ++i) A[(6 + i) * 9 + 3 + i] = dt;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { B[(3 + i) * 6 + 3 + j] = Rdt[i * 3 + j]; B[(6 + i) * 6 + 3 + j] = 0.5f * R[i * 3 + j] * dt * dt; }
  // bias Jacobians of position / velocity (:208-211)
  M3 TW, TWJ;
  for (int i = 0; i < 9; ++i) JPa.v[i] = JPa.v[i] + JVa.v[i] * dt - 0.5f * R[i] * dt * dt;
  for (int i = 0; i < 9; ++i) TW[i] = 0.5f * R[i] * dt * dt;
  mul(TW, Wacc, TW); mul(TW, JRg.v, TWJ);
  for (int i = 0; i < 9; ++i) JPg.v[i] = JPg.v[i] + JVg.v[i] * dt - TWJ[i];
  for (int i = 0; i < 9; ++i) JVa.v[i] = JVa.v[i] - Rdt[i];
  mul(Rdt, Wacc, TW); mul(TW, JRg.v, TWJ);
  for (int i = 0; i < 9; ++i) JVg.v[i] = JVg.v[i] - TWJ[i];
  // IntegratedRotation (:86-108)
  const float x = (angVel(0) - b.bwx) * dt, y = (angVel(1) - b.bwy) * dt, z = (angVel(2) - b.bwz) * dt;
  const float d2 = x * x + y * y + z * z, d = std::sqrt(d2);
  const float v[3] = {x, y, z};
  M3 W, W2, deltaR, rightJ;
  hat(v, W); mul(W, W, W2);
  for (int i = 0; i < 9; ++i) {
    const float I = (i % 4 == 0) ? 1.f : 0.f;
    if (d < 1e-4f) { deltaR[i] = I + W[i]; rightJ[i] = I; }
    else { deltaR[i] = I + W[i] * std::sin(d) / d + W2[i] * (1.0f - std::cos(d)) / d2; rightJ[i] = I - W[i] * (1.0f - std::cos(d)) / d2 + W2[i] * (d - std::sin(d)) / (d2 * d); }
  }
  mul(R, deltaR, R);
  normalize_rotation(R);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[i * 9 + j] = deltaR[j * 3 + i]; B[i * 6 + j] = rightJ[i * 3 + j] * dt; }
  // covariance (:229-230): C[0:9,0:9] = A C A^T + B Nga B^T ; C[9:15,9:15] += NgaWalk
  float AC[81], Cn[81];
  for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) { float s = 0; for (int k = 0; k < 9; ++k) s += A[i * 9 + k] * C(k, j); AC[i * 9 + j] = s; }
  for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) {
    float s = 0; for (int k = 0; k < 9; ++k) s += AC[i * 9 + k] * A[j * 9 + k];
    float t = 0; for (int k = 0; k < 6; ++k) t += (B[i * 6 + k] * Nga[k]) * B[j * 6 + k];
    Cn[i * 9 + j] = s + t;
  }
  for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) C(i, j) = Cn[i * 9 + j];
  for (int i = 0; i < 6; ++i) C(9 + i, 9 + i) += NgaWalk[i];
  // rotation Jacobian wrt the gyro bias (:233)
  M3 dRt; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dRt[i * 3 + j] = deltaR[j * 3 + i];
  mul(dRt, JRg.v, T);
  for (int i = 0; i < 9; ++i) JRg.v[i] = T[i] - rightJ[i] * dt;
  dT += dt;
}

void Preintegrated::SetNewBias(const Bias& bu_) {
  bu = bu_;
  db[0] = bu_.bwx - b.bwx; db[1] = bu_.bwy - b.bwy; db[2] = bu_.bwz - b.bwz;
  db[3] = bu_.bax - b.bax; db[4] = bu_.bay - b.bay; db[5] = bu_.baz - b.baz;
}

Bias Preintegrated::GetDeltaBias(const Bias& b_) {
  return Bias(b_.bax - b.bax, b_.bay - b.bay, b_.baz - b.baz, b_.bwx - b.bwx, b_.bwy - b.bwy, b_.bwz - b.bwz);
}

}  // namespace IMU
}  // namespace ORB_SLAM3
