// liba_edges.h -- device functions of the visual-inertial edges shared by LocalInertialBA (liba_device.hip) and
// PoseInertialOptimization (posei_device.hip): EdgeMono / EdgeStereo (and their OnlyPose forms, whose pose Jacobian is the same
// expression) on camera 0 or 1 of ImuCamPose, EdgeInertial.
#pragma once
#include "common.h"
#include "lba_math.h"
#include "liba_math.h"

namespace osh {

constexpr double kGrav = (double)9.81f;   // g << 0, 0, -IMU::GRAVITY_VALUE (a float constant)

struct LibaDesc {
  int N, NV, K, L, E, NL, n, max_iter;
  int pose_off, vel_off, pt_off, edge_off, link_off, lmoff_off, pel_off, peloff_off, lmpose_off;
  long long H_off;       // n*n doubles (H and S use the same offset in their own arrays)
  int b_off;             // n doubles
  double Rcb[9], tcb[3], tbc[3], cam[5];
  double kb8[4];   // KannalaBrandt8 k1..k4 (osh_liba_problem.kb8)
  int kb8_on;      // 1: mono edges project through KannalaBrandt8
  int rig_on;      // 1: fisheye stereo rig, OSH_EDGE_RIGHT edges are EdgeMono(1) on camera 1 of ImuCamPose (src/G2oTypes.cc:56-66)
  double Rrl[9], trl[3], Rcb1[9], tbc1[3], cam2[8];   // Trl; Rcb[1] = Rrl Rcb[0]; tbc[1] = -Rbc[1] tcb[1]; right camera fx fy cx cy k1..k4
  double huber_mono, huber_stereo, huber_inertial, lambda_init;
  int n_colours;         // inertial links are coloured so that the links of one colour share no keyframe (liba_device.hip)
  int il;                // layout of the reduced unknowns: 0 = [pose 6] x N then [velocity, gyro bias, accelerometer bias 9] x N (every
                         // LocalInertialBA window), 1 = [pose 6 | v bg ba 9] per keyframe (map-sized problems: with the keyframes in
                         // temporal order the reduced system is then BANDED -- landmarks and IMU links couple nearby keyframes only)
  int bw, bw_kf;         // il = 1: entries (r, c) with |r - c| > bw are structurally zero and never touched; bw_kf = the same in keyframes
};

struct VisEval { double r[3], chi2, Xc[3]; };

// EdgeMono / EdgeStereo computeError with the ImuCamPose camera pose (include/G2oTypes.h:355-361,438-444)
__device__ __forceinline__ void vis_residual(const LibaDesc& d, int kind, const double* pose, const double* X, const double* obs,
                                             double info, VisEval& o) {
  double u, v;
  if (kind == OSH_EDGE_RIGHT) {
    // EdgeMono(1): pCamera[1]->project(Rcw[1] Xw + tcw[1]) with Rcw[1] = Rrl Rcw[0], tcw[1] = Rrl tcw[0] + trl
    double R1[9], t1[3];
    imu::m3_mul(d.Rrl, pose, R1);
    imu::m3_vec(d.Rrl, pose + 9, t1);
    imu::m3_vec(R1, X, o.Xc);
    o.Xc[0] += t1[0] + d.trl[0]; o.Xc[1] += t1[1] + d.trl[1]; o.Xc[2] += t1[2] + d.trl[2];
    dev::kb8_project(d.cam2, d.cam2 + 4, o.Xc, u, v);
    o.r[0] = obs[0] - u; o.r[1] = obs[1] - v; o.r[2] = 0.0;
    o.chi2 = o.r[0] * (info * o.r[0]) + o.r[1] * (info * o.r[1]);
    return;
  }
  imu::m3_vec(pose, X, o.Xc);
  o.Xc[0] += pose[9]; o.Xc[1] += pose[10]; o.Xc[2] += pose[11];
  if (d.kb8_on) dev::kb8_project(d.cam, d.kb8, o.Xc, u, v);   // ImuCamPose::Project -> pCamera->project (src/G2oTypes.cc:166-171)
  else { u = d.cam[0] * o.Xc[0] / o.Xc[2] + d.cam[2]; v = d.cam[1] * o.Xc[1] / o.Xc[2] + d.cam[3]; }
  o.r[0] = obs[0] - u; o.r[1] = obs[1] - v; o.r[2] = 0.0;
  if (kind == OSH_EDGE_STEREO) {
    const double invZ = 1 / o.Xc[2];   // ProjectStereo keeps 1/z in double (src/G2oTypes.cc:181)
    o.r[2] = obs[2] - (u - d.cam[4] * invZ);
    o.chi2 = o.r[0] * (info * o.r[0]) + o.r[1] * (info * o.r[1]) + o.r[2] * (info * o.r[2]);
  } else {
    o.chi2 = o.r[0] * (info * o.r[0]) + o.r[1] * (info * o.r[1]);
  }
}
// linearizeOplus (src/G2oTypes.cc:349-373,397-427): JX 3x3, Jp 3x6 (row 2 zero for mono)
__device__ __forceinline__ void vis_jacobians(const LibaDesc& d, int kind, const double* pose, const double* Xc, double* JX, double* Jp) {
  double Xb[3];
  if (kind == OSH_EDGE_RIGHT) {
    // cam_idx = 1 (src/G2oTypes.cc:354-372): Xb = Rbc[1] Xc + tbc[1], JX = -projJac Rcw[1], Jp = projJac Rcb[1] SE3deriv(Xb)
    double pj1[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, R1[9], M1[9];
    imu::m3_tvec(d.Rcb1, Xc, Xb);
    Xb[0] += d.tbc1[0]; Xb[1] += d.tbc1[1]; Xb[2] += d.tbc1[2];
    dev::kb8_project_jac(d.cam2, d.cam2 + 4, Xc, pj1);
    imu::m3_mul(d.Rrl, pose, R1);
    imu::m3_mul(pj1, R1, M1);
#pragma unroll
    for (int i = 0; i < 9; ++i) JX[i] = -M1[i];
    const double x1 = Xb[0], y1 = Xb[1], z1 = Xb[2];
    const double D1[18] = {0, z1, -y1, 1, 0, 0, -z1, 0, x1, 0, 1, 0, y1, -x1, 0, 0, 0, 1};
    imu::m3_mul(pj1, d.Rcb1, M1);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) Jp[i * 6 + j] = M1[i * 3] * D1[j] + M1[i * 3 + 1] * D1[6 + j] + M1[i * 3 + 2] * D1[12 + j];
    return;
  }
  imu::m3_tvec(d.Rcb, Xc, Xb);   // Rbc = Rcb^T
  Xb[0] += d.tbc[0]; Xb[1] += d.tbc[1]; Xb[2] += d.tbc[2];
  double pj[9] = {d.cam[0] / Xc[2], 0, -d.cam[0] * Xc[0] / (Xc[2] * Xc[2]), 0, d.cam[1] / Xc[2], -d.cam[1] * Xc[1] / (Xc[2] * Xc[2]), 0, 0, 0};
  if (kind == OSH_EDGE_STEREO) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + d.cam[4] * (1.0 / (Xc[2] * Xc[2])); }
  if (d.kb8_on) dev::kb8_project_jac(d.cam, d.kb8, Xc, pj);   // pCamera->projectJac (src/G2oTypes.cc:359); a fisheye window is monocular
  double M[9];
  imu::m3_mul(pj, pose, M);
#pragma unroll
  for (int i = 0; i < 9; ++i) JX[i] = -M[i];
  const double x = Xb[0], y = Xb[1], z = Xb[2];
  const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
  imu::m3_mul(pj, d.Rcb, M);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) Jp[i * 6 + j] = M[i * 3] * D[j] + M[i * 3 + 1] * D[6 + j] + M[i * 3 + 2] * D[12 + j];
}

// EdgeInertial::computeError (src/G2oTypes.cc:513-533); P1/P2 = 24-double pose records, s1/s2 = v|bg|ba records
__device__ inline void inertial_residual(const float* rec, const double* P1, const double* s1, const double* P2, const double* s2, double* r) {
  const double dt = (double)rec[0];
  double dR[9], dV[3], dP[3], Rbw1[9], T[9], eR[9], t[3];
  imu::preint_deltas(rec, s1 + 3, s1 + 6, dR, dV, dP, nullptr);
  const double* Rwb1 = P1 + 12; const double* Rwb2 = P2 + 12;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Rbw1[i * 3 + j] = Rwb1[j * 3 + i];
  imu::m3_tmul(dR, Rbw1, T);
  imu::m3_mul(T, Rwb2, eR);
  imu::log_so3(eR, r);
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = s2[i] - s1[i] - (i == 2 ? -kGrav : 0.0) * dt;
  imu::m3_tvec(Rwb1, t, t);
#pragma unroll
  for (int i = 0; i < 3; ++i) r[3 + i] = t[i] - dV[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = P2[21 + i] - P1[21 + i] - s1[i] * dt - (i == 2 ? -kGrav : 0.0) * dt * dt / 2;
  imu::m3_tvec(Rwb1, t, t);
#pragma unroll
  for (int i = 0; i < 3; ++i) r[6 + i] = t[i] - dP[i];
}

// EdgeInertial::linearizeOplus (src/G2oTypes.cc:535-594) -> J [9][24], columns P1(6) V1(3) G1(3) A1(3) P2(6) V2(3)
__device__ inline void inertial_jacobian(const float* rec, const double* P1, const double* s1, const double* P2, const double* s2, double* J) {
  const double dt = (double)rec[0];
  double dR[9], dV[3], dP[3], dbg[3], Rbw1[9], T[9], eR[9], er[3], invJr[9], M[9], v[3], W[9];
  imu::preint_deltas(rec, s1 + 3, s1 + 6, dR, dV, dP, dbg);
  const double* Rwb1 = P1 + 12; const double* Rwb2 = P2 + 12;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbw1[i * 3 + j] = Rwb1[j * 3 + i];
  imu::m3_tmul(dR, Rbw1, T); imu::m3_mul(T, Rwb2, eR);
  imu::log_so3(eR, er);
  imu::inv_right_jac(er, invJr);
  for (int i = 0; i < 9 * 24; ++i) J[i] = 0.0;
#define OSH_PUT(r0, c0, Mx, sgn) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J[((r0) + i) * 24 + (c0) + j] = (sgn) * (Mx)[i * 3 + j]
  imu::m3_tmul(Rwb2, Rwb1, M); imu::m3_mul(invJr, M, M); OSH_PUT(0, 0, M, -1.0);
  for (int i = 0; i < 3; ++i) v[i] = s2[i] - s1[i] - (i == 2 ? -kGrav : 0.0) * dt;
  imu::m3_vec(Rbw1, v, v); imu::m3_hat(v, W); OSH_PUT(3, 0, W, 1.0);
  for (int i = 0; i < 3; ++i) v[i] = P2[21 + i] - P1[21 + i] - s1[i] * dt - 0.5 * (i == 2 ? -kGrav : 0.0) * dt * dt;
  imu::m3_vec(Rbw1, v, v); imu::m3_hat(v, W); OSH_PUT(6, 0, W, 1.0);
  { const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; OSH_PUT(6, 3, I, -1.0); }
  OSH_PUT(3, 6, Rbw1, -1.0);
  for (int i = 0; i < 9; ++i) M[i] = Rbw1[i] * dt;
  OSH_PUT(6, 6, M, -1.0);
  double JRg[9], JVg[9], JVa[9], JPg[9], JPa[9], rj[9], w[3], eRt[9];
  for (int i = 0; i < 9; ++i) { JRg[i] = rec[16 + i]; JVg[i] = rec[25 + i]; JVa[i] = rec[34 + i]; JPg[i] = rec[43 + i]; JPa[i] = rec[52 + i]; }
  imu::m3_vec(JRg, dbg, w); imu::right_jac(w, rj);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) eRt[i * 3 + j] = eR[j * 3 + i];
  imu::m3_mul(invJr, eRt, M); imu::m3_mul(M, rj, M); imu::m3_mul(M, JRg, M); OSH_PUT(0, 9, M, -1.0);
  OSH_PUT(3, 9, JVg, -1.0); OSH_PUT(6, 9, JPg, -1.0);
  OSH_PUT(3, 12, JVa, -1.0); OSH_PUT(6, 12, JPa, -1.0);
  OSH_PUT(0, 15, invJr, 1.0);
  imu::m3_mul(Rbw1, Rwb2, M); OSH_PUT(6, 18, M, 1.0);
  OSH_PUT(3, 21, Rbw1, 1.0);
#undef OSH_PUT
}

// computeError + linearizeOplus of EdgeInertial in one pass: the bias-corrected deltas, Rbw1 and the rotation error (and its log)
// are formed once for both (the residual and the Jacobian are always wanted together on the device)
__device__ inline void inertial_residual_jacobian(const float* rec, const double* P1, const double* s1, const double* P2, const double* s2, double* r, double* J) {
  const double dt = (double)rec[0];
  double dR[9], dV[3], dP[3], dbg[3], Rbw1[9], T[9], eR[9], er[3], invJr[9], M[9], v[3], W[9], t[3];
  imu::preint_deltas(rec, s1 + 3, s1 + 6, dR, dV, dP, dbg);
  const double* Rwb1 = P1 + 12; const double* Rwb2 = P2 + 12;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbw1[i * 3 + j] = Rwb1[j * 3 + i];
  imu::m3_tmul(dR, Rbw1, T); imu::m3_mul(T, Rwb2, eR);
  imu::log_so3(eR, er);
  // residual (src/G2oTypes.cc:513-533)
  for (int i = 0; i < 3; ++i) r[i] = er[i];
  for (int i = 0; i < 3; ++i) t[i] = s2[i] - s1[i] - (i == 2 ? -kGrav : 0.0) * dt;
  imu::m3_tvec(Rwb1, t, t);
  for (int i = 0; i < 3; ++i) r[3 + i] = t[i] - dV[i];
  for (int i = 0; i < 3; ++i) t[i] = P2[21 + i] - P1[21 + i] - s1[i] * dt - (i == 2 ? -kGrav : 0.0) * dt * dt / 2;
  imu::m3_tvec(Rwb1, t, t);
  for (int i = 0; i < 3; ++i) r[6 + i] = t[i] - dP[i];
  // Jacobian (src/G2oTypes.cc:535-594), columns P1(6) V1(3) G1(3) A1(3) P2(6) V2(3)
  imu::inv_right_jac(er, invJr);
  for (int i = 0; i < 9 * 24; ++i) J[i] = 0.0;
#define OSH_PUT(r0, c0, Mx, sgn) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J[((r0) + i) * 24 + (c0) + j] = (sgn) * (Mx)[i * 3 + j]
  imu::m3_tmul(Rwb2, Rwb1, M); imu::m3_mul(invJr, M, M); OSH_PUT(0, 0, M, -1.0);
  for (int i = 0; i < 3; ++i) v[i] = s2[i] - s1[i] - (i == 2 ? -kGrav : 0.0) * dt;
  imu::m3_vec(Rbw1, v, v); imu::m3_hat(v, W); OSH_PUT(3, 0, W, 1.0);
  for (int i = 0; i < 3; ++i) v[i] = P2[21 + i] - P1[21 + i] - s1[i] * dt - 0.5 * (i == 2 ? -kGrav : 0.0) * dt * dt;
  imu::m3_vec(Rbw1, v, v); imu::m3_hat(v, W); OSH_PUT(6, 0, W, 1.0);
  { const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; OSH_PUT(6, 3, I, -1.0); }
  OSH_PUT(3, 6, Rbw1, -1.0);
  for (int i = 0; i < 9; ++i) M[i] = Rbw1[i] * dt;
  OSH_PUT(6, 6, M, -1.0);
  double JRg[9], JVg[9], JVa[9], JPg[9], JPa[9], rj[9], w[3], eRt[9];
  for (int i = 0; i < 9; ++i) { JRg[i] = rec[16 + i]; JVg[i] = rec[25 + i]; JVa[i] = rec[34 + i]; JPg[i] = rec[43 + i]; JPa[i] = rec[52 + i]; }
  imu::m3_vec(JRg, dbg, w); imu::right_jac(w, rj);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) eRt[i * 3 + j] = eR[j * 3 + i];
  imu::m3_mul(invJr, eRt, M); imu::m3_mul(M, rj, M); imu::m3_mul(M, JRg, M); OSH_PUT(0, 9, M, -1.0);
  OSH_PUT(3, 9, JVg, -1.0); OSH_PUT(6, 9, JPg, -1.0);
  OSH_PUT(3, 12, JVa, -1.0); OSH_PUT(6, 12, JPa, -1.0);
  OSH_PUT(0, 15, invJr, 1.0);
  imu::m3_mul(Rbw1, Rwb2, M); OSH_PUT(6, 18, M, 1.0);
  OSH_PUT(3, 21, Rbw1, 1.0);
#undef OSH_PUT
}

}  // namespace osh
