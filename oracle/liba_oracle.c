/*
 * liba_oracle.c -- CPU restatement (plain C) of the g2o path behind
 * ORB_SLAM3::Optimizer::LocalInertialBA (src/Optimizer.cc:2387-2964).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see lba_oracle.c).  PARITY UNPINNED against a reference
 * binary (Eigen/OpenCV/Sophus-on-Eigen absent); pinned by oracle/liba_numpy.py (independent numpy
 * re-derivation with numeric Jacobians) and tests/test_oracle_liba.py.
 *
 * Restated pieces (paths relative to /root/reference, "g2o/" = Thirdparty/g2o/g2o/):
 *   ImuCamPose::Update / Project / ProjectStereo / isDepthPositive      src/G2oTypes.cc:170-220
 *   EdgeMono / EdgeStereo computeError + linearizeOplus                 include/G2oTypes.h:342-463, src/G2oTypes.cc:349-427
 *   EdgeInertial computeError / linearizeOplus                          src/G2oTypes.cc:513-594
 *   EdgeGyroRW / EdgeAccRW                                              include/G2oTypes.h:635-704
 *   ExpSO3 / LogSO3 / RightJacobianSO3 / InverseRightJacobianSO3        src/G2oTypes.cc:777-861
 *   IMU::Preintegrated::GetDeltaRotation/Velocity/Position/GetDeltaBias (float32)  src/ImuTypes.cc:277-309
 *   Sophus::SO3f::exp                                                   Thirdparty/Sophus/sophus/so3.hpp:583-619
 *   BaseMultiEdge::constructQuadraticForm                               g2o/core/base_multi_edge.hpp:171-222
 *   LM controller / block solver                                        as in lba_oracle.c
 * Eigen::JacobiSVD-based NormalizeRotation (U V^T) is restated as the orthogonal polar factor computed by
 * Newton iteration (same matrix up to rounding).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* ------------------------------------------------------------------ 3x3 helpers (row-major) */
static void m3_mul(const double* A, const double* B, double* C) {
  double T[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
  memcpy(C, T, sizeof(T));
}
static void m3_tmul(const double* A, const double* B, double* C) { /* A^T B */
  double T[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
  memcpy(C, T, sizeof(T));
}
static void m3_vec(const double* A, const double* v, double* o) {
  double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2], t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2], t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
  o[0] = t0; o[1] = t1; o[2] = t2;
}
static void m3_tvec(const double* A, const double* v, double* o) { /* A^T v */
  double t0 = A[0] * v[0] + A[3] * v[1] + A[6] * v[2], t1 = A[1] * v[0] + A[4] * v[1] + A[7] * v[2], t2 = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
  o[0] = t0; o[1] = t1; o[2] = t2;
}
static void m3_hat(const double* v, double* W) { W[0] = 0; W[1] = -v[2]; W[2] = v[1]; W[3] = v[2]; W[4] = 0; W[5] = -v[0]; W[6] = -v[1]; W[7] = v[0]; W[8] = 0; }
static void m3_inv(const double* m, double* inv) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c10 = m[5] * m[6] - m[3] * m[8], c20 = m[3] * m[7] - m[4] * m[6];
  const double id = 1.0 / (m[0] * c00 + m[1] * c10 + m[2] * c20);
  double T[9] = {c00 * id, (m[2] * m[7] - m[1] * m[8]) * id, (m[1] * m[5] - m[2] * m[4]) * id,
                 c10 * id, (m[0] * m[8] - m[2] * m[6]) * id, (m[2] * m[3] - m[0] * m[5]) * id,
                 c20 * id, (m[1] * m[6] - m[0] * m[7]) * id, (m[0] * m[4] - m[1] * m[3]) * id};
  memcpy(inv, T, sizeof(T));
}
/* NormalizeRotation: U V^T of the SVD == orthogonal polar factor (include/G2oTypes.h:67-71) */
static void normalize_rotation(double* R) {
  for (int it = 0; it < 12; ++it) {
    double Ri[9], d = 0;
    m3_inv(R, Ri);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
      const double n = 0.5 * (R[i * 3 + j] + Ri[j * 3 + i]);
      d = fmax(d, fabs(n - R[i * 3 + j]));
      R[i * 3 + j] = n;
    }
    if (d < 1e-16) break;
  }
}
static void normalize_rotation_f(float* R) { /* IMU::NormalizeRotation, src/ImuTypes.cc:34-37, float32 */
  for (int it = 0; it < 12; ++it) {
    const float c00 = R[4] * R[8] - R[5] * R[7], c10 = R[5] * R[6] - R[3] * R[8], c20 = R[3] * R[7] - R[4] * R[6];
    const float id = 1.0f / (R[0] * c00 + R[1] * c10 + R[2] * c20);
    const float Ri[9] = {c00 * id, (R[2] * R[7] - R[1] * R[8]) * id, (R[1] * R[5] - R[2] * R[4]) * id,
                         c10 * id, (R[0] * R[8] - R[2] * R[6]) * id, (R[2] * R[3] - R[0] * R[5]) * id,
                         c20 * id, (R[1] * R[6] - R[0] * R[7]) * id, (R[0] * R[4] - R[1] * R[3]) * id};
    float d = 0;
    float N[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
      N[i * 3 + j] = 0.5f * (R[i * 3 + j] + Ri[j * 3 + i]);
      d = fmaxf(d, fabsf(N[i * 3 + j] - R[i * 3 + j]));
    }
    memcpy(R, N, sizeof(N));
    if (d < 1e-7f) break;
  }
}

/* src/G2oTypes.cc:782-798 */
void oracle_exp_so3(const double* w, double* R) {
  const double x = w[0], y = w[1], z = w[2];
  const double d2 = x * x + y * y + z * z, d = sqrt(d2);
  double W[9], W2[9];
  m3_hat(w, W);
  m3_mul(W, W, W2);
  for (int i = 0; i < 9; ++i) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    if (d < 1e-5) R[i] = I + W[i] + 0.5 * W2[i];
    else R[i] = I + W[i] * sin(d) / d + W2[i] * (1.0 - cos(d)) / d2;
  }
  normalize_rotation(R);
}
/* src/G2oTypes.cc:800-813 (note the 0.5f literal and the un-scaled small-sin branch) */
void oracle_log_so3(const double* R, double* w) {
  const double tr = R[0] + R[4] + R[8];
  w[0] = (R[7] - R[5]) / 2; w[1] = (R[2] - R[6]) / 2; w[2] = (R[3] - R[1]) / 2;
  const double costheta = (tr - 1.0) * 0.5f;
  if (costheta > 1 || costheta < -1) return;
  const double theta = acos(costheta), s = sin(theta);
  if (fabs(s) < 1e-5) return;
  w[0] = theta * w[0] / s; w[1] = theta * w[1] / s; w[2] = theta * w[2] / s;
}
static void inv_right_jac(const double* v, double* J) { /* :820-833 */
  const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrt(d2);
  double W[9], W2[9];
  m3_hat(v, W); m3_mul(W, W, W2);
  for (int i = 0; i < 9; ++i) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    J[i] = (d < 1e-5) ? I : I + W[i] / 2 + W2[i] * (1.0 / d2 - (1.0 + cos(d)) / (2.0 * d * sin(d)));
  }
}
static void right_jac(const double* v, double* J) { /* :840-853 */
  const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = sqrt(d2);
  double W[9], W2[9];
  m3_hat(v, W); m3_mul(W, W, W2);
  for (int i = 0; i < 9; ++i) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    J[i] = (d < 1e-5) ? I : I - W[i] * (1.0 - cos(d)) / d2 + W2[i] * (d - sin(d)) / (d2 * d);
  }
}

/* ------------------------------------------------------------------ float32 preintegration getters */
typedef struct { float dT, dR[9], dV[3], dP[3], JRg[9], JVg[9], JVa[9], JPg[9], JPa[9], b[6]; } preint_t;
static void preint_load(const float* p, preint_t* o) {
  o->dT = p[0];
  memcpy(o->dR, p + 1, 36); memcpy(o->dV, p + 10, 12); memcpy(o->dP, p + 13, 12);
  memcpy(o->JRg, p + 16, 36); memcpy(o->JVg, p + 25, 36); memcpy(o->JVa, p + 34, 36);
  memcpy(o->JPg, p + 43, 36); memcpy(o->JPa, p + 52, 36); memcpy(o->b, p + 61, 24);
}
/* Sophus::SO3f::exp(v).matrix() in float32 (so3.hpp:583-619 + Eigen toRotationMatrix) */
static void so3f_exp_matrix(const float* v, float* R) {
  const float theta_sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  float imag, real;
  if (theta_sq < 1e-5f * 1e-5f) {
    const float theta_po4 = theta_sq * theta_sq;
    imag = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * theta_po4;
    real = 1.0f - (float)(1.0 / 8.0) * theta_sq + (float)(1.0 / 384.0) * theta_po4;
  } else {
    const float theta = sqrtf(theta_sq), half = 0.5f * theta;
    imag = sinf(half) / theta;
    real = cosf(half);
  }
  const float x = imag * v[0], y = imag * v[1], z = imag * v[2], w = real;
  const float tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
/* b1 = IMU::Bias built from the DOUBLE estimates -> rounded to float (src/G2oTypes.cc:522) */
static void preint_deltas(const preint_t* p, const double* bg, const double* ba, double* dR, double* dV, double* dP, double* dbg_out) {
  const float bwx = (float)bg[0], bwy = (float)bg[1], bwz = (float)bg[2], bax = (float)ba[0], bay = (float)ba[1], baz = (float)ba[2];
  const float dbg[3] = {bwx - p->b[3], bwy - p->b[4], bwz - p->b[5]};
  const float dba[3] = {bax - p->b[0], bay - p->b[1], baz - p->b[2]};
  float w[3], E[9], M[9];
  for (int i = 0; i < 3; ++i) w[i] = p->JRg[i * 3] * dbg[0] + p->JRg[i * 3 + 1] * dbg[1] + p->JRg[i * 3 + 2] * dbg[2];
  so3f_exp_matrix(w, E);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i * 3 + j] = p->dR[i * 3] * E[j] + p->dR[i * 3 + 1] * E[3 + j] + p->dR[i * 3 + 2] * E[6 + j];
  normalize_rotation_f(M);
  for (int i = 0; i < 9; ++i) dR[i] = (double)M[i];
  for (int i = 0; i < 3; ++i) {
    const float jv = p->JVg[i * 3] * dbg[0] + p->JVg[i * 3 + 1] * dbg[1] + p->JVg[i * 3 + 2] * dbg[2];
    const float ja = p->JVa[i * 3] * dba[0] + p->JVa[i * 3 + 1] * dba[1] + p->JVa[i * 3 + 2] * dba[2];
    dV[i] = (double)(p->dV[i] + jv + ja);
    const float pv = p->JPg[i * 3] * dbg[0] + p->JPg[i * 3 + 1] * dbg[1] + p->JPg[i * 3 + 2] * dbg[2];
    const float pa = p->JPa[i * 3] * dba[0] + p->JPa[i * 3 + 1] * dba[1] + p->JPa[i * 3 + 2] * dba[2];
    dP[i] = (double)(p->dP[i] + pv + pa);
    if (dbg_out) dbg_out[i] = (double)dbg[i];
  }
}

/* ------------------------------------------------------------------ state */
typedef struct {
  const osh_liba_problem* pr;
  int N, NV, K, L, E, NL, n;        /* n = 15 N */
  double *Rcw, *tcw, *Rwb, *twb;    /* K */
  double *vel, *bg, *ba;            /* NV */
  double* X;                        /* L */
  double* bak;                      /* backup of everything */
  size_t bak_len;
  double *err;                      /* E*3 visual errors */
  double *ierr;                     /* NL*9 inertial, then NL*3 gyro rw, NL*3 acc rw */
  double *H, *b;                    /* n*n dense (symmetric), n + 3L */
  double *Hll, *Hpl;                /* L*9, E*18 (per visual edge with optimisable pose) */
  int *col_off, *blk_edge;          /* CCS of Hpl by landmark: edges sorted by pose */
  int *slot;                        /* E: the edge whose Hpl block this edge adds to (itself; a right-camera edge sharing its
                                       (keyframe, landmark) pair with a left edge: that left edge -- ONE Hessian block) */
  int rig;                          /* fisheye stereo rig: camera 1 of ImuCamPose (src/G2oTypes.cc:56-66) */
  double *Rcw1, *tcw1;              /* K: Rcw[1], tcw[1] */
  double Rrl[9], trl[3], Rcb1[9], tcb1[3], tbc1[3];
  double *S, *bs, *coeff, *x, *Dinv, *tmp;
} istate;

static const double kG = (double)9.81f; /* g << 0,0,-IMU::GRAVITY_VALUE (float 9.81) */

static void cam_from_body(istate* s, int k) { /* ImuCamPose::Update tail, :212-218 */
  const osh_liba_problem* p = s->pr;
  double Rbw[9], tbw[3], t[3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbw[i * 3 + j] = s->Rwb[9 * k + j * 3 + i];
  m3_vec(Rbw, s->twb + 3 * k, tbw);
  tbw[0] = -tbw[0]; tbw[1] = -tbw[1]; tbw[2] = -tbw[2];
  m3_mul(p->Rcb, Rbw, s->Rcw + 9 * k);
  m3_vec(p->Rcb, tbw, t);
  for (int i = 0; i < 3; ++i) s->tcw[3 * k + i] = t[i] + p->tcb[i];
  if (s->rig) {   /* the loop over pCamera.size() (:214-218): Rcw[1] = Rcb[1] Rbw, tcw[1] = Rcb[1] tbw + tcb[1] */
    m3_mul(s->Rcb1, Rbw, s->Rcw1 + 9 * k);
    m3_vec(s->Rcb1, tbw, t);
    for (int i = 0; i < 3; ++i) s->tcw1[3 * k + i] = t[i] + s->tcb1[i];
  }
}

/* EdgeMono / EdgeStereo computeError (include/G2oTypes.h:355-361,438-444) */
static void vis_error(const istate* s, int e, double* r) {
  const osh_liba_problem* p = s->pr;
  const int k = p->edge_pose[e], l = p->edge_point[e];
  double Xc[3];
  if (p->edge_kind[e] == OSH_EDGE_RIGHT) {   /* EdgeMono(1): Project(Xw, 1) = pCamera[1]->project(Rcw[1] Xw + tcw[1]) (src/G2oTypes.cc:166-171) */
    double uv[2];
    m3_vec(s->Rcw1 + 9 * k, s->X + 3 * l, Xc);
    for (int i = 0; i < 3; ++i) Xc[i] += s->tcw1[3 * k + i];
    oracle_kb8_project(p->cam2, p->cam2 + 4, Xc, uv);
    r[0] = p->edge_obs[3 * e] - uv[0]; r[1] = p->edge_obs[3 * e + 1] - uv[1]; r[2] = 0;
    return;
  }
  m3_vec(s->Rcw + 9 * k, s->X + 3 * l, Xc);
  for (int i = 0; i < 3; ++i) Xc[i] += s->tcw[3 * k + i];
  double u = p->cam[0] * Xc[0] / Xc[2] + p->cam[2], v = p->cam[1] * Xc[1] / Xc[2] + p->cam[3];
  if (p->kb8) {   /* ImuCamPose::Project -> pCamera->project (src/G2oTypes.cc:166-171) through KannalaBrandt8 */
    double uv[2];
    oracle_kb8_project(p->cam, p->kb8, Xc, uv);
    u = uv[0]; v = uv[1];
  }
  r[0] = p->edge_obs[3 * e] - u;
  r[1] = p->edge_obs[3 * e + 1] - v;
  r[2] = 0;
  if (p->edge_kind[e] == OSH_EDGE_STEREO) {
    const double invZ = 1 / Xc[2];            /* ProjectStereo: double invZ (src/G2oTypes.cc:181) */
    r[2] = p->edge_obs[3 * e + 2] - (u - p->cam[4] * invZ);
  }
}
static double vis_chi2(const istate* s, int e) {
  const double w = s->pr->edge_info[e];
  const double* r = s->err + 3 * e;
  return (s->pr->edge_kind[e] != OSH_EDGE_STEREO) ? r[0] * (w * r[0]) + r[1] * (w * r[1]) : r[0] * (w * r[0]) + r[1] * (w * r[1]) + r[2] * (w * r[2]);
}
/* linearizeOplus of EdgeMono / EdgeStereo (src/G2oTypes.cc:349-373,397-427): JX 3x3, Jp 3x6 (row 2 zero for mono) */
static void vis_jac(const istate* s, int e, double* JX, double* Jp) {
  const osh_liba_problem* p = s->pr;
  const int k = p->edge_pose[e], l = p->edge_point[e];
  double Xc[3], Xb[3], Rbc[9];
  if (p->edge_kind[e] == OSH_EDGE_RIGHT) {   /* cam_idx = 1: Rcw[1], tcw[1], Rbc[1], tbc[1], Rcb[1], pCamera[1] (src/G2oTypes.cc:354-372) */
    double pj1[9], M1[9];
    m3_vec(s->Rcw1 + 9 * k, s->X + 3 * l, Xc);
    for (int i = 0; i < 3; ++i) Xc[i] += s->tcw1[3 * k + i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbc[i * 3 + j] = s->Rcb1[j * 3 + i];
    m3_vec(Rbc, Xc, Xb);
    for (int i = 0; i < 3; ++i) Xb[i] += s->tbc1[i];
    memset(pj1, 0, sizeof(pj1));
    oracle_kb8_project_jac(p->cam2, p->cam2 + 4, Xc, pj1);
    m3_mul(pj1, s->Rcw1 + 9 * k, M1);
    for (int i = 0; i < 9; ++i) JX[i] = -M1[i];
    const double x1 = Xb[0], y1 = Xb[1], z1 = Xb[2];
    const double D1[18] = {0, z1, -y1, 1, 0, 0, -z1, 0, x1, 0, 1, 0, y1, -x1, 0, 0, 0, 1};
    m3_mul(pj1, s->Rcb1, M1);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) Jp[i * 6 + j] = M1[i * 3] * D1[j] + M1[i * 3 + 1] * D1[6 + j] + M1[i * 3 + 2] * D1[12 + j];
    return;
  }
  m3_vec(s->Rcw + 9 * k, s->X + 3 * l, Xc);
  for (int i = 0; i < 3; ++i) Xc[i] += s->tcw[3 * k + i];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbc[i * 3 + j] = p->Rcb[j * 3 + i];
  m3_vec(Rbc, Xc, Xb);
  for (int i = 0; i < 3; ++i) Xb[i] += p->tbc[i];
  double pj[9];
  memset(pj, 0, sizeof(pj));
  pj[0] = p->cam[0] / Xc[2]; pj[2] = -p->cam[0] * Xc[0] / (Xc[2] * Xc[2]);
  pj[4] = p->cam[1] / Xc[2]; pj[5] = -p->cam[1] * Xc[1] / (Xc[2] * Xc[2]);
  if (p->edge_kind[e] == OSH_EDGE_STEREO) {
    pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + p->cam[4] * (1.0 / (Xc[2] * Xc[2]));
  }
  if (p->kb8) oracle_kb8_project_jac(p->cam, p->kb8, Xc, pj);   /* pCamera->projectJac (src/G2oTypes.cc:359); monocular window */
  double M[9];
  m3_mul(pj, s->Rcw + 9 * k, M);
  for (int i = 0; i < 9; ++i) JX[i] = -M[i];
  const double x = Xb[0], y = Xb[1], z = Xb[2];
  const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
  m3_mul(pj, p->Rcb, M);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) Jp[i * 6 + j] = M[i * 3] * D[j] + M[i * 3 + 1] * D[6 + j] + M[i * 3 + 2] * D[12 + j];
}

/* EdgeInertial::computeError (src/G2oTypes.cc:513-533) on explicit vertex values (1 = earlier, 2 = later state) */
static void inertial_error_core(const float* rec, const double* Rwb1, const double* twb1, const double* v1, const double* bg1, const double* ba1,
                                const double* Rwb2, const double* twb2, const double* v2, double* r) {
  preint_t pi;
  preint_load(rec, &pi);
  const double dt = (double)pi.dT;
  double dR[9], dV[3], dP[3];
  preint_deltas(&pi, bg1, ba1, dR, dV, dP, NULL);
  double T[9], eR[9], Rbw1[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbw1[i * 3 + j] = Rwb1[j * 3 + i];
  m3_tmul(dR, Rbw1, T); /* dR^T Rbw1 */
  m3_mul(T, Rwb2, eR);
  oracle_log_so3(eR, r);
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = v2[i] - v1[i] - (i == 2 ? -kG : 0.0) * dt;
  m3_tvec(Rwb1, t, t);
  for (int i = 0; i < 3; ++i) r[3 + i] = t[i] - dV[i];
  for (int i = 0; i < 3; ++i) t[i] = twb2[i] - twb1[i] - v1[i] * dt - (i == 2 ? -kG : 0.0) * dt * dt / 2;
  m3_tvec(Rwb1, t, t);
  for (int i = 0; i < 3; ++i) r[6 + i] = t[i] - dP[i];
}
/* EdgeInertial::linearizeOplus (src/G2oTypes.cc:535-594): J[v] 9 x dim(v) for v = P1(6) V1(3) G1(3) A1(3) P2(6) V2(3), packed 9 x 24 */
static void inertial_jac_core(const float* rec, const double* Rwb1, const double* twb1, const double* v1, const double* bg1, const double* ba1,
                              const double* Rwb2, const double* twb2, const double* v2, double* J) {
  preint_t pi;
  preint_load(rec, &pi);
  const double dt = (double)pi.dT;
  double dR[9], dV[3], dP[3], dbg[3];
  preint_deltas(&pi, bg1, ba1, dR, dV, dP, dbg);
  double Rbw1[9], eR[9], T[9], er[3], invJr[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbw1[i * 3 + j] = Rwb1[j * 3 + i];
  m3_tmul(dR, Rbw1, T); m3_mul(T, Rwb2, eR);
  oracle_log_so3(eR, er);
  inv_right_jac(er, invJr);
  memset(J, 0, sizeof(double) * 9 * 24);
#define JB(row, col) (J + (row) * 24 + (col))
#define PUT(r0, c0, M, sgn) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) JB(r0 + i, c0 + j)[0] = (sgn) * (M)[i * 3 + j]
  double M[9], v[3], W[9];
  /* Pose 1 */
  m3_tmul(Rwb2, Rwb1, M); m3_mul(invJr, M, M); PUT(0, 0, M, -1.0);
  for (int i = 0; i < 3; ++i) v[i] = v2[i] - v1[i] - (i == 2 ? -kG : 0.0) * dt;
  m3_vec(Rbw1, v, v); m3_hat(v, W); PUT(3, 0, W, 1.0);
  for (int i = 0; i < 3; ++i) v[i] = twb2[i] - twb1[i] - v1[i] * dt - 0.5 * (i == 2 ? -kG : 0.0) * dt * dt;
  m3_vec(Rbw1, v, v); m3_hat(v, W); PUT(6, 0, W, 1.0);
  { const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; PUT(6, 3, I, -1.0); }
  /* Velocity 1 */
  PUT(3, 6, Rbw1, -1.0);
  for (int i = 0; i < 9; ++i) M[i] = Rbw1[i] * dt;
  PUT(6, 6, M, -1.0);
  /* Gyro 1 */
  double JRg[9], JVg[9], JPg[9], JVa[9], JPa[9], rj[9], w[3];
  for (int i = 0; i < 9; ++i) { JRg[i] = pi.JRg[i]; JVg[i] = pi.JVg[i]; JPg[i] = pi.JPg[i]; JVa[i] = pi.JVa[i]; JPa[i] = pi.JPa[i]; }
  m3_vec(JRg, dbg, w); right_jac(w, rj);
  { double eRt[9]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) eRt[i * 3 + j] = eR[j * 3 + i];
    m3_mul(invJr, eRt, M); m3_mul(M, rj, M); m3_mul(M, JRg, M); PUT(0, 9, M, -1.0); }
  PUT(3, 9, JVg, -1.0); PUT(6, 9, JPg, -1.0);
  /* Acc 1 */
  PUT(3, 12, JVa, -1.0); PUT(6, 12, JPa, -1.0);
  /* Pose 2 */
  PUT(0, 15, invJr, 1.0);
  m3_mul(Rbw1, Rwb2, M); PUT(6, 18, M, 1.0);
  /* Velocity 2 */
  PUT(3, 21, Rbw1, 1.0);
#undef PUT
#undef JB
}
static void inertial_error(const istate* s, int l, double* r) {
  const osh_liba_problem* p = s->pr;
  const int a = p->link_prev[l], c = p->link_cur[l];
  const int ab = p->link_bias ? p->link_bias[l] : a;   /* the keyframe that stores the edge's bias vertices (bInit: one shared pair, src/Optimizer.cc:514-518) */
  inertial_error_core(p->link_preint + (size_t)l * OSH_PREINT_FLOATS, s->Rwb + 9 * a, s->twb + 3 * a, s->vel + 3 * a, s->bg + 3 * ab, s->ba + 3 * ab,
                      s->Rwb + 9 * c, s->twb + 3 * c, s->vel + 3 * c, r);
}
static void inertial_jac(const istate* s, int l, double* J) {
  const osh_liba_problem* p = s->pr;
  const int a = p->link_prev[l], c = p->link_cur[l];
  const int ab = p->link_bias ? p->link_bias[l] : a;
  inertial_jac_core(p->link_preint + (size_t)l * OSH_PREINT_FLOATS, s->Rwb + 9 * a, s->twb + 3 * a, s->vel + 3 * a, s->bg + 3 * ab, s->ba + 3 * ab,
                    s->Rwb + 9 * c, s->twb + 3 * c, s->vel + 3 * c, J);
}

static int lnk_off(const istate* s, int v /*0..5 vertex of the inertial edge*/, int l, int* dim) {
  /* reduced-state offset of edge vertex v, or -1 when fixed */
  const osh_liba_problem* p = s->pr;
  const int a = p->link_prev[l], c = p->link_cur[l], N = s->N;
  static const int dims[6] = {6, 3, 3, 3, 6, 3};
  *dim = dims[v];
  const int kf = (v == 2 || v == 3) ? (p->link_bias ? p->link_bias[l] : a) : (v < 4) ? a : c;
  if (kf >= N) return -1;
  switch (v) {
    case 0: case 4: return 6 * kf;
    case 1: case 5: return 6 * N + 9 * kf;
    case 2: return 6 * N + 9 * kf + 3;
    default: return 6 * N + 9 * kf + 6;
  }
}

static void compute_errors(istate* s) {
  for (int e = 0; e < s->E; ++e) vis_error(s, e, s->err + 3 * e);
  const osh_liba_problem* p = s->pr;
  for (int l = 0; l < s->NL; ++l) {
    inertial_error(s, l, s->ierr + 9 * l);
    const int a = p->link_prev[l], c = p->link_cur[l];
    for (int i = 0; i < 3; ++i) {
      s->ierr[9 * s->NL + 3 * l + i] = s->bg[3 * c + i] - s->bg[3 * a + i];          /* EdgeGyroRW */
      s->ierr[12 * s->NL + 3 * l + i] = s->ba[3 * c + i] - s->ba[3 * a + i];         /* EdgeAccRW  */
    }
  }
}
static double quad(const double* r, const double* Om, int d) {
  double c = 0;
  for (int i = 0; i < d; ++i) { double t = 0; for (int j = 0; j < d; ++j) t += Om[i * d + j] * r[j]; c += r[i] * t; }
  return c;
}
static double robust_chi2(const istate* s) {
  const osh_liba_problem* p = s->pr;
  double chi = 0, rho[3];
  /* inertial edges were added first (edge ids), then visual (src/Optimizer.cc:2648-2662, 2750...) */
  for (int l = 0; l < s->NL; ++l) {
    const double c = quad(s->ierr + 9 * l, p->link_info + 81 * (size_t)l, 9);
    if (p->link_robust[l]) { oracle_huber(c, p->huber_inertial, rho); chi += rho[0]; } else chi += c;
    chi += quad(s->ierr + 9 * s->NL + 3 * l, p->link_info_g + 9 * (size_t)l, 3);
    chi += quad(s->ierr + 12 * s->NL + 3 * l, p->link_info_a + 9 * (size_t)l, 3);
  }
  for (int e = 0; e < s->E; ++e) {
    oracle_huber(vis_chi2(s, e), p->edge_kind[e] != OSH_EDGE_STEREO ? p->huber_mono : p->huber_stereo, rho);
    chi += rho[0];
  }
  return chi;
}

/* H += Ja^T W Jb for two column ranges of a d-row Jacobian (both optimisable) */
static void add_block(istate* s, const double* J, int ld, int d, const double* W /*d x d*/, int ca, int da, int oa, int cb, int db, int ob) {
  for (int i = 0; i < da; ++i)
    for (int j = 0; j < db; ++j) {
      double acc = 0;
      for (int k = 0; k < d; ++k) {
        double t = 0;
        for (int m = 0; m < d; ++m) t += W[k * d + m] * J[m * ld + cb + j];
        acc += J[k * ld + ca + i] * t;
      }
      s->H[(size_t)(oa + i) * s->n + ob + j] += acc;
      if (oa != ob) s->H[(size_t)(ob + j) * s->n + oa + i] += acc;
    }
}

static void build_system(istate* s) {
  const osh_liba_problem* p = s->pr;
  const int n = s->n;
  memset(s->H, 0, sizeof(double) * (size_t)n * n);
  memset(s->b, 0, sizeof(double) * ((size_t)n + 3 * s->L));
  memset(s->Hll, 0, sizeof(double) * 9 * (size_t)s->L);
  memset(s->Hpl, 0, sizeof(double) * 18 * (size_t)s->E);
  /* inertial links (BaseMultiEdge::constructQuadraticForm, robust or not) */
  for (int l = 0; l < s->NL; ++l) {
    double J[9 * 24], W[81], wr[9];
    inertial_jac(s, l, J);
    const double* Om = p->link_info + 81 * (size_t)l;
    double rho1 = 1.0;
    if (p->link_robust[l]) { double rho[3]; oracle_huber(quad(s->ierr + 9 * l, Om, 9), p->huber_inertial, rho); rho1 = rho[1]; }
    for (int i = 0; i < 81; ++i) W[i] = rho1 * Om[i];
    for (int i = 0; i < 9; ++i) { double t = 0; for (int j = 0; j < 9; ++j) t += Om[i * 9 + j] * s->ierr[9 * l + j]; wr[i] = -t * rho1; }
    static const int col[6] = {0, 6, 9, 12, 15, 21};
    for (int va = 0; va < 6; ++va) {
      int da, oa = lnk_off(s, va, l, &da);
      if (oa < 0) continue;
      for (int i = 0; i < da; ++i) { double t = 0; for (int k = 0; k < 9; ++k) t += J[k * 24 + col[va] + i] * wr[k]; s->b[oa + i] += t; }
      for (int vb = va; vb < 6; ++vb) {
        int db, ob = lnk_off(s, vb, l, &db);
        if (ob < 0) continue;
        add_block(s, J, 24, 9, W, col[va], da, oa, col[vb], db, ob);
      }
    }
    /* random walks: r = b2 - b1, J = [-I, I], no robust kernel */
    for (int which = 0; which < 2; ++which) {
      const double* Og = (which == 0 ? p->link_info_g : p->link_info_a) + 9 * (size_t)l;
      const double* r = s->ierr + (which == 0 ? 9 : 12) * s->NL + 3 * l;
      const int a = p->link_prev[l], c = p->link_cur[l];
      const int o1 = (a < s->N) ? 6 * s->N + 9 * a + (which == 0 ? 3 : 6) : -1;
      const int o2 = 6 * s->N + 9 * c + (which == 0 ? 3 : 6);
      double Or[3];
      for (int i = 0; i < 3; ++i) Or[i] = -(Og[i * 3] * r[0] + Og[i * 3 + 1] * r[1] + Og[i * 3 + 2] * r[2]);
      for (int i = 0; i < 3; ++i) {
        if (o1 >= 0) s->b[o1 + i] += -Or[i];
        s->b[o2 + i] += Or[i];
        for (int j = 0; j < 3; ++j) {
          if (o1 >= 0) {
            s->H[(size_t)(o1 + i) * n + o1 + j] += Og[i * 3 + j];
            s->H[(size_t)(o1 + i) * n + o2 + j] += -Og[i * 3 + j];
            s->H[(size_t)(o2 + j) * n + o1 + i] += -Og[i * 3 + j];
          }
          s->H[(size_t)(o2 + i) * n + o2 + j] += Og[i * 3 + j];
        }
      }
    }
  }
  /* visual edges (binary, robust) */
  double* bl = s->b + n;
  for (int e = 0; e < s->E; ++e) {
    const int k = p->edge_pose[e], l = p->edge_point[e];
    double A[9], B[18], rho[3];
    vis_jac(s, e, A, B);
    const double w = p->edge_info[e];
    const double* r = s->err + 3 * e;
    oracle_huber(vis_chi2(s, e), p->edge_kind[e] != OSH_EDGE_STEREO ? p->huber_mono : p->huber_stereo, rho);
    const double ww = rho[1] * w;
    const double wr[3] = {-(w * r[0]) * rho[1], -(w * r[1]) * rho[1], -(w * r[2]) * rho[1]};
    for (int i = 0; i < 3; ++i) {
      bl[3 * l + i] += A[i] * wr[0] + A[3 + i] * wr[1] + A[6 + i] * wr[2];
      for (int j = 0; j < 3; ++j) s->Hll[9 * l + i * 3 + j] += (A[i] * ww) * A[j] + (A[3 + i] * ww) * A[3 + j] + (A[6 + i] * ww) * A[6 + j];
    }
    if (k < s->N) {
      for (int i = 0; i < 6; ++i) {
        s->b[6 * k + i] += B[i] * wr[0] + B[6 + i] * wr[1] + B[12 + i] * wr[2];
        for (int j = 0; j < 6; ++j)
          s->H[(size_t)(6 * k + i) * n + 6 * k + j] += (B[i] * ww) * B[j] + (B[6 + i] * ww) * B[6 + j] + (B[12 + i] * ww) * B[12 + j];
        for (int j = 0; j < 3; ++j)
          s->Hpl[18 * (size_t)s->slot[e] + i * 3 + j] += (B[i] * ww) * A[j] + (B[6 + i] * ww) * A[3 + j] + (B[12 + i] * ww) * A[6 + j];
      }
    }
  }
}

static int block_solve(istate* s, double lambda) {
  const int n = s->n;
  const osh_liba_problem* p = s->pr;
  memcpy(s->S, s->H, sizeof(double) * (size_t)n * n);
  for (int i = 0; i < n; ++i) s->S[(size_t)i * n + i] += lambda;     /* setLambda on every pose-side diagonal */
  memset(s->coeff, 0, sizeof(double) * n);
  const double* bl = s->b + n;
  for (int j = 0; j < s->L; ++j) {
    double D[9];
    memcpy(D, s->Hll + 9 * j, sizeof(D));
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    double* Dinv = s->Dinv + 9 * j;
    m3_inv(D, Dinv);
    double db[3];
    m3_vec(Dinv, bl + 3 * j, db);
    for (int a = s->col_off[j]; a < s->col_off[j + 1]; ++a) {
      const int ea = s->blk_edge[a], i1 = p->edge_pose[ea];
      const double* Bi = s->Hpl + 18 * (size_t)ea;
      double BD[18];
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 3; ++c) BD[r * 3 + c] = Bi[r * 3] * Dinv[c] + Bi[r * 3 + 1] * Dinv[3 + c] + Bi[r * 3 + 2] * Dinv[6 + c];
      for (int r = 0; r < 6; ++r) s->coeff[6 * i1 + r] += Bi[r * 3] * db[0] + Bi[r * 3 + 1] * db[1] + Bi[r * 3 + 2] * db[2];
      for (int a2 = a; a2 < s->col_off[j + 1]; ++a2) {
        const int eb = s->blk_edge[a2], i2 = p->edge_pose[eb];
        const double* Bj = s->Hpl + 18 * (size_t)eb;
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) {
          const double v = BD[r * 3] * Bj[c * 3] + BD[r * 3 + 1] * Bj[c * 3 + 1] + BD[r * 3 + 2] * Bj[c * 3 + 2];
          s->S[(size_t)(6 * i1 + r) * n + 6 * i2 + c] -= v;
        }
      }
    }
  }
  for (int i = 0; i < n; ++i) s->bs[i] = s->b[i] - s->coeff[i];
  if (!oracle_ldlt_solve(n, s->S, s->bs, s->x, s->tmp)) return 0;
  double* xl = s->x + n;
  for (int j = 0; j < s->L; ++j) {
    double cl[3] = {bl[3 * j], bl[3 * j + 1], bl[3 * j + 2]};
    for (int a = s->col_off[j]; a < s->col_off[j + 1]; ++a) {
      const int ea = s->blk_edge[a];
      const double* Bi = s->Hpl + 18 * (size_t)ea;
      const double* xp = s->x + 6 * p->edge_pose[ea];
      for (int c = 0; c < 3; ++c) { double acc = 0; for (int r = 0; r < 6; ++r) acc += Bi[r * 3 + c] * (-xp[r]); cl[c] += acc; }
    }
    m3_vec(s->Dinv + 9 * j, cl, xl + 3 * j);
  }
  return 1;
}

static void apply_update(istate* s) {
  const int N = s->N;
  for (int k = 0; k < N; ++k) {   /* VertexPose::oplusImpl -> ImuCamPose::Update (src/G2oTypes.cc:187-220) */
    const double* pu = s->x + 6 * k;
    double t[3], E[9];
    m3_vec(s->Rwb + 9 * k, pu + 3, t);
    for (int i = 0; i < 3; ++i) s->twb[3 * k + i] += t[i];
    oracle_exp_so3(pu, E);
    m3_mul(s->Rwb + 9 * k, E, s->Rwb + 9 * k);
    cam_from_body(s, k);
  }
  for (int k = 0; k < N; ++k)
    for (int i = 0; i < 3; ++i) {
      s->vel[3 * k + i] += s->x[6 * N + 9 * k + i];
      s->bg[3 * k + i] += s->x[6 * N + 9 * k + 3 + i];
      s->ba[3 * k + i] += s->x[6 * N + 9 * k + 6 + i];
    }
  for (int i = 0; i < 3 * s->L; ++i) s->X[i] += s->x[s->n + i];
}

static void state_pack(istate* s, double* dst) {
  size_t o = 0;
#define CP(ptr, cnt) memcpy(dst + o, ptr, sizeof(double) * (cnt)); o += (cnt)
  CP(s->Rcw, 9 * (size_t)s->K); CP(s->tcw, 3 * (size_t)s->K); CP(s->Rwb, 9 * (size_t)s->K); CP(s->twb, 3 * (size_t)s->K);
  CP(s->vel, 3 * (size_t)s->NV); CP(s->bg, 3 * (size_t)s->NV); CP(s->ba, 3 * (size_t)s->NV); CP(s->X, 3 * (size_t)s->L);
  if (s->rig) { CP(s->Rcw1, 9 * (size_t)s->K); CP(s->tcw1, 3 * (size_t)s->K); }
#undef CP
}
static void state_unpack(istate* s, const double* src) {
  size_t o = 0;
#define CP(ptr, cnt) memcpy(ptr, src + o, sizeof(double) * (cnt)); o += (cnt)
  CP(s->Rcw, 9 * (size_t)s->K); CP(s->tcw, 3 * (size_t)s->K); CP(s->Rwb, 9 * (size_t)s->K); CP(s->twb, 3 * (size_t)s->K);
  CP(s->vel, 3 * (size_t)s->NV); CP(s->bg, 3 * (size_t)s->NV); CP(s->ba, 3 * (size_t)s->NV); CP(s->X, 3 * (size_t)s->L);
  if (s->rig) { CP(s->Rcw1, 9 * (size_t)s->K); CP(s->tcw1, 3 * (size_t)s->K); }
#undef CP
}

static void* zalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }

static int istate_init(istate* s, const osh_liba_problem* p) {
  memset(s, 0, sizeof(*s));
  s->pr = p;
  s->N = p->n_opt; s->NV = p->n_opt + p->n_fixed_imu; s->K = s->NV + p->n_fixed; s->L = p->n_points; s->E = p->n_edges; s->NL = p->n_links;
  s->n = 15 * s->N;
  const size_t n = s->n, K = s->K, NV = s->NV, L = s->L, E = s->E, NL = s->NL;
  s->Rcw = zalloc(9 * K, 8); s->tcw = zalloc(3 * K, 8); s->Rwb = zalloc(9 * K, 8); s->twb = zalloc(3 * K, 8);
  s->vel = zalloc(3 * NV, 8); s->bg = zalloc(3 * NV, 8); s->ba = zalloc(3 * NV, 8); s->X = zalloc(3 * L, 8);
  s->rig = (p->kb8 && p->cam2 && p->trl) ? 1 : 0;
  s->bak_len = 24 * K + 9 * NV + 3 * L + (s->rig ? 12 * K : 0); s->bak = zalloc(s->bak_len, 8);
  s->err = zalloc(3 * E, 8); s->ierr = zalloc(15 * NL, 8);
  s->H = zalloc(n * n, 8); s->b = zalloc(n + 3 * L, 8); s->Hll = zalloc(9 * L, 8); s->Hpl = zalloc(18 * E, 8);
  s->S = zalloc(n * n, 8); s->bs = zalloc(n, 8); s->coeff = zalloc(n, 8); s->x = zalloc(n + 3 * L, 8); s->Dinv = zalloc(9 * L, 8);
  s->tmp = zalloc(n, 8);
  memcpy(s->Rcw, p->pose_Rcw, 72 * K); memcpy(s->tcw, p->pose_tcw, 24 * K); memcpy(s->Rwb, p->pose_Rwb, 72 * K); memcpy(s->twb, p->pose_twb, 24 * K);
  memcpy(s->vel, p->vel, 24 * NV); memcpy(s->bg, p->bias_g, 24 * NV); memcpy(s->ba, p->bias_a, 24 * NV); memcpy(s->X, p->points, 24 * L);
  s->Rcw1 = zalloc(9 * K, 8); s->tcw1 = zalloc(3 * K, 8);
  for (size_t e = 0; e < E; ++e) if (p->edge_kind[e] == OSH_EDGE_RIGHT && !s->rig) return 0;
  if (s->rig) {   /* ImuCamPose(KeyFrame*) for camera 1 (src/G2oTypes.cc:56-66) */
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) s->Rrl[i * 3 + j] = p->trl[i * 4 + j]; s->trl[i] = p->trl[i * 4 + 3]; }
    double t[3], Rbc1[9];
    m3_mul(s->Rrl, p->Rcb, s->Rcb1);
    m3_vec(s->Rrl, p->tcb, t);
    for (int i = 0; i < 3; ++i) s->tcb1[i] = t[i] + s->trl[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbc1[i * 3 + j] = s->Rcb1[j * 3 + i];
    m3_vec(Rbc1, s->tcb1, t);
    for (int i = 0; i < 3; ++i) s->tbc1[i] = -t[i];
    for (size_t k = 0; k < K; ++k) {
      m3_mul(s->Rrl, s->Rcw + 9 * k, s->Rcw1 + 9 * k);
      m3_vec(s->Rrl, s->tcw + 3 * k, t);
      for (int i = 0; i < 3; ++i) s->tcw1[3 * k + i] = t[i] + s->trl[i];
    }
  }
  /* Hessian block of every edge: the right-camera edge of a (keyframe, landmark) pair that also has a left edge adds to that
   * left edge's block (the later edge of the pair in the caller's order adds to the earlier one) */
  s->slot = zalloc(E, sizeof(int));
  {
    int* first = zalloc((size_t)L * (s->N ? s->N : 1), sizeof(int));
    for (size_t q = 0; q < (size_t)L * s->N; ++q) first[q] = -1;
    for (size_t e = 0; e < E; ++e) {
      s->slot[e] = (int)e;
      if (p->edge_pose[e] >= s->N) continue;
      int* f = &first[(size_t)p->edge_point[e] * s->N + p->edge_pose[e]];
      if (*f < 0) *f = (int)e; else s->slot[e] = *f;
    }
    free(first);
  }
  /* CCS of the optimisable-pose visual blocks per landmark, rows ascending */
  s->col_off = zalloc(L + 1, sizeof(int)); s->blk_edge = zalloc(E, sizeof(int));
  for (size_t e = 0; e < E; ++e) if (p->edge_pose[e] < s->N && s->slot[e] == (int)e) s->col_off[p->edge_point[e] + 1]++;
  for (size_t j = 0; j < L; ++j) s->col_off[j + 1] += s->col_off[j];
  int* fill = zalloc(L, sizeof(int));
  for (size_t j = 0; j < L; ++j) fill[j] = s->col_off[j];
  for (size_t e = 0; e < E; ++e) if (p->edge_pose[e] < s->N && s->slot[e] == (int)e) s->blk_edge[fill[p->edge_point[e]]++] = (int)e;
  for (size_t j = 0; j < L; ++j)
    for (int a = s->col_off[j] + 1; a < s->col_off[j + 1]; ++a) {
      int v = s->blk_edge[a], b = a - 1;
      while (b >= s->col_off[j] && p->edge_pose[s->blk_edge[b]] > p->edge_pose[v]) { s->blk_edge[b + 1] = s->blk_edge[b]; --b; }
      s->blk_edge[b + 1] = v;
    }
  free(fill);
  return 1;
}
static void istate_free(istate* s) {
  free(s->Rcw); free(s->tcw); free(s->Rwb); free(s->twb); free(s->vel); free(s->bg); free(s->ba); free(s->X); free(s->bak);
  free(s->err); free(s->ierr); free(s->H); free(s->b); free(s->Hll); free(s->Hpl); free(s->S); free(s->bs); free(s->coeff);
  free(s->x); free(s->Dinv); free(s->tmp); free(s->col_off); free(s->blk_edge); free(s->slot); free(s->Rcw1); free(s->tcw1);
}

/* Debug / parity aids */
int oracle_liba_linearize(const osh_liba_problem* p, double* H, double* b, double* Hll, double* Hpl, double* chi2) {
  istate s;
  istate_init(&s, p);
  compute_errors(&s);
  if (chi2) *chi2 = robust_chi2(&s);
  build_system(&s);
  if (H) memcpy(H, s.H, sizeof(double) * (size_t)s.n * s.n);
  if (b) memcpy(b, s.b, sizeof(double) * ((size_t)s.n + 3 * s.L));
  if (Hll) memcpy(Hll, s.Hll, sizeof(double) * 9 * (size_t)s.L);
  if (Hpl) memcpy(Hpl, s.Hpl, sizeof(double) * 18 * (size_t)s.E);
  istate_free(&s);
  return OSH_OK;
}
int oracle_liba_inertial_edge(const osh_liba_problem* p, int link, double* r9, double* J9x24) {
  istate s;
  istate_init(&s, p);
  if (r9) inertial_error(&s, link, r9);
  if (J9x24) inertial_jac(&s, link, J9x24);
  istate_free(&s);
  return OSH_OK;
}

/* optimizer.computeActiveErrors(); err = activeRobustChi2(); optimize(opt_it); err_end = activeRobustChi2() (:2843-2848) */
int oracle_liba_solve(const osh_liba_problem* p, osh_liba_result* res) {
  istate s;
  if (!istate_init(&s, p)) return OSH_ERR_INVALID;
  const int nall = s.n + 3 * s.L;
  double lambda = -1., ni = 2.;
  int nBad = 0, cj = 0, trials_total = 0, ok = 1;
  res->n_trace = 0;
  compute_errors(&s);
  res->chi2_initial = robust_chi2(&s);
  for (int it = 0; it < p->max_iterations && ok; ++it) {   /* no stop flag during optimize (attached after, :2849-2850) */
    compute_errors(&s);
    double currentChi = robust_chi2(&s), tempChi = currentChi;
    const double iniChi = currentChi;
    build_system(&s);
    if (it == 0) {
      if (p->lambda_init > 0) lambda = p->lambda_init;
      else { double m = 0; for (int i = 0; i < s.n; ++i) m = fmax(m, fabs(s.H[(size_t)i * s.n + i])); for (int j = 0; j < s.L; ++j) for (int d = 0; d < 3; ++d) m = fmax(m, fabs(s.Hll[9 * j + 4 * d])); lambda = 1e-5 * m; }
      ni = 2; nBad = 0;
    }
    double rho = 0; int qmax = 0;
    do {
      state_pack(&s, s.bak);
      const int ok2 = block_solve(&s, lambda);
      apply_update(&s);
      compute_errors(&s);
      tempChi = robust_chi2(&s);
      if (!ok2) tempChi = DBL_MAX;
      rho = currentChi - tempChi;
      double scale = 0;
      for (int j = 0; j < nall; ++j) scale += s.x[j] * (lambda * s.x[j] + s.b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        lambda *= fmax(1. / 3., alpha);
        ni = 2; currentChi = tempChi;
      } else {
        lambda *= ni; ni *= 2;
        state_unpack(&s, s.bak);
      }
      qmax++; trials_total++;
    } while (rho < 0 && qmax < 10);
    ++cj;
    if (res->n_trace < OSH_LBA_MAX_TRACE) { res->chi2_trace[res->n_trace] = currentChi; res->lambda_trace[res->n_trace] = lambda; res->trials_trace[res->n_trace] = qmax; res->n_trace++; }
    if (qmax == 10 || rho == 0) { ok = 0; continue; }
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) { ok = 0; continue; }
  }
  res->iterations = cj; res->trials = trials_total; res->status = OSH_OK;
  /* err_end: activeRobustChi2() on the stored (possibly stale) errors */
  res->chi2_final = robust_chi2(&s);
  const int N = s.N;
  if (res->pose_Rcw) memcpy(res->pose_Rcw, s.Rcw, 72 * (size_t)N);
  if (res->pose_tcw) memcpy(res->pose_tcw, s.tcw, 24 * (size_t)N);
  if (res->pose_Rwb) memcpy(res->pose_Rwb, s.Rwb, 72 * (size_t)N);
  if (res->pose_twb) memcpy(res->pose_twb, s.twb, 24 * (size_t)N);
  if (res->vel) memcpy(res->vel, s.vel, 24 * (size_t)N);
  if (res->bias_g) memcpy(res->bias_g, s.bg, 24 * (size_t)N);
  if (res->bias_a) memcpy(res->bias_a, s.ba, 24 * (size_t)N);
  if (res->points) memcpy(res->points, s.X, 24 * (size_t)s.L);
  if (res->edge_chi2) for (int e = 0; e < s.E; ++e) res->edge_chi2[e] = vis_chi2(&s, e);
  if (res->edge_depth_pos)
    for (int e = 0; e < s.E; ++e) {   /* ImuCamPose::isDepthPositive (src/G2oTypes.cc:185-188) */
      const int k = p->edge_pose[e];
      const int right = p->edge_kind[e] == OSH_EDGE_RIGHT;   /* isDepthPositive(Xw, cam_idx) */
      const double* R = (right ? s.Rcw1 : s.Rcw) + 9 * k; const double* X = s.X + 3 * p->edge_point[e];
      res->edge_depth_pos[e] = (R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + (right ? s.tcw1 : s.tcw)[3 * k + 2]) > 0.0;
    }
  istate_free(&s);
  return OSH_OK;
}

/* ==================================================================================================================
 * Optimizer::PoseInertialOptimizationLastKeyFrame (src/Optimizer.cc:4499-4899, mode 0) and
 * Optimizer::PoseInertialOptimizationLastFrame (src/Optimizer.cc:4901-5299, mode 1): Gauss-Newton
 * (g2o/core/optimization_algorithm_gauss_newton.cpp:49-90) with the dense LDL^T of LinearSolverDense
 * (g2o/solvers/linear_solver_dense.h:60-118) on the frame's pose / velocity / biases.
 * ================================================================================================================== */
typedef struct {
  const osh_posei_problem* p;
  int rig;
  double Rcw[9], tcw[3], Rwb[9], twb[3], v[3], bg[3], ba[3], Rcw1[9], tcw1[3];   /* current frame, camera 0 and camera 1 */
  double pRwb[9], ptwb[3], pv[3], pbg[3], pba[3];                                 /* previous state */
  double Rrl[9], trl[3], Rcb1[9], tcb1[3], tbc1[3];
} pstate;

static void pi_cams_from_body(pstate* s) {   /* ImuCamPose::Update tail (src/G2oTypes.cc:212-218) */
  const osh_posei_problem* p = s->p;
  double Rbw[9], tbw[3], t[3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbw[i * 3 + j] = s->Rwb[j * 3 + i];
  m3_vec(Rbw, s->twb, tbw);
  for (int i = 0; i < 3; ++i) tbw[i] = -tbw[i];
  m3_mul(p->Rcb, Rbw, s->Rcw);
  m3_vec(p->Rcb, tbw, t);
  for (int i = 0; i < 3; ++i) s->tcw[i] = t[i] + p->tcb[i];
  if (s->rig) {
    m3_mul(s->Rcb1, Rbw, s->Rcw1);
    m3_vec(s->Rcb1, tbw, t);
    for (int i = 0; i < 3; ++i) s->tcw1[i] = t[i] + s->tcb1[i];
  }
}
static void pi_cam_point(const pstate* s, int e, double* Xc) {
  const osh_posei_problem* p = s->p;
  const int right = p->edge_kind[e] == OSH_EDGE_RIGHT;
  m3_vec(right ? s->Rcw1 : s->Rcw, p->points + 3 * (size_t)e, Xc);
  for (int i = 0; i < 3; ++i) Xc[i] += (right ? s->tcw1 : s->tcw)[i];
}
/* EdgeMonoOnlyPose / EdgeStereoOnlyPose computeError (include/G2oTypes.h:399-405,475-481; ImuCamPose::Project / ProjectStereo) */
static void pi_vis_error(const pstate* s, int e, double* r) {
  const osh_posei_problem* p = s->p;
  const int kind = p->edge_kind[e];
  double Xc[3], uv[2];
  pi_cam_point(s, e, Xc);
  if (kind == OSH_EDGE_RIGHT) oracle_kb8_project(p->cam2, p->cam2 + 4, Xc, uv);
  else if (p->kb8) oracle_kb8_project(p->cam, p->kb8, Xc, uv);
  else { uv[0] = p->cam[0] * Xc[0] / Xc[2] + p->cam[2]; uv[1] = p->cam[1] * Xc[1] / Xc[2] + p->cam[3]; }
  r[0] = p->edge_obs[3 * (size_t)e] - uv[0];
  r[1] = p->edge_obs[3 * (size_t)e + 1] - uv[1];
  r[2] = 0;
  if (kind == OSH_EDGE_STEREO) { const double invZ = 1 / Xc[2]; r[2] = p->edge_obs[3 * (size_t)e + 2] - (uv[0] - p->cam[4] * invZ); }
}
static double pi_vis_chi2(const pstate* s, int e, const double* r) {
  const double w = s->p->edge_info[e];
  return s->p->edge_kind[e] == OSH_EDGE_STEREO ? r[0] * (w * r[0]) + r[1] * (w * r[1]) + r[2] * (w * r[2]) : r[0] * (w * r[0]) + r[1] * (w * r[1]);
}
/* linearizeOplus of the two unary edges (src/G2oTypes.cc:375-395,429-455): Jp 3x6 (row 2 zero for the mono kinds) */
static void pi_vis_jac(const pstate* s, int e, double* Jp) {
  const osh_posei_problem* p = s->p;
  const int kind = p->edge_kind[e], right = kind == OSH_EDGE_RIGHT;
  double Xc[3], Xb[3], Rbc[9], pj[9], M[9];
  pi_cam_point(s, e, Xc);
  const double* Rcb = right ? s->Rcb1 : p->Rcb;
  const double* tbc = right ? s->tbc1 : p->tbc;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbc[i * 3 + j] = Rcb[j * 3 + i];
  m3_vec(Rbc, Xc, Xb);
  for (int i = 0; i < 3; ++i) Xb[i] += tbc[i];
  memset(pj, 0, sizeof(pj));
  if (right) oracle_kb8_project_jac(p->cam2, p->cam2 + 4, Xc, pj);
  else if (p->kb8) oracle_kb8_project_jac(p->cam, p->kb8, Xc, pj);
  else { pj[0] = p->cam[0] / Xc[2]; pj[2] = -p->cam[0] * Xc[0] / (Xc[2] * Xc[2]); pj[4] = p->cam[1] / Xc[2]; pj[5] = -p->cam[1] * Xc[1] / (Xc[2] * Xc[2]); }
  if (kind == OSH_EDGE_STEREO) { const double inv_z2 = 1.0 / (Xc[2] * Xc[2]); pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + p->cam[4] * inv_z2; }
  const double x = Xb[0], y = Xb[1], z = Xb[2];
  const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
  m3_mul(pj, Rcb, M);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 6; ++j) Jp[i * 6 + j] = M[i * 3] * D[j] + M[i * 3 + 1] * D[6 + j] + M[i * 3 + 2] * D[12 + j];
}
static int pi_depth_positive(const pstate* s, int e) {
  const int right = s->p->edge_kind[e] == OSH_EDGE_RIGHT;
  const double* R = right ? s->Rcw1 : s->Rcw; const double* X = s->p->points + 3 * (size_t)e;
  return (R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + (right ? s->tcw1 : s->tcw)[2]) > 0.0;
}
/* EdgePriorPoseImu (src/G2oTypes.cc:731-763): residual 15, Jacobian 15 x 15 over (pose 6, v 3, bg 3, ba 3) of the previous frame */
static void pi_prior(const pstate* s, double* r, double* J) {
  const osh_posei_problem* p = s->p;
  double T[9], d[3];
  m3_tmul(p->prior_Rwb, s->pRwb, T);
  oracle_log_so3(T, r);
  for (int i = 0; i < 3; ++i) d[i] = s->ptwb[i] - p->prior_twb[i];
  m3_tvec(p->prior_Rwb, d, r + 3);
  for (int i = 0; i < 3; ++i) { r[6 + i] = s->pv[i] - p->prior_vel[i]; r[9 + i] = s->pbg[i] - p->prior_bg[i]; r[12 + i] = s->pba[i] - p->prior_ba[i]; }
  if (!J) return;
  double invJr[9];
  inv_right_jac(r, invJr);
  memset(J, 0, sizeof(double) * 225);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { J[i * 15 + j] = invJr[i * 3 + j]; J[(3 + i) * 15 + 3 + j] = T[i * 3 + j]; }
  for (int i = 6; i < 15; ++i) J[i * 15 + i] = 1.0;
}
/* dense LDL^T solve; fails unless every pivot is positive (Eigen::LDLT::isPositive) */
static int pi_solve(int n, double* A, const double* b, double* x) {
  for (int k = 0; k < n; ++k) {
    const double dk = A[k * n + k];
    if (!(dk > 0.0)) return 0;
    for (int i = k + 1; i < n; ++i) {
      const double l = A[k * n + i] / dk;
      for (int j = i; j < n; ++j) A[i * n + j] -= l * A[k * n + j];
      A[i * n + k] = l;
    }
  }
  for (int i = 0; i < n; ++i) { double t = b[i]; for (int k = 0; k < i; ++k) t -= A[i * n + k] * x[k]; x[i] = t; }
  for (int i = 0; i < n; ++i) x[i] /= A[i * n + i];
  for (int i = n - 1; i >= 0; --i) { double t = x[i]; for (int k = i + 1; k < n; ++k) t -= A[k * n + i] * x[k]; x[i] = t; }
  return 1;
}
/* adds J^T W J (columns ca..ca+da x cb..cb+db of a d-row Jacobian with leading dimension ld) into H at (oa, ob) and its mirror */
static void pi_add(double* H, int n, const double* J, int ld, int d, const double* W, int ca, int da, int oa, int cb, int db, int ob) {
  for (int i = 0; i < da; ++i)
    for (int j = 0; j < db; ++j) {
      double acc = 0;
      for (int k = 0; k < d; ++k) { double t = 0; for (int m = 0; m < d; ++m) t += W[k * d + m] * J[m * ld + cb + j]; acc += J[k * ld + ca + i] * t; }
      H[(size_t)(oa + i) * n + ob + j] += acc;
      if (oa != ob) H[(size_t)(ob + j) * n + oa + i] += acc;
    }
}
static void pi_pose_update(double* Rwb, double* twb, const double* pu) {   /* ImuCamPose::Update (src/G2oTypes.cc:187-210) */
  double t[3], E[9];
  m3_vec(Rwb, pu + 3, t);
  for (int i = 0; i < 3; ++i) twb[i] += t[i];
  oracle_exp_so3(pu, E);
  m3_mul(Rwb, E, Rwb);
}
/* symmetric eigen-decomposition by cyclic Jacobi rotations: A (n x n, symmetric) -> eigenvalues w, eigenvectors in the columns of V */
static void pi_sym_eig(int n, double* A, double* w, double* V) {
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (A[p * n + q] == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2 * A[p * n + q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
        const double c = 1 / sqrt(t * t + 1), sn = t * c;
        for (int k = 0; k < n; ++k) { const double akp = A[k * n + p], akq = A[k * n + q]; A[k * n + p] = c * akp - sn * akq; A[k * n + q] = sn * akp + c * akq; }
        for (int k = 0; k < n; ++k) { const double apk = A[p * n + k], aqk = A[q * n + k]; A[p * n + k] = c * apk - sn * aqk; A[q * n + k] = sn * apk + c * aqk; }
        for (int k = 0; k < n; ++k) { const double vkp = V[k * n + p], vkq = V[k * n + q]; V[k * n + p] = c * vkp - sn * vkq; V[k * n + q] = sn * vkp + c * vkq; }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}
/* Optimizer::Marginalize(H, 0, 14) of a 30x30 H followed by .block<15,15>(15,15) (src/Optimizer.cc:2967-3050, :5293-5294):
 * Hc - Hcb pinv(Hb) Hbc with the pseudo-inverse from the singular values above 1e-6 (symmetric Hb: |eigenvalues|) */
void oracle_marginalize_previous(const double* H30, double* out15) {
  double Hb[225], w[15], V[225], inv[225];
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) Hb[i * 15 + j] = 0.5 * (H30[i * 30 + j] + H30[j * 30 + i]);
  pi_sym_eig(15, Hb, w, V);
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) {
    double t = 0;
    for (int k = 0; k < 15; ++k) if (fabs(w[k]) > 1e-6) t += V[i * 15 + k] * V[j * 15 + k] / w[k];
    inv[i * 15 + j] = t;
  }
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) {
    double acc = H30[(15 + i) * 30 + 15 + j];
    for (int k = 0; k < 15; ++k) { double t = 0; for (int m = 0; m < 15; ++m) t += inv[k * 15 + m] * H30[m * 30 + 15 + j]; acc -= H30[(15 + i) * 30 + k] * t; }
    out15[i * 15 + j] = acc;
  }
}
/* ConstraintPoseImu constructor (include/G2oTypes.h:711-722): symmetrise, zero the eigenvalues below 1e-12, rebuild */
void oracle_constraint_pose_imu_H(double* H15) {
  double A[225], w[15], V[225];
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) A[i * 15 + j] = 0.5 * (H15[i * 15 + j] + H15[j * 15 + i]);
  pi_sym_eig(15, A, w, V);
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) {
    double t = 0;
    for (int k = 0; k < 15; ++k) if (!(w[k] < 1e-12)) t += V[i * 15 + k] * w[k] * V[j * 15 + k];
    H15[i * 15 + j] = t;
  }
}

/* computeActiveErrors + buildSystem of one Gauss-Newton iteration: unknowns [cur P V G A] then, in mode 1, [prev P V G A]
 * (vertex ids 0..3, 4..7); err receives the errors of the active visual edges */
static void pi_build_system(pstate* s, const unsigned char* level, int robust, double* err, double* H, double* b) {
  const osh_posei_problem* p = s->p;
  const int E = p->n_edges, mode1 = p->mode == 1, n = mode1 ? 30 : 15;
  /* computeActiveErrors */
  for (int e = 0; e < E; ++e) if (!level[e]) pi_vis_error(s, e, err + 3 * (size_t)e);
  double ri[9], Ji[9 * 24], rp[15], Jp15[225];
  inertial_error_core(p->preint, s->pRwb, s->ptwb, s->pv, s->pbg, s->pba, s->Rwb, s->twb, s->v, ri);
  /* buildSystem: unknowns [cur P V G A] then, in mode 1, [prev P V G A] (vertex ids 0..3, 4..7) */
  memset(H, 0, sizeof(double) * (size_t)n * n); memset(b, 0, sizeof(double) * n);
  for (int e = 0; e < E; ++e) {
    if (level[e]) continue;
    const double* r = err + 3 * (size_t)e;
    const double w = p->edge_info[e];
    double rho[3] = {0, 1, 0}, J[18];
    if (robust) oracle_huber(pi_vis_chi2(s, e, r), p->edge_kind[e] == OSH_EDGE_STEREO ? p->huber_stereo : p->huber_mono, rho);
    pi_vis_jac(s, e, J);
    const double ww = rho[1] * w;
    for (int i = 0; i < 6; ++i) {
      b[i] += J[i] * (-(w * r[0]) * rho[1]) + J[6 + i] * (-(w * r[1]) * rho[1]) + J[12 + i] * (-(w * r[2]) * rho[1]);
      for (int j = 0; j < 6; ++j) H[i * n + j] += (J[i] * ww) * J[j] + (J[6 + i] * ww) * J[6 + j] + (J[12 + i] * ww) * J[12 + j];
    }
  }
  inertial_jac_core(p->preint, s->pRwb, s->ptwb, s->pv, s->pbg, s->pba, s->Rwb, s->twb, s->v, Ji);
  {
    /* EdgeInertial vertices (P1 V1 G1 A1 P2 V2): columns 0 6 9 12 15 21; offsets in the unknown vector (-1 fixed) */
    static const int col[6] = {0, 6, 9, 12, 15, 21}, dim[6] = {6, 3, 3, 3, 6, 3};
    const int off[6] = {mode1 ? 15 : -1, mode1 ? 21 : -1, mode1 ? 24 : -1, mode1 ? 27 : -1, 0, 6};
    double wr[9];
    for (int i = 0; i < 9; ++i) { double t = 0; for (int j = 0; j < 9; ++j) t += p->info_inertial[i * 9 + j] * ri[j]; wr[i] = -t; }
    for (int va = 0; va < 6; ++va) {
      if (off[va] < 0) continue;
      for (int i = 0; i < dim[va]; ++i) { double t = 0; for (int k = 0; k < 9; ++k) t += Ji[k * 24 + col[va] + i] * wr[k]; b[off[va] + i] += t; }
      for (int vb = va; vb < 6; ++vb) {
        if (off[vb] < 0) continue;
        pi_add(H, n, Ji, 24, 9, p->info_inertial, col[va], dim[va], off[va], col[vb], dim[vb], off[vb]);
      }
    }
  }
  for (int which = 0; which < 2; ++which) {   /* EdgeGyroRW / EdgeAccRW: r = b_cur - b_prev, J = [-I, I] */
    const double* Og = which == 0 ? p->info_g : p->info_a;
    const double* c2 = which == 0 ? s->bg : s->ba; const double* c1 = which == 0 ? s->pbg : s->pba;
    const int o2 = 9 + 3 * which, o1 = mode1 ? 24 + 3 * which : -1;
    double r[3], Or[3];
    for (int i = 0; i < 3; ++i) r[i] = c2[i] - c1[i];
    for (int i = 0; i < 3; ++i) Or[i] = -(Og[i * 3] * r[0] + Og[i * 3 + 1] * r[1] + Og[i * 3 + 2] * r[2]);
    for (int i = 0; i < 3; ++i) {
      b[o2 + i] += Or[i];
      if (o1 >= 0) b[o1 + i] += -Or[i];
      for (int j = 0; j < 3; ++j) {
        H[(o2 + i) * n + o2 + j] += Og[i * 3 + j];
        if (o1 >= 0) { H[(o1 + i) * n + o1 + j] += Og[i * 3 + j]; H[(o1 + i) * n + o2 + j] += -Og[i * 3 + j]; H[(o2 + j) * n + o1 + i] += -Og[i * 3 + j]; }
      }
    }
  }
  if (mode1) {   /* EdgePriorPoseImu, Huber(huber_prior) */
    pi_prior(s, rp, Jp15);
    double rho[3], W[225], wr[15];
    oracle_huber(quad(rp, p->prior_H, 15), p->huber_prior, rho);
    for (int i = 0; i < 225; ++i) W[i] = rho[1] * p->prior_H[i];
    for (int i = 0; i < 15; ++i) { double t = 0; for (int j = 0; j < 15; ++j) t += p->prior_H[i * 15 + j] * rp[j]; wr[i] = -t * rho[1]; }
    for (int i = 0; i < 15; ++i) { double t = 0; for (int k = 0; k < 15; ++k) t += Jp15[k * 15 + i] * wr[k]; b[15 + i] += t; }
    /* the prior's four vertices are contiguous in the unknown vector: one 15x15 block (BaseMultiEdge adds every vertex pair) */
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) {
      double acc = 0;
      for (int k = 0; k < 15; ++k) { double t = 0; for (int m = 0; m < 15; ++m) t += W[k * 15 + m] * Jp15[m * 15 + j]; acc += Jp15[k * 15 + i] * t; }
      H[(15 + i) * n + 15 + j] += acc;
    }
  }
}

static int pi_init(pstate* s, const osh_posei_problem* p) {
  memset(s, 0, sizeof(*s));
  s->p = p;
  const int E = p->n_edges;
  s->rig = (p->kb8 && p->cam2 && p->trl) ? 1 : 0;
  memcpy(s->Rcw, p->Rcw, 72); memcpy(s->tcw, p->tcw, 24); memcpy(s->Rwb, p->Rwb, 72); memcpy(s->twb, p->twb, 24);
  memcpy(s->v, p->vel, 24); memcpy(s->bg, p->bias_g, 24); memcpy(s->ba, p->bias_a, 24);
  memcpy(s->pRwb, p->prev_Rwb, 72); memcpy(s->ptwb, p->prev_twb, 24); memcpy(s->pv, p->prev_vel, 24); memcpy(s->pbg, p->prev_bias_g, 24); memcpy(s->pba, p->prev_bias_a, 24);
  for (int e = 0; e < E; ++e) if (p->edge_kind[e] == OSH_EDGE_RIGHT && !s->rig) return 0;
  if (s->rig) {   /* ImuCamPose(Frame*) camera 1 (src/G2oTypes->cc:104-115) */
    double t[3], Rbc1[9];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) s->Rrl[i * 3 + j] = p->trl[i * 4 + j]; s->trl[i] = p->trl[i * 4 + 3]; }
    m3_mul(s->Rrl, s->Rcw, s->Rcw1);
    m3_vec(s->Rrl, s->tcw, t); for (int i = 0; i < 3; ++i) s->tcw1[i] = t[i] + s->trl[i];
    m3_mul(s->Rrl, p->Rcb, s->Rcb1);
    m3_vec(s->Rrl, p->tcb, t); for (int i = 0; i < 3; ++i) s->tcb1[i] = t[i] + s->trl[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbc1[i * 3 + j] = s->Rcb1[j * 3 + i];
    m3_vec(Rbc1, s->tcb1, t); for (int i = 0; i < 3; ++i) s->tbc1[i] = -t[i];
  }
  return 1;
}

/* Debug / parity aid: the system of the first Gauss-Newton iteration (all edges active, Huber on) */
int oracle_posei_linearize(const osh_posei_problem* p, double* H, double* b) {
  pstate s;
  if (!pi_init(&s, p)) return OSH_ERR_INVALID;
  unsigned char* level = calloc(p->n_edges ? p->n_edges : 1, 1);
  double* err = calloc(3 * (size_t)(p->n_edges ? p->n_edges : 1), sizeof(double));
  pi_build_system(&s, level, 1, err, H, b);
  free(level); free(err);
  return OSH_OK;
}

int oracle_posei_optimize(const osh_posei_problem* p, osh_posei_result* res) {
  pstate s;
  if (!pi_init(&s, p)) return OSH_ERR_INVALID;
  const int E = p->n_edges, mode1 = p->mode == 1, n = mode1 ? 30 : 15;
  unsigned char* level = calloc(E ? E : 1, 1);     /* 1: outside the active set */
  unsigned char* outlier = calloc(E ? E : 1, 1);   /* pFrame->mvbOutlier */
  double* err = calloc(3 * (size_t)(E ? E : 1), sizeof(double));   /* _error of every edge as last computed */
  double H[900], b[30], x[30], A[900];
  memset(x, 0, sizeof(x));
  int robust = 1, n_bad = 0, n_inliers = 0, rounds = 0;
  const int n_graph_edges = E + (mode1 ? 4 : 3);
  for (int round = 0; round < 4; ++round) {
    int ok = 1, any = 0;
    for (int e = 0; e < E; ++e) any |= !level[e];
    (void)any;   /* the inertial edges keep the active set non-empty */
    for (int it = 0; it < p->iterations[round] && ok; ++it) {
      pi_build_system(&s, level, robust, err, H, b);
      memcpy(A, H, sizeof(double) * (size_t)n * n);
      ok = pi_solve(n, A, b, x);   /* on failure x keeps the previous iteration's values and is still applied (update() precedes the check) */
      pi_pose_update(s.Rwb, s.twb, x);
      pi_cams_from_body(&s);
      for (int i = 0; i < 3; ++i) { s.v[i] += x[6 + i]; s.bg[i] += x[9 + i]; s.ba[i] += x[12 + i]; }
      if (mode1) {
        pi_pose_update(s.pRwb, s.ptwb, x + 15);
        for (int i = 0; i < 3; ++i) { s.pv[i] += x[21 + i]; s.pbg[i] += x[24 + i]; s.pba[i] += x[27 + i]; }
      }
    }
    /* classification (:4747-4818 / :5139-5208): an inlier keeps the error of the last computeActiveErrors (start of the last
     * iteration), an outlier is recomputed at the final estimate; the depth test always uses the final estimate */
    int bad = 0, inl = 0;
    const float chi2close = 1.5 * p->chi2_mono[round];
    for (int pass = 0; pass < 2; ++pass)
      for (int e = 0; e < E; ++e) {
        const int stereo = p->edge_kind[e] == OSH_EDGE_STEREO;
        if (stereo != pass) continue;
        if (outlier[e]) pi_vis_error(&s, e, err + 3 * (size_t)e);
        const float chi2 = (float)pi_vis_chi2(&s, e, err + 3 * (size_t)e);
        int out;
        if (!stereo) {
          const int bClose = p->edge_close ? p->edge_close[e] : 0;
          out = (chi2 > p->chi2_mono[round] && !bClose) || (bClose && chi2 > chi2close) || !pi_depth_positive(&s, e);
        } else out = chi2 > p->chi2_stereo[round];
        outlier[e] = (unsigned char)out; level[e] = (unsigned char)out;
        if (out) ++bad; else ++inl;
      }
    n_bad = bad; n_inliers = inl; rounds = round + 1;
    if (round == 2) robust = 0;
    if (n_graph_edges < 10) break;
  }
  if (n_inliers < 30 && !p->rec_init) {   /* recovery (:4821-4848): every edge re-evaluated, generous thresholds */
    n_bad = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (int e = 0; e < E; ++e) {
        const int stereo = p->edge_kind[e] == OSH_EDGE_STEREO;
        if (stereo != pass) continue;
        pi_vis_error(&s, e, err + 3 * (size_t)e);
        if (pi_vis_chi2(&s, e, err + 3 * (size_t)e) < (stereo ? 24.f : 18.f)) outlier[e] = 0; else n_bad++;
      }
  }
  /* Hessian of the frame's ConstraintPoseImu (:4858-4893 / :5252-5293): every edge re-linearised at the final estimate, plain information */
  {
    double Ji[9 * 24];
    const int nn = mode1 ? 30 : 15, o2 = mode1 ? 15 : 0;
    memset(res->H, 0, sizeof(res->H));
    inertial_jac_core(p->preint, s.pRwb, s.ptwb, s.pv, s.pbg, s.pba, s.Rwb, s.twb, s.v, Ji);
    if (mode1) pi_add(res->H, nn, Ji, 24, 9, p->info_inertial, 0, 24, 0, 0, 24, 0);                   /* GetHessian(): 24 x 24 at (0,0) */
    else { pi_add(res->H, nn, Ji, 24, 9, p->info_inertial, 15, 9, 0, 15, 9, 0); }                     /* GetHessian2(): (P2, V2) 9 x 9 */
    for (int which = 0; which < 2; ++which) {
      const double* Og = which == 0 ? p->info_g : p->info_a;
      const int c2 = o2 + 9 + 3 * which, c1 = 9 + 3 * which;
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        res->H[(c2 + i) * nn + c2 + j] += Og[i * 3 + j];
        if (mode1) { res->H[(c1 + i) * nn + c1 + j] += Og[i * 3 + j]; res->H[(c1 + i) * nn + c2 + j] += -Og[i * 3 + j]; res->H[(c2 + i) * nn + c1 + j] += -Og[i * 3 + j]; }
      }
    }
    if (mode1) {
      double rp[15], Jp15[225];
      pi_prior(&s, rp, Jp15);
      for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) {
        double acc = 0;
        for (int k = 0; k < 15; ++k) { double t = 0; for (int m = 0; m < 15; ++m) t += p->prior_H[k * 15 + m] * Jp15[m * 15 + j]; acc += Jp15[k * 15 + i] * t; }
        res->H[i * nn + j] += acc;
      }
    }
    for (int pass = 0; pass < 2; ++pass)
      for (int e = 0; e < E; ++e) {
        const int stereo = p->edge_kind[e] == OSH_EDGE_STEREO;
        if (stereo != pass || outlier[e]) continue;
        double J[18];
        pi_vis_jac(&s, e, J);
        const double w = p->edge_info[e];
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) res->H[(o2 + i) * nn + o2 + j] += (J[i] * w) * J[j] + (J[6 + i] * w) * J[6 + j] + (J[12 + i] * w) * J[12 + j];
      }
  }
  memcpy(res->Rcw, s.Rcw, 72); memcpy(res->tcw, s.tcw, 24); memcpy(res->Rwb, s.Rwb, 72); memcpy(res->twb, s.twb, 24);
  memcpy(res->vel, s.v, 24); memcpy(res->bias_g, s.bg, 24); memcpy(res->bias_a, s.ba, 24);
  if (res->outlier) memcpy(res->outlier, outlier, E);
  if (res->edge_chi2) for (int e = 0; e < E; ++e) res->edge_chi2[e] = pi_vis_chi2(&s, e, err + 3 * (size_t)e);
  res->n_bad = n_bad; res->n_inliers = n_inliers; res->rounds = rounds; res->status = OSH_OK;
  free(level); free(outlier); free(err);
  return OSH_OK;
}
