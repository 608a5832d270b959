/*
 * lba_oracle.c -- CPU restatement (FP64, plain C, Eigen-free) of the g2o path
 * behind ORB_SLAM3::Optimizer::LocalBundleAdjustment.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the shipped
 * library (orb_slam3_study_kr_amd/csrc) never links or calls it.
 *
 * PARITY UNPINNED: the reference cannot be compiled in this image (Eigen3,
 * OpenCV, Boost are absent) and it ships no golden
 * vectors or unit tests for this path.  This file is pinned instead by
 *   (1) an independent numpy re-derivation (oracle/lm_numpy.py),
 *   (2) g2o's own central-difference Jacobian recipe (base_binary_edge.hpp:147-197),
 *   (3) Schur solve == dense full-system solve, and zero-noise convergence,
 * see tests/test_oracle_*.py.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference; "g2o/" = Thirdparty/g2o/g2o/).
 */
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "../include/orbslam3_hip.h"
#include "oracle.h"

/* ------------------------------------------------------------------------ */
/* Eigen primitives                                   */
/* ------------------------------------------------------------------------ */

/* Eigen Quaternion::_transformVector: uv = 2 (q_v x v); v + w uv + q_v x uv.
 * q = (x,y,z,w).  Used by SE3Quat::map, g2o/types/se3quat.h:217-221. */
static void quat_rotate(const double q[4], const double v[3], double o[3]) {
  double uv0 = q[1] * v[2] - q[2] * v[1];
  double uv1 = q[2] * v[0] - q[0] * v[2];
  double uv2 = q[0] * v[1] - q[1] * v[0];
  uv0 += uv0; uv1 += uv1; uv2 += uv2;
  o[0] = v[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
  o[1] = v[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
  o[2] = v[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
}

/* Hamilton product a*b (Eigen quaternion operator*), q = (x,y,z,w). */
static void quat_mul(const double a[4], const double b[4], double o[4]) {
  double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

/* SE3Quat::normalizeRotation, g2o/types/se3quat.h:280-285. */
static void quat_normalize_rotation(double q[4]) {
  if (q[3] < 0) { q[0] *= -1; q[1] *= -1; q[2] *= -1; q[3] *= -1; }
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* Eigen Quaternion::toRotationMatrix, row-major R[9]. */
static void quat_to_R(const double q[4], double R[9]) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* Eigen Quaterniond(Matrix3d) (quaternionbase_assign_impl), row-major R. */
static void R_to_quat(const double R[9], double q[4]) {
  double t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
}

static void mat3_mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}

/* Eigen Matrix3d::inverse(): cofactors times 1/det, no pivoting
 * (used at g2o/core/block_solver.hpp:389). */
static void mat3_inverse(const double m[9], double inv[9]) {
  const double c00 = m[4] * m[8] - m[5] * m[7];
  const double c10 = m[5] * m[6] - m[3] * m[8];
  const double c20 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 + m[1] * c10 + m[2] * c20;
  const double invdet = 1.0 / det;
  inv[0] = c00 * invdet;
  inv[3] = c10 * invdet;
  inv[6] = c20 * invdet;
  inv[1] = (m[2] * m[7] - m[1] * m[8]) * invdet;
  inv[4] = (m[0] * m[8] - m[2] * m[6]) * invdet;
  inv[7] = (m[1] * m[6] - m[0] * m[7]) * invdet;
  inv[2] = (m[1] * m[5] - m[2] * m[4]) * invdet;
  inv[5] = (m[2] * m[3] - m[0] * m[5]) * invdet;
  inv[8] = (m[0] * m[4] - m[1] * m[3]) * invdet;
}

/* ------------------------------------------------------------------------ */
/* SE3Quat::exp and VertexSE3Expmap::oplusImpl                               */
/* ------------------------------------------------------------------------ */

/* SE3Quat::exp, g2o/types/se3quat.h:223-259.  update = (omega, upsilon).
 * Note the small-angle branch R = I + Omega + Omega^2 (sic). */
static void se3_exp(const double u[6], double q[4], double t[3]) {
  const double w0 = u[0], w1 = u[1], w2 = u[2];
  const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
  const double Om[9] = {0, -w2, w1, w2, 0, -w0, -w1, w0, 0}; /* skew, se3_ops.hpp:27-40 */
  double Om2[9], R[9], V[9];
  mat3_mul(Om, Om, Om2);
  if (theta < 0.00001) {
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + Om[i] + Om2[i];
    memcpy(V, R, sizeof(R));
  } else {
    const double a = sin(theta) / theta;
    const double b = (1 - cos(theta)) / (theta * theta);
    const double c = (theta - sin(theta)) / pow(theta, 3);
    for (int i = 0; i < 9; ++i) {
      const double I = (i % 4 == 0) ? 1.0 : 0.0;
      R[i] = I + a * Om[i] + b * Om2[i];
      V[i] = I + b * Om[i] + c * Om2[i];
    }
  }
  R_to_quat(R, q);
  for (int i = 0; i < 3; ++i) t[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
  quat_normalize_rotation(q); /* SE3Quat(const Quaterniond&, const Vector3d&), se3quat.h:61-63 */
}

/* setEstimate(SE3Quat::exp(update)*estimate()), types_six_dof_expmap.h:73-76
 * with SE3Quat::operator*, se3quat.h:104-110. */
void oracle_pose_oplus(const double update[6], double qt[7]) {
  double eq[4], et[3], rt[3], nq[4];
  se3_exp(update, eq, et);
  quat_rotate(eq, qt + 4, rt);
  quat_mul(eq, qt, nq);
  qt[4] = et[0] + rt[0]; qt[5] = et[1] + rt[1]; qt[6] = et[2] + rt[2];
  quat_normalize_rotation(nq);
  memcpy(qt, nq, sizeof(nq));
}

/* ------------------------------------------------------------------------ */
/* Edge models                                                               */
/* ------------------------------------------------------------------------ */

/* SE3Quat::map, se3quat.h:217-221 */
static void se3_map(const double qt[7], const double X[3], double Xc[3]) {
  double r[3];
  quat_rotate(qt, X, r);
  Xc[0] = r[0] + qt[4]; Xc[1] = r[1] + qt[5]; Xc[2] = r[2] + qt[6];
}

/* computeError of the two visual edges.
 *  mono  : ORB_SLAM3::EdgeSE3ProjectXYZ::computeError include/OptimizableTypes.h:99-104
 *          with Pinhole::project src/CameraModels/Pinhole.cpp:35-41
 *  stereo: g2o::EdgeStereoSE3ProjectXYZ::computeError types_six_dof_expmap.h:155-160
 *          with cam_project types_six_dof_expmap.cpp:190-197 (float invz, float bf) */
void oracle_edge_error(int kind, const double qt[7], const double cam[5],
                       const double X[3], const double obs[3], double err[3]) {
  double Xc[3];
  se3_map(qt, X, Xc);
  if (kind == OSH_EDGE_MONO) {
    err[0] = obs[0] - (cam[0] * Xc[0] / Xc[2] + cam[2]);
    err[1] = obs[1] - (cam[1] * Xc[1] / Xc[2] + cam[3]);
    err[2] = 0.0;
  } else {
    const float invz = (float)(1.0f / Xc[2]);
    const float bf = (float)cam[4];
    const double u = Xc[0] * invz * cam[0] + cam[2];
    const double v = Xc[1] * invz * cam[1] + cam[3];
    const float bfz = bf * invz;
    err[0] = obs[0] - u;
    err[1] = obs[1] - v;
    err[2] = obs[2] - (u - bfz);
  }
}

int oracle_edge_depth_positive(const double qt[7], const double X[3]) {
  double Xc[3];
  se3_map(qt, X, Xc);
  return Xc[2] > 0.0;
}

/* linearizeOplus.  Jxi = d err / d point (d x 3), Jxj = d err / d pose (d x 6),
 * row-major with 3 rows allocated (row 2 zero for mono).
 *  mono  : src/OptimizableTypes.cpp:139-160, Pinhole::projectJac Pinhole.cpp:71-81
 *  stereo: types_six_dof_expmap.cpp:228-273 */
void oracle_edge_jacobians(int kind, const double qt[7], const double cam[5],
                           const double X[3], double Jxi[9], double Jxj[18]) {
  double Xc[3], R[9];
  se3_map(qt, X, Xc);
  quat_to_R(qt, R);
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  memset(Jxi, 0, 9 * sizeof(double));
  memset(Jxj, 0, 18 * sizeof(double));
  if (kind == OSH_EDGE_MONO) {
    /* projectJac = -pCamera->projectJac(xyz_trans) */
    double pj[6];
    pj[0] = -(cam[0] / z); pj[1] = -0.0; pj[2] = -(-cam[0] * x / (z * z));
    pj[3] = -0.0; pj[4] = -(cam[1] / z); pj[5] = -(-cam[1] * y / (z * z));
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += pj[i * 3 + k] * R[k * 3 + j];
        Jxi[i * 3 + j] = s;
      }
    const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 6; ++j) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += pj[i * 3 + k] * D[k * 6 + j];
        Jxj[i * 6 + j] = s;
      }
  } else {
    const double fx = cam[0], fy = cam[1], bf = cam[4];
    const double z_2 = z * z;
    Jxi[0] = -fx * R[0] / z + fx * x * R[6] / z_2;
    Jxi[1] = -fx * R[1] / z + fx * x * R[7] / z_2;
    Jxi[2] = -fx * R[2] / z + fx * x * R[8] / z_2;
    Jxi[3] = -fy * R[3] / z + fy * y * R[6] / z_2;
    Jxi[4] = -fy * R[4] / z + fy * y * R[7] / z_2;
    Jxi[5] = -fy * R[5] / z + fy * y * R[8] / z_2;
    Jxi[6] = Jxi[0] - bf * R[6] / z_2;
    Jxi[7] = Jxi[1] - bf * R[7] / z_2;
    Jxi[8] = Jxi[2] - bf * R[8] / z_2;

    Jxj[0] = x * y / z_2 * fx;
    Jxj[1] = -(1 + (x * x / z_2)) * fx;
    Jxj[2] = y / z * fx;
    Jxj[3] = -1. / z * fx;
    Jxj[4] = 0;
    Jxj[5] = x / z_2 * fx;

    Jxj[6] = (1 + y * y / z_2) * fy;
    Jxj[7] = -x * y / z_2 * fy;
    Jxj[8] = -x / z * fy;
    Jxj[9] = 0;
    Jxj[10] = -1. / z * fy;
    Jxj[11] = y / z_2 * fy;

    Jxj[12] = Jxj[0] - bf * y / z_2;
    Jxj[13] = Jxj[1] + bf * x / z_2;
    Jxj[14] = Jxj[2];
    Jxj[15] = Jxj[3];
    Jxj[16] = 0;
    Jxj[17] = Jxj[5] - bf / z_2;
  }
}

/* RobustKernelHuber::robustify, g2o/core/robust_kernel_impl.cpp:78-91;
 * dsqr = delta*delta (setDelta :65-69). */
void oracle_huber(double e, double delta, double rho[3]) {
  const double dsqr = delta * delta;
  if (e <= dsqr) {
    rho[0] = e; rho[1] = 1.; rho[2] = 0.;
  } else {
    const double sqrte = sqrt(e);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
    rho[2] = -0.5 * rho[1] / e;
  }
}

/* ------------------------------------------------------------------------ */
/* The optimiser state                                                       */
/* ------------------------------------------------------------------------ */
typedef struct {
  int P, F, L, E, nblk, sizePoses, sizeLm;
  const osh_lba_problem* pr;
  double* qt;      /* (P+F)*7 estimates */
  double* X;       /* L*3 */
  double* qt_bak;  /* backup stack (depth 1 is all LM needs) */
  double* X_bak;
  double* err;     /* E*3 _error of each edge */
  /* Hessian (block_solver.hpp buildStructure :143-295) */
  double* Hpp;     /* P*36 */
  double* Hll;     /* L*9 */
  double* Hpl;     /* nblk*18, 6x3 row-major (pose row, landmark col) */
  double* b;       /* 6P+3L */
  int* edge_blk;   /* E -> block or -1 (fixed pose) */
  int* col_off;    /* L+1: blocks of landmark column j, rows ascending (CCS) */
  int* blk_row;    /* nblk */
  double* Hschur;  /* (6P)^2 dense, upper triangle used */
  double* bschur;  /* 6P */
  double* coeff;   /* 6P+3L */
  double* x;       /* 6P+3L */
  double* Dinv;    /* L*9 */
  double* diag_bak;/* 6P+3L */
  double* ldl_tmp; /* 6P */
} ostate;

static double edge_chi2(const ostate* s, int e) {
  /* BaseEdge::chi2 = _error.dot(information()*_error), g2o/core/base_edge.h:58-61 */
  const double w = s->pr->edge_info[e];
  const double* r = s->err + 3 * e;
  if (s->pr->edge_kind[e] != OSH_EDGE_STEREO) return r[0] * (w * r[0]) + r[1] * (w * r[1]);   /* mono and body edges: 2 rows */
  return r[0] * (w * r[0]) + r[1] * (w * r[1]) + r[2] * (w * r[2]);
}

/* ---- KannalaBrandt8 monocular edge: ORB_SLAM3::EdgeSE3ProjectXYZ (src/OptimizableTypes.cpp:139-160) with
 * KannalaBrandt8::project(Vector3d) (src/CameraModels/KannalaBrandt8.cpp:45-63) and ::projectJac (:147-175).
 * cam = fx fy cx cy (mvParameters[0..3]), kb = k1..k4 (mvParameters[4..7]).
 * The reference calls atan2f / sqrtf on double arguments (theta and psi are float32 values).  float32 atan2 is taken as
 * its correctly rounded value (double atan2 rounded once): libm independent, so the device reproduces it; glibc's atan2f
 * differs from it only in rare 1-ulp cases. */
static float atan2f_rn(float y, float x) { return (float)atan2((double)y, (double)x); }

/* KannalaBrandt8::project(Vector3d), src/CameraModels/KannalaBrandt8.cpp:45-63 */
void oracle_kb8_project(const double cam[5], const double kb[4], const double Xc[3], double uv[2]) {
  const double x2_plus_y2 = Xc[0] * Xc[0] + Xc[1] * Xc[1];
  const double theta = atan2f_rn(sqrtf((float)x2_plus_y2), (float)Xc[2]);
  const double psi = atan2f_rn((float)Xc[1], (float)Xc[0]);
  const double theta2 = theta * theta;
  const double theta3 = theta * theta2;
  const double theta5 = theta3 * theta2;
  const double theta7 = theta5 * theta2;
  const double theta9 = theta7 * theta2;
  const double r = theta + kb[0] * theta3 + kb[1] * theta5 + kb[2] * theta7 + kb[3] * theta9;
  uv[0] = cam[0] * r * cos(psi) + cam[2];
  uv[1] = cam[1] * r * sin(psi) + cam[3];
}

/* KannalaBrandt8::projectJac, src/CameraModels/KannalaBrandt8.cpp:147-175; J row-major 2x3 */
void oracle_kb8_project_jac(const double cam[5], const double kb[4], const double Xc[3], double J[6]) {
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  const double x2 = x * x, y2 = y * y, z2 = z * z;
  const double r2 = x2 + y2;
  const double r = sqrt(r2);
  const double r3 = r2 * r;
  const double theta = atan2(r, z);
  const double theta2 = theta * theta, theta3 = theta2 * theta;
  const double theta4 = theta2 * theta2, theta5 = theta4 * theta;
  const double theta6 = theta2 * theta4, theta7 = theta6 * theta;
  const double theta8 = theta4 * theta4, theta9 = theta8 * theta;
  const double f = theta + theta3 * kb[0] + theta5 * kb[1] + theta7 * kb[2] + theta9 * kb[3];
  const double fd = 1 + 3 * kb[0] * theta2 + 5 * kb[1] * theta4 + 7 * kb[2] * theta6 + 9 * kb[3] * theta8;
  J[0] = cam[0] * (fd * z * x2 / (r2 * (r2 + z2)) + f * y2 / r3);
  J[3] = cam[1] * (fd * z * y * x / (r2 * (r2 + z2)) - f * y * x / r3);
  J[1] = cam[0] * (fd * z * y * x / (r2 * (r2 + z2)) - f * y * x / r3);
  J[4] = cam[1] * (fd * z * y2 / (r2 * (r2 + z2)) + f * x2 / r3);
  J[2] = -cam[0] * fd * x / (r2 + z2);
  J[5] = -cam[1] * fd * y / (r2 + z2);
}

void oracle_edge_error_kb8(const double qt[7], const double cam[5], const double kb[4],
                           const double X[3], const double obs[3], double err[3]) {
  double Xc[3], uv[2];
  se3_map(qt, X, Xc);
  oracle_kb8_project(cam, kb, Xc, uv);
  err[0] = obs[0] - uv[0];
  err[1] = obs[1] - uv[1];
  err[2] = 0.0;
}

void oracle_edge_jacobians_kb8(const double qt[7], const double cam[5], const double kb[4],
                               const double X[3], double Jxi[9], double Jxj[18]) {
  double Xc[3], R[9], J[6];
  se3_map(qt, X, Xc);
  quat_to_R(qt, R);
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  memset(Jxi, 0, 9 * sizeof(double));
  memset(Jxj, 0, 18 * sizeof(double));
  oracle_kb8_project_jac(cam, kb, Xc, J);
  double pj[6];
  for (int k = 0; k < 6; ++k) pj[k] = -J[k];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += pj[i * 3 + k] * R[k * 3 + j];
      Jxi[i * 3 + j] = s;
    }
  const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 6; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += pj[i * 3 + k] * D[k * 6 + j];
      Jxj[i * 6 + j] = s;
    }
}

/* ---- right-camera edge of a fisheye stereo rig: ORB_SLAM3::EdgeSE3ProjectXYZToBody
 * (include/OptimizableTypes.h:117-144, src/OptimizableTypes.cpp:192-213), created at src/Optimizer.cc:1365-1399 with
 * mTrl = SE3Quat(Trl) and pCamera = mpCamera2.  cam2 = fx fy cx cy k1..k4 of the right camera, trl = qx qy qz qw tx ty tz.
 * T_rw = mTrl * T_lw is the SE3Quat product (se3quat.h:104-110: rotation product normalised, t = r1 * t2 + t1). */
static void body_T_rw(const double trl[7], const double qt[7], double T_rw[7]) {
  double q[4], rt[3];
  quat_mul(trl, qt, q);
  quat_normalize_rotation(q);
  quat_rotate(trl, qt + 4, rt);
  T_rw[0] = q[0]; T_rw[1] = q[1]; T_rw[2] = q[2]; T_rw[3] = q[3];
  T_rw[4] = rt[0] + trl[4]; T_rw[5] = rt[1] + trl[5]; T_rw[6] = rt[2] + trl[6];
}

void oracle_edge_error_body(const double qt[7], const double cam2[8], const double trl[7],
                            const double X[3], const double obs[3], double err[3]) {
  /* _error = obs - pCamera->project((mTrl * v1->estimate()).map(v2->estimate())) */
  double T_rw[7], Xr[3], uv[2];
  body_T_rw(trl, qt, T_rw);
  se3_map(T_rw, X, Xr);
  oracle_kb8_project(cam2, cam2 + 4, Xr, uv);
  err[0] = obs[0] - uv[0];
  err[1] = obs[1] - uv[1];
  err[2] = 0.0;
}

int oracle_edge_depth_positive_body(const double qt[7], const double trl[7], const double X[3]) {
  double T_rw[7], Xr[3];
  body_T_rw(trl, qt, T_rw);
  se3_map(T_rw, X, Xr);
  return Xr[2] > 0.0;
}

void oracle_edge_jacobians_body(const double qt[7], const double cam2[8], const double trl[7],
                                const double X[3], double Jxi[9], double Jxj[18]) {
  /* X_l = T_lw.map(X_w); X_r = mTrl.map(T_lw.map(X_w))
   * _jacobianOplusXi = -projectJac(X_r) * T_rw.rotation();  _jacobianOplusXj = -projectJac(X_r) * mTrl.rotation() * SE3deriv(X_l) */
  double T_rw[7], Xl[3], Xr[3], Rrw[9], Rrl[9], J[6], pj[6], pjr[6];
  body_T_rw(trl, qt, T_rw);
  se3_map(qt, X, Xl);
  se3_map(trl, Xl, Xr);
  quat_to_R(T_rw, Rrw);
  quat_to_R(trl, Rrl);
  memset(Jxi, 0, 9 * sizeof(double));
  memset(Jxj, 0, 18 * sizeof(double));
  oracle_kb8_project_jac(cam2, cam2 + 4, Xr, J);
  for (int k = 0; k < 6; ++k) pj[k] = -J[k];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0, b = 0;
      for (int k = 0; k < 3; ++k) { a += pj[i * 3 + k] * Rrw[k * 3 + j]; b += pj[i * 3 + k] * Rrl[k * 3 + j]; }
      Jxi[i * 3 + j] = a;
      pjr[i * 3 + j] = b;
    }
  const double x = Xl[0], y = Xl[1], z = Xl[2];
  const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 6; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += pjr[i * 3 + k] * D[k * 6 + j];
      Jxj[i * 6 + j] = a;
    }
}

static double edge_delta(const ostate* s, int e) {
  /* rk->setDelta(thHuberMono) for the mono and the body edge (src/Optimizer.cc:1321,1386), thHuberStereo for the stereo edge */
  return s->pr->edge_kind[e] != OSH_EDGE_STEREO ? s->pr->huber_mono : s->pr->huber_stereo;
}

/* SparseOptimizer::computeActiveErrors, g2o/core/sparse_optimizer.cpp:61-88 */
static void compute_active_errors(ostate* s) {
  const osh_lba_problem* p = s->pr;
  for (int e = 0; e < s->E; ++e) {
    const int ip = p->edge_pose[e], il = p->edge_point[e];
    if (p->edge_kind[e] == OSH_EDGE_BODY)
      oracle_edge_error_body(s->qt + 7 * ip, p->cam2, p->trl, s->X + 3 * il, p->edge_obs + 3 * e, s->err + 3 * e);
    else if (p->kb8 && p->edge_kind[e] == OSH_EDGE_MONO)
      oracle_edge_error_kb8(s->qt + 7 * ip, p->pose_cam + 5 * ip, p->kb8, s->X + 3 * il, p->edge_obs + 3 * e, s->err + 3 * e);
    else
      oracle_edge_error(p->edge_kind[e], s->qt + 7 * ip, p->pose_cam + 5 * ip, s->X + 3 * il,
                        p->edge_obs + 3 * e, s->err + 3 * e);
  }
}

/* SparseOptimizer::activeRobustChi2, sparse_optimizer.cpp:100-114 */
static double active_robust_chi2(const ostate* s) {
  double chi = 0.0, rho[3];
  for (int e = 0; e < s->E; ++e) {
    oracle_huber(edge_chi2(s, e), edge_delta(s, e), rho);
    chi += rho[0];
  }
  return chi;
}

/* BlockSolver::buildSystem, block_solver.hpp:502-560, with
 * BaseBinaryEdge::constructQuadraticForm (robust branch), base_binary_edge.hpp:55-120.
 * vertex 0 ("from", Xi) = point, vertex 1 ("to", Xj) = pose. */
static void build_system(ostate* s) {
  const osh_lba_problem* p = s->pr;
  memset(s->Hpp, 0, sizeof(double) * 36 * s->P);
  memset(s->Hll, 0, sizeof(double) * 9 * s->L);
  memset(s->Hpl, 0, sizeof(double) * 18 * s->nblk);
  memset(s->b, 0, sizeof(double) * (s->sizePoses + s->sizeLm));
  double* bp = s->b;
  double* bl = s->b + s->sizePoses;
  for (int e = 0; e < s->E; ++e) {
    const int ip = p->edge_pose[e], il = p->edge_point[e];
    const int kind = p->edge_kind[e];
    const int D = (kind != OSH_EDGE_STEREO) ? 2 : 3;
    double A[9], B[18];
    if (kind == OSH_EDGE_BODY) oracle_edge_jacobians_body(s->qt + 7 * ip, p->cam2, p->trl, s->X + 3 * il, A, B);
    else if (p->kb8 && kind == OSH_EDGE_MONO) oracle_edge_jacobians_kb8(s->qt + 7 * ip, p->pose_cam + 5 * ip, p->kb8, s->X + 3 * il, A, B);
    else oracle_edge_jacobians(kind, s->qt + 7 * ip, p->pose_cam + 5 * ip, s->X + 3 * il, A, B);
    const double w = p->edge_info[e];
    const double* r = s->err + 3 * e;
    double omega_r[3] = {-(w * r[0]), -(w * r[1]), -(w * r[2])};
    double rho[3];
    oracle_huber(edge_chi2(s, e), edge_delta(s, e), rho);
    const double ww = rho[1] * w; /* robustInformation: first-order only, base_edge.h:96-102 */
    for (int k = 0; k < 3; ++k) omega_r[k] *= rho[1];
    /* from (point) is never fixed in local BA */
    {
      double AtW[9]; /* 3 x D */
      for (int i = 0; i < 3; ++i)
        for (int k = 0; k < D; ++k) AtW[i * 3 + k] = A[k * 3 + i] * ww;
      for (int i = 0; i < 3; ++i) {
        double sb = 0;
        for (int k = 0; k < D; ++k) sb += A[k * 3 + i] * omega_r[k];
        bl[3 * il + i] += sb;
        for (int j = 0; j < 3; ++j) {
          double sh = 0;
          for (int k = 0; k < D; ++k) sh += AtW[i * 3 + k] * A[k * 3 + j];
          s->Hll[9 * il + i * 3 + j] += sh;
        }
      }
    }
    if (ip < s->P) {
      double BtW[18]; /* 6 x D */
      for (int i = 0; i < 6; ++i)
        for (int k = 0; k < D; ++k) BtW[i * 3 + k] = B[k * 6 + i] * ww;
      double* H = s->Hpl + 18 * s->edge_blk[e];
      for (int i = 0; i < 6; ++i) {
        /* _hessianTransposed += B^T W A  (pose row x landmark col) */
        for (int j = 0; j < 3; ++j) {
          double sh = 0;
          for (int k = 0; k < D; ++k) sh += BtW[i * 3 + k] * A[k * 3 + j];
          H[i * 3 + j] += sh;
        }
        double sb = 0;
        for (int k = 0; k < D; ++k) sb += B[k * 6 + i] * omega_r[k];
        bp[6 * ip + i] += sb;
        for (int j = 0; j < 6; ++j) {
          double sh = 0;
          for (int k = 0; k < D; ++k) sh += BtW[i * 3 + k] * B[k * 6 + j];
          s->Hpp[36 * ip + i * 6 + j] += sh;
        }
      }
    }
  }
}

/* OptimizationAlgorithmLevenberg::computeLambdaInit, levenberg.cpp:171-185 */
static double compute_lambda_init(const ostate* s) {
  if (s->pr->lambda_init > 0) return s->pr->lambda_init;
  double maxDiagonal = 0.;
  for (int i = 0; i < s->P; ++i)
    for (int j = 0; j < 6; ++j) maxDiagonal = fmax(fabs(s->Hpp[36 * i + 7 * j]), maxDiagonal);
  for (int i = 0; i < s->L; ++i)
    for (int j = 0; j < 3; ++j) maxDiagonal = fmax(fabs(s->Hll[9 * i + 4 * j]), maxDiagonal);
  return 1e-5 * maxDiagonal; /* _tau, levenberg.cpp:47 */
}

/* BlockSolver::setLambda / restoreDiagonal, block_solver.hpp:564-604 */
static void set_lambda(ostate* s, double lambda) {
  for (int i = 0; i < s->P; ++i)
    for (int j = 0; j < 6; ++j) {
      s->diag_bak[6 * i + j] = s->Hpp[36 * i + 7 * j];
      s->Hpp[36 * i + 7 * j] += lambda;
    }
  for (int i = 0; i < s->L; ++i)
    for (int j = 0; j < 3; ++j) {
      s->diag_bak[s->sizePoses + 3 * i + j] = s->Hll[9 * i + 4 * j];
      s->Hll[9 * i + 4 * j] += lambda;
    }
}
static void restore_diagonal(ostate* s) {
  for (int i = 0; i < s->P; ++i)
    for (int j = 0; j < 6; ++j) s->Hpp[36 * i + 7 * j] = s->diag_bak[6 * i + j];
  for (int i = 0; i < s->L; ++i)
    for (int j = 0; j < 3; ++j) s->Hll[9 * i + 4 * j] = s->diag_bak[s->sizePoses + 3 * i + j];
}

/* Dense LDL^T (no pivoting, upper storage) + solve: admissible stand-in for
 * Eigen::SimplicialLDLT<Upper> behind LinearSolverEigen::solve
 * (g2o/solvers/linear_solver_eigen.h:94-124).  Fails
 * only on an exactly-zero pivot, like SimplicialLDLT's info(). */
int oracle_ldlt_solve(int n, double* A, const double* b, double* x, double* tmp) {
  for (int k = 0; k < n; ++k) {
    const double d = A[k * n + k];
    if (d == 0.0) return 0;
    double* rk = A + (size_t)k * n;
    for (int i = k + 1; i < n; ++i) tmp[i] = rk[i] / d; /* l_ik */
    for (int i = k + 1; i < n; ++i) {
      const double l = tmp[i];
      double* ri = A + (size_t)i * n;
      for (int j = i; j < n; ++j) ri[j] -= l * rk[j];
    }
    for (int i = k + 1; i < n; ++i) rk[i] = tmp[i];
  }
  /* L y = b */
  for (int i = 0; i < n; ++i) x[i] = b[i];
  for (int k = 0; k < n; ++k) {
    const double yk = x[k];
    const double* rk = A + (size_t)k * n;
    for (int i = k + 1; i < n; ++i) x[i] -= rk[i] * yk;
  }
  for (int k = 0; k < n; ++k) x[k] /= A[k * n + k];
  /* L^T x = z */
  for (int k = n - 1; k >= 0; --k) {
    const double* rk = A + (size_t)k * n;
    double sum = x[k];
    for (int i = k + 1; i < n; ++i) sum -= rk[i] * x[i];
    x[k] = sum;
  }
  return 1;
}

/* BlockSolver::solve (Schur branch), block_solver.hpp:367-486 */
static int block_solve(ostate* s) {
  const int n = s->sizePoses;
  /* _Hschur = _Hpp (upper blocks only exist: diagonal) */
  memset(s->Hschur, 0, sizeof(double) * (size_t)n * n);
  for (int i = 0; i < s->P; ++i)
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c)
        s->Hschur[(size_t)(6 * i + r) * n + 6 * i + c] = s->Hpp[36 * i + r * 6 + c];
  memset(s->coeff, 0, sizeof(double) * n);
  const double* bl = s->b + n;
  for (int j = 0; j < s->L; ++j) {
    double* Dinv = s->Dinv + 9 * j;
    mat3_inverse(s->Hll + 9 * j, Dinv);
    double db[3];
    for (int r = 0; r < 3; ++r)
      db[r] = Dinv[r * 3] * bl[3 * j] + Dinv[r * 3 + 1] * bl[3 * j + 1] + Dinv[r * 3 + 2] * bl[3 * j + 2];
    for (int a = s->col_off[j]; a < s->col_off[j + 1]; ++a) {
      const int i1 = s->blk_row[a];
      const double* Bi = s->Hpl + 18 * a;
      double BDinv[18];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 3; ++c)
          BDinv[r * 3 + c] = Bi[r * 3] * Dinv[c] + Bi[r * 3 + 1] * Dinv[3 + c] + Bi[r * 3 + 2] * Dinv[6 + c];
      for (int r = 0; r < 6; ++r)
        s->coeff[6 * i1 + r] += Bi[r * 3] * db[0] + Bi[r * 3 + 1] * db[1] + Bi[r * 3 + 2] * db[2];
      for (int a2 = a; a2 < s->col_off[j + 1]; ++a2) {
        const int i2 = s->blk_row[a2];
        const double* Bj = s->Hpl + 18 * a2;
        double* H = s->Hschur + (size_t)(6 * i1) * n + 6 * i2;
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c)
            H[(size_t)r * n + c] -= BDinv[r * 3] * Bj[c * 3] + BDinv[r * 3 + 1] * Bj[c * 3 + 1] + BDinv[r * 3 + 2] * Bj[c * 3 + 2];
      }
    }
  }
  for (int i = 0; i < n; ++i) s->bschur[i] = s->b[i] - s->coeff[i];
  if (!oracle_ldlt_solve(n, s->Hschur, s->bschur, s->x, s->ldl_tmp)) return 0;
  /* landmarks: cl = bl - Hpl^T xp ; xl = Dinv cl   (:461-483) */
  double* xl = s->x + n;
  double* cl = s->coeff + n;
  memcpy(cl, bl, sizeof(double) * s->sizeLm);
  for (int j = 0; j < s->L; ++j) {
    for (int a = s->col_off[j]; a < s->col_off[j + 1]; ++a) {
      /* SparseBlockMatrixCCS::rightMultiply with cp = -xp, sparse_block_matrix_ccs.h:103-129 */
      const double* Bi = s->Hpl + 18 * a;
      const double* xp = s->x + 6 * s->blk_row[a];
      for (int c = 0; c < 3; ++c) {
        double acc = 0;
        for (int r = 0; r < 6; ++r) acc += Bi[r * 3 + c] * (-xp[r]);
        cl[3 * j + c] += acc;
      }
    }
    const double* Dinv = s->Dinv + 9 * j;
    for (int r = 0; r < 3; ++r)
      xl[3 * j + r] = Dinv[r * 3] * cl[3 * j] + Dinv[r * 3 + 1] * cl[3 * j + 1] + Dinv[r * 3 + 2] * cl[3 * j + 2];
  }
  return 1;
}

/* SparseOptimizer::update, sparse_optimizer.cpp:422-435
 * (VertexSE3Expmap::oplusImpl, VertexSBAPointXYZ::oplusImpl types_sba.h:52-56) */
static void apply_update(ostate* s) {
  for (int i = 0; i < s->P; ++i) oracle_pose_oplus(s->x + 6 * i, s->qt + 7 * i);
  for (int i = 0; i < s->sizeLm; ++i) s->X[i] += s->x[s->sizePoses + i];
}

static int terminate_requested(const ostate* s) {
  return s->pr->stop_flag && *s->pr->stop_flag; /* sparse_optimizer.h:188 */
}

/* Build the CCS structure of Hpl (buildStructure, block_solver.hpp:143-295). */
static int build_structure(ostate* s) {
  const osh_lba_problem* p = s->pr;
  int* cnt = (int*)calloc((size_t)s->L + 1, sizeof(int));
  if (!cnt) return 0;
  for (int e = 0; e < s->E; ++e)
    if (p->edge_pose[e] < s->P) cnt[p->edge_point[e] + 1]++;
  for (int j = 0; j < s->L; ++j) cnt[j + 1] += cnt[j];
  const int nfree = cnt[s->L];
  int* tmp_pose = (int*)malloc(sizeof(int) * (size_t)(nfree > 0 ? nfree : 1));
  int* fill = (int*)malloc(sizeof(int) * (size_t)(s->L > 0 ? s->L : 1));
  for (int j = 0; j < s->L; ++j) fill[j] = cnt[j];
  for (int e = 0; e < s->E; ++e)
    if (p->edge_pose[e] < s->P) tmp_pose[fill[p->edge_point[e]]++] = p->edge_pose[e];
  /* sort rows of every column, unique -> blocks */
  s->col_off = (int*)malloc(sizeof(int) * ((size_t)s->L + 1));
  s->blk_row = (int*)malloc(sizeof(int) * (size_t)(nfree > 0 ? nfree : 1));
  int nb = 0;
  for (int j = 0; j < s->L; ++j) {
    s->col_off[j] = nb;
    int lo = cnt[j], hi = cnt[j + 1];
    for (int a = lo + 1; a < hi; ++a) { /* insertion sort, columns are short */
      int v = tmp_pose[a], b = a - 1;
      while (b >= lo && tmp_pose[b] > v) { tmp_pose[b + 1] = tmp_pose[b]; --b; }
      tmp_pose[b + 1] = v;
    }
    for (int a = lo; a < hi; ++a)
      if (a == lo || tmp_pose[a] != tmp_pose[a - 1]) s->blk_row[nb++] = tmp_pose[a];
  }
  s->col_off[s->L] = nb;
  s->nblk = nb;
  s->edge_blk = (int*)malloc(sizeof(int) * (size_t)(s->E > 0 ? s->E : 1));
  for (int e = 0; e < s->E; ++e) {
    s->edge_blk[e] = -1;
    if (p->edge_pose[e] < s->P) {
      const int j = p->edge_point[e];
      for (int a = s->col_off[j]; a < s->col_off[j + 1]; ++a)
        if (s->blk_row[a] == p->edge_pose[e]) { s->edge_blk[e] = a; break; }
    }
  }
  free(cnt); free(tmp_pose); free(fill);
  return 1;
}

static int validate(const osh_lba_problem* p) {
  if (!p || p->n_free < 0 || p->n_fixed < 0 || p->n_points < 0 || p->n_edges < 0) return 0;
  for (int e = 0; e < p->n_edges; ++e) {
    if (p->edge_pose[e] < 0 || p->edge_pose[e] >= p->n_free + p->n_fixed) return 0;
    if (p->edge_point[e] < 0 || p->edge_point[e] >= p->n_points) return 0;
    if (p->edge_kind[e] > OSH_EDGE_BODY) return 0;
    if (p->kb8 && p->edge_kind[e] == OSH_EDGE_STEREO) return 0;   /* a fisheye window has no rectified-stereo edges */
    if (p->edge_kind[e] == OSH_EDGE_BODY && !(p->kb8 && p->cam2 && p->trl)) return 0;
  }
  return 1;
}

static int state_init(ostate* s, const osh_lba_problem* p) {
  memset(s, 0, sizeof(*s));
  s->pr = p;
  s->P = p->n_free; s->F = p->n_fixed; s->L = p->n_points; s->E = p->n_edges;
  s->sizePoses = 6 * s->P; s->sizeLm = 3 * s->L;
  const size_t NP = (size_t)(s->P + s->F), n = (size_t)s->sizePoses, N = n + s->sizeLm;
#define ALLOC(ptr, count) ptr = (double*)calloc((count) > 0 ? (count) : 1, sizeof(double)); if (!(ptr)) return 0;
  ALLOC(s->qt, NP * 7) ALLOC(s->X, (size_t)s->L * 3) ALLOC(s->qt_bak, NP * 7) ALLOC(s->X_bak, (size_t)s->L * 3)
  ALLOC(s->err, (size_t)s->E * 3) ALLOC(s->Hpp, (size_t)s->P * 36) ALLOC(s->Hll, (size_t)s->L * 9)
  ALLOC(s->b, N) ALLOC(s->Hschur, n * n) ALLOC(s->bschur, n) ALLOC(s->coeff, N) ALLOC(s->x, N)
  ALLOC(s->Dinv, (size_t)s->L * 9) ALLOC(s->diag_bak, N) ALLOC(s->ldl_tmp, n)
  memcpy(s->qt, p->pose_qt, sizeof(double) * NP * 7);
  memcpy(s->X, p->points, sizeof(double) * (size_t)s->L * 3);
  /* g2o::SE3Quat(q,t) constructor normalises, se3quat.h:61-63 (Optimizer.cc:1218,1237) */
  for (size_t i = 0; i < NP; ++i) quat_normalize_rotation(s->qt + 7 * i);
  if (!build_structure(s)) return 0;
  ALLOC(s->Hpl, (size_t)s->nblk * 18)
#undef ALLOC
  return 1;
}

static void state_free(ostate* s) {
  free(s->qt); free(s->X); free(s->qt_bak); free(s->X_bak); free(s->err); free(s->Hpp); free(s->Hll);
  free(s->Hpl); free(s->b); free(s->edge_blk); free(s->col_off); free(s->blk_row); free(s->Hschur);
  free(s->bschur); free(s->coeff); free(s->x); free(s->Dinv); free(s->diag_bak); free(s->ldl_tmp);
}

/* One linearisation at the initial estimates; blocks in the layout documented
 * for osh_lba_linearize (include/orbslam3_hip.h). */
int oracle_lba_linearize(const osh_lba_problem* p, double* Hpp, double* bp, double* Hll, double* bl,
                         double* Hpl, double* chi2, double* robust_chi2) {
  if (!validate(p)) return OSH_ERR_INVALID;
  ostate s;
  if (!state_init(&s, p)) { state_free(&s); return OSH_ERR_INVALID; }
  compute_active_errors(&s);
  if (robust_chi2) *robust_chi2 = active_robust_chi2(&s);
  build_system(&s);
  if (Hpp) memcpy(Hpp, s.Hpp, sizeof(double) * 36 * (size_t)s.P);
  if (bp) memcpy(bp, s.b, sizeof(double) * (size_t)s.sizePoses);
  if (Hll) memcpy(Hll, s.Hll, sizeof(double) * 9 * (size_t)s.L);
  if (bl) memcpy(bl, s.b + s.sizePoses, sizeof(double) * (size_t)s.sizeLm);
  if (Hpl) {
    /* per input edge: only meaningful when no (pose,point) pair repeats */
    memset(Hpl, 0, sizeof(double) * 18 * (size_t)s.E);
    for (int e = 0; e < s.E; ++e)
      if (s.edge_blk[e] >= 0) memcpy(Hpl + 18 * (size_t)e, s.Hpl + 18 * (size_t)s.edge_blk[e], sizeof(double) * 18);
  }
  if (chi2) for (int e = 0; e < s.E; ++e) chi2[e] = edge_chi2(&s, e);
  state_free(&s);
  return OSH_OK;
}

/* One LM trial at the initial estimates with a given lambda: returns the
 * Schur complement (dense, upper triangle valid), reduced rhs and the
 * solution vector x (6P+3L).  Parity/debug aid. */
int oracle_lba_schur_step(const osh_lba_problem* p, double lambda, double* S, double* bs, double* x) {
  if (!validate(p)) return OSH_ERR_INVALID;
  ostate s;
  if (!state_init(&s, p)) { state_free(&s); return OSH_ERR_INVALID; }
  compute_active_errors(&s);
  build_system(&s);
  set_lambda(&s, lambda);
  const int n = s.sizePoses;
  /* run the Schur part separately to export S before it is factorised */
  int ok = 1;
  {
    /* duplicate of block_solve's first half so S can be copied out */
    double* keep = (double*)malloc(sizeof(double) * (size_t)n * n);
    ok = block_solve(&s);
    /* recompute S for export (block_solve overwrote it with the factor) */
    if (S) {
      memset(keep, 0, sizeof(double) * (size_t)n * n);
      for (int i = 0; i < s.P; ++i)
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c) keep[(size_t)(6 * i + r) * n + 6 * i + c] = s.Hpp[36 * i + r * 6 + c];
      for (int j = 0; j < s.L; ++j) {
        const double* Dinv = s.Dinv + 9 * j;
        for (int a = s.col_off[j]; a < s.col_off[j + 1]; ++a) {
          const double* Bi = s.Hpl + 18 * a;
          double BDinv[18];
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 3; ++c)
              BDinv[r * 3 + c] = Bi[r * 3] * Dinv[c] + Bi[r * 3 + 1] * Dinv[3 + c] + Bi[r * 3 + 2] * Dinv[6 + c];
          for (int a2 = a; a2 < s.col_off[j + 1]; ++a2) {
            const double* Bj = s.Hpl + 18 * a2;
            double* H = keep + (size_t)(6 * s.blk_row[a]) * n + 6 * s.blk_row[a2];
            for (int r = 0; r < 6; ++r)
              for (int c = 0; c < 6; ++c)
                H[(size_t)r * n + c] -= BDinv[r * 3] * Bj[c * 3] + BDinv[r * 3 + 1] * Bj[c * 3 + 1] + BDinv[r * 3 + 2] * Bj[c * 3 + 2];
          }
        }
      }
      memcpy(S, keep, sizeof(double) * (size_t)n * n);
    }
    free(keep);
  }
  if (bs) memcpy(bs, s.bschur, sizeof(double) * (size_t)n);
  if (x) memcpy(x, s.x, sizeof(double) * (size_t)(n + s.sizeLm));
  state_free(&s);
  return ok ? OSH_OK : OSH_ERR_INVALID;
}

/* SparseOptimizer::optimize, sparse_optimizer.cpp:354-419, with
 * OptimizationAlgorithmLevenberg::solve, levenberg.cpp:61-169. */
int oracle_lba_solve(const osh_lba_problem* p, osh_lba_result* res) {
  if (!validate(p) || !res) return OSH_ERR_INVALID;
  ostate s;
  if (!state_init(&s, p)) { state_free(&s); return OSH_ERR_INVALID; }
  const size_t NP7 = (size_t)(s.P + s.F) * 7;
  double lambda = -1., ni = 2.;
  int nBad = 0, cjIterations = 0, trials_total = 0, ok = 1;
  res->n_trace = 0; res->chi2_initial = 0;
  const int maxTrialsAfterFailure = 10; /* levenberg.cpp:50 */
  for (int it = 0; it < p->max_iterations && !terminate_requested(&s) && ok; ++it) {
    compute_active_errors(&s);
    double currentChi = active_robust_chi2(&s);
    double tempChi = currentChi;
    const double iniChi = currentChi;
    if (it == 0) res->chi2_initial = currentChi;
    build_system(&s);
    if (it == 0) { lambda = compute_lambda_init(&s); ni = 2; nBad = 0; }
    double rho = 0;
    int qmax = 0;
    do {
      memcpy(s.qt_bak, s.qt, sizeof(double) * NP7);           /* push */
      memcpy(s.X_bak, s.X, sizeof(double) * (size_t)s.sizeLm);
      set_lambda(&s, lambda);
      const int ok2 = block_solve(&s);
      apply_update(&s);
      restore_diagonal(&s);
      compute_active_errors(&s);
      tempChi = active_robust_chi2(&s);
      if (!ok2) tempChi = DBL_MAX;
      rho = (currentChi - tempChi);
      double scale = 0.; /* computeScale, levenberg.cpp:187-194 */
      for (int j = 0; j < s.sizePoses + s.sizeLm; ++j) scale += s.x[j] * (lambda * s.x[j] + s.b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        const double scaleFactor = fmax(1. / 3., alpha);
        lambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        memcpy(s.qt, s.qt_bak, sizeof(double) * NP7);          /* pop */
        memcpy(s.X, s.X_bak, sizeof(double) * (size_t)s.sizeLm);
      }
      qmax++;
      trials_total++;
    } while (rho < 0 && qmax < maxTrialsAfterFailure && !terminate_requested(&s));
    ++cjIterations;
    if (res->n_trace < OSH_LBA_MAX_TRACE) {
      res->chi2_trace[res->n_trace] = currentChi;
      res->lambda_trace[res->n_trace] = lambda;
      res->trials_trace[res->n_trace] = qmax;
      res->n_trace++;
    }
    if (qmax == maxTrialsAfterFailure || rho == 0) { ok = 0; continue; } /* Terminate */
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;  /* Raul's stop */
    if (nBad >= 3) { ok = 0; continue; }
  }
  res->iterations = cjIterations;
  res->trials = trials_total;
  res->status = OSH_OK;
  if (res->pose_qt) memcpy(res->pose_qt, s.qt, sizeof(double) * 7 * (size_t)s.P);
  if (res->points) memcpy(res->points, s.X, sizeof(double) * (size_t)s.sizeLm);
  /* e->chi2() uses the last evaluated _error (stale after a rejected final
   * trial), isDepthPositive() recomputes from the estimates: Optimizer.cc:1425 */
  if (res->edge_chi2) for (int e = 0; e < s.E; ++e) res->edge_chi2[e] = edge_chi2(&s, e);
  if (res->edge_depth_pos)
    for (int e = 0; e < s.E; ++e)
      res->edge_depth_pos[e] = (uint8_t)(p->edge_kind[e] == OSH_EDGE_BODY
          ? oracle_edge_depth_positive_body(s.qt + 7 * p->edge_pose[e], p->trl, s.X + 3 * p->edge_point[e])
          : oracle_edge_depth_positive(s.qt + 7 * p->edge_pose[e], s.X + 3 * p->edge_point[e]));
  state_free(&s);
  return OSH_OK;
}
