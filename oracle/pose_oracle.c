/*
 * pose_oracle.c -- CPU restatement of the solver part of ORB_SLAM3::Optimizer::PoseOptimization
 * (src/Optimizer.cc:815-1114).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see lba_oracle.c header).  PARITY UNPINNED against a
 * reference binary (the reference cannot be built here); pinned by the same edge functions that
 * lba_oracle.c cross-checks against numpy and central differences, and by property tests (zero-noise
 * recovery, outliers rejected) in tests/.
 *
 * One VertexSE3Expmap, unary edges EdgeSE3ProjectXYZOnlyPose (src/OptimizableTypes.cpp:49-61: the pose part
 * of the binary mono edge) / g2o::EdgeStereoSE3ProjectXYZOnlyPose (types_six_dof_expmap.cpp:306-405: the pose
 * part of the binary stereo edge, float invz in the residual).  No marginalised vertex, so BlockSolver skips
 * the Schur branch and hands (Hpp + lambda I) x = b to LinearSolverDense (Eigen::LDLT, solve() fails unless
 * the factor is positive; g2o/solvers/linear_solver_dense.h:97-105).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

typedef struct {
  const osh_pose_problem* p;
  double qt[7];
  unsigned char* level;   /* 1: outlier of the previous round, not in the active set */
  double* err;            /* [E*3] _error of every edge as last computed */
  int robust;
} pstate;

static double edge_chi2(const pstate* s, int e) {
  const double* r = s->err + 3 * (size_t)e;
  const double w = s->p->edge_info[e];
  return r[0] * (w * r[0]) + r[1] * (w * r[1]) + r[2] * (w * r[2]);
}

static void compute_error(pstate* s, int e) {
  const osh_pose_problem* p = s->p;
  if (p->edge_kind[e] == OSH_EDGE_BODY)   /* EdgeSE3ProjectXYZOnlyPoseToBody (include/OptimizableTypes.h:62-87): same expressions as the binary edge */
    oracle_edge_error_body(s->qt, p->cam2, p->trl, p->points + 3 * (size_t)e, p->edge_obs + 3 * (size_t)e, s->err + 3 * (size_t)e);
  else if (p->kb8 && p->edge_kind[e] == OSH_EDGE_MONO)
    oracle_edge_error_kb8(s->qt, p->cam, p->kb8, p->points + 3 * (size_t)e, p->edge_obs + 3 * (size_t)e, s->err + 3 * (size_t)e);
  else
    oracle_edge_error(p->edge_kind[e], s->qt, p->cam, p->points + 3 * (size_t)e, p->edge_obs + 3 * (size_t)e, s->err + 3 * (size_t)e);
}

static void compute_active_errors(pstate* s) {
  for (int e = 0; e < s->p->n_edges; ++e) if (!s->level[e]) compute_error(s, e);
}

static double active_robust_chi2(const pstate* s) {
  double chi = 0;
  for (int e = 0; e < s->p->n_edges; ++e) {
    if (s->level[e]) continue;
    const double c = edge_chi2(s, e);
    if (s->robust) {
      double rho[3];
      oracle_huber(c, s->p->edge_kind[e] != OSH_EDGE_STEREO ? s->p->huber_mono : s->p->huber_stereo, rho);
      chi += rho[0];
    } else chi += c;
  }
  return chi;
}

/* Hpp (6x6 full) and b of the active edges, base_unary_edge.hpp constructQuadraticForm */
static void build_system(pstate* s, double H[36], double b[6]) {
  const osh_pose_problem* p = s->p;
  memset(H, 0, sizeof(double) * 36); memset(b, 0, sizeof(double) * 6);
  for (int e = 0; e < p->n_edges; ++e) {
    if (s->level[e]) continue;
    double JX[9], Jp[18];
    if (p->edge_kind[e] == OSH_EDGE_BODY) oracle_edge_jacobians_body(s->qt, p->cam2, p->trl, p->points + 3 * (size_t)e, JX, Jp);
    else if (p->kb8 && p->edge_kind[e] == OSH_EDGE_MONO) oracle_edge_jacobians_kb8(s->qt, p->cam, p->kb8, p->points + 3 * (size_t)e, JX, Jp);
    else oracle_edge_jacobians(p->edge_kind[e], s->qt, p->cam, p->points + 3 * (size_t)e, JX, Jp);
    const double* r = s->err + 3 * (size_t)e;
    const double info = p->edge_info[e];
    double rho1 = 1.0;
    if (s->robust) {
      double rho[3];
      oracle_huber(edge_chi2(s, e), p->edge_kind[e] != OSH_EDGE_STEREO ? p->huber_mono : p->huber_stereo, rho);
      rho1 = rho[1];
    }
    const double ww = rho1 * info;
    const int d = p->edge_kind[e] != OSH_EDGE_STEREO ? 2 : 3;
    for (int a = 0; a < 6; ++a) {
      for (int c = 0; c < 6; ++c) {
        double t = 0;
        for (int k = 0; k < d; ++k) t += (Jp[k * 6 + a] * ww) * Jp[k * 6 + c];
        H[a * 6 + c] += t;
      }
      double t = 0;
      for (int k = 0; k < d; ++k) t += Jp[k * 6 + a] * (-(info * r[k]) * rho1);
      b[a] += t;
    }
  }
}

/* LinearSolverDense::solve: LDL^T of the 6x6 system; false unless every pivot is positive (isPositive()). */
static int dense_solve(const double H[36], const double b[6], double x[6]) {
  double A[36], tmp[6];
  memcpy(A, H, sizeof(A));
  if (!oracle_ldlt_solve(6, A, b, x, tmp)) return 0;
  for (int k = 0; k < 6; ++k) if (!(A[k * 6 + k] > 0)) return 0;
  return 1;
}

/* SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve (sparse_optimizer.cpp:354-419,
 * optimization_algorithm_levenberg.cpp:61-169) on the single pose vertex. */
static int optimize(pstate* s, int max_iterations, double* chi_out) {
  double lambda = -1., ni = 2.;
  int nBad = 0, cj = 0, ok = 1;
  double last_chi = 0;
  int any = 0;
  for (int e = 0; e < s->p->n_edges; ++e) any |= !s->level[e];
  if (!any) { *chi_out = 0; return 0; }   /* initializeOptimization: no active edge, optimize() returns -1 */
  for (int it = 0; it < max_iterations && ok; ++it) {
    compute_active_errors(s);
    double currentChi = active_robust_chi2(s);
    double tempChi = currentChi;
    const double iniChi = currentChi;
    double H[36], b[6], x[6];
    build_system(s, H, b);
    if (it == 0) {
      double m = 0;
      for (int k = 0; k < 6; ++k) m = fmax(m, fabs(H[k * 6 + k]));
      lambda = 1e-5 * m; ni = 2; nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      double bak[7];
      memcpy(bak, s->qt, sizeof(bak));
      double Hl[36];
      memcpy(Hl, H, sizeof(Hl));
      for (int k = 0; k < 6; ++k) Hl[k * 6 + k] += lambda;
      const int ok2 = dense_solve(Hl, b, x);
      if (!ok2) memset(x, 0, sizeof(x));   /* the solver leaves x untouched (zero-initialised) on failure */
      oracle_pose_oplus(x, s->qt);
      compute_active_errors(s);
      tempChi = active_robust_chi2(s);
      if (!ok2) tempChi = DBL_MAX;
      rho = (currentChi - tempChi);
      double scale = 0.;
      for (int j = 0; j < 6; ++j) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        lambda *= fmax(1. / 3., alpha);
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni; ni *= 2;
        memcpy(s->qt, bak, sizeof(bak));
      }
      qmax++;
    } while (rho < 0 && qmax < 10);
    ++cj;
    last_chi = currentChi;
    if (qmax == 10 || rho == 0) { ok = 0; continue; }
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) { ok = 0; continue; }
  }
  *chi_out = last_chi;
  return cj;
}

int oracle_pose_optimize(const osh_pose_problem* p, osh_pose_result* res) {
  if (!p || !res || p->n_edges < 0) return OSH_ERR_INVALID;
  pstate s;
  s.p = p; s.robust = 1;
  s.level = (unsigned char*)calloc((size_t)p->n_edges + 1, 1);
  s.err = (double*)calloc((size_t)p->n_edges * 3 + 1, sizeof(double));
  unsigned char* outlier = (unsigned char*)calloc((size_t)p->n_edges + 1, 1);
  int nBad = 0;
  res->rounds = 0;
  memcpy(s.qt, p->pose_qt, sizeof(double) * 7);
  for (int it = 0; it < 4; ++it) {
    memcpy(s.qt, p->pose_qt, sizeof(double) * 7);   /* vSE3->setEstimate(pFrame->GetPose()) every round (:1024-1025) */
    {  /* g2o::SE3Quat(q, t) normalises the rotation */
      double* q = s.qt;
      if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
      const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      for (int k = 0; k < 4; ++k) q[k] /= n;
    }
    res->iterations[it] = optimize(&s, p->iterations[it], &res->chi2_final[it]);
    res->rounds = it + 1;
    nBad = 0;
    for (int e = 0; e < p->n_edges; ++e) {
      if (outlier[e]) compute_error(&s, e);         /* level-1 edges were not evaluated by the optimiser (:1036-1039) */
      const float chi2 = (float)edge_chi2(&s, e);    /* const float chi2 = e->chi2() */
      const float th = p->edge_kind[e] != OSH_EDGE_STEREO ? p->chi2_mono[it] : p->chi2_stereo[it];   /* right-camera edges: chi2Mono (:1053) */
      if (res->edge_chi2) res->edge_chi2[e] = edge_chi2(&s, e);
      if (chi2 > th) { outlier[e] = 1; s.level[e] = 1; nBad++; }
      else { outlier[e] = 0; s.level[e] = 0; }
    }
    if (it == 2) s.robust = 0;                       /* e->setRobustKernel(0) (:1054) */
    if (p->n_edges < 10) break;                      /* optimizer.edges().size() < 10 (:1107) */
  }
  memcpy(res->pose_qt, s.qt, sizeof(double) * 7);
  if (res->outlier) memcpy(res->outlier, outlier, (size_t)p->n_edges);
  res->n_bad = nBad;
  res->status = OSH_OK;
  free(s.level); free(s.err); free(outlier);
  return OSH_OK;
}
