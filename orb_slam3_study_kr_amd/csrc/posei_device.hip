// posei_device.hip -- Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame (src/Optimizer.cc:4499-5299) on MI355X (gfx950).
//
// 15 unknowns (pose, velocity, gyro and accelerometer bias of the tracked frame; mode 0) or 30 (the previous frame's as well;
// mode 1), O(10^3) unary visual edges, one EdgeInertial, two random-walk edges and, in mode 1, the EdgePriorPoseImu of the
// previous frame.  Gauss-Newton (g2o/core/optimization_algorithm_gauss_newton.cpp:49-90): every iteration computes the errors,
// builds the dense system, solves it (LinearSolverDense: LDL^T that must be positive, linear_solver_dense.h:60-118) and applies
// the update; nothing is re-evaluated after the last update, so the classification after each of the four rounds sees the
// errors of the START of the round's last iteration (an outlier is re-evaluated at the final estimate, :4763-4766).
//
// Like the pose-only kernel (pose_device.hip) the problem is tiny and latency-critical: ONE block per frame runs all four
// rounds -- thread per visual edge with fixed-order block reductions, the small dense algebra of the inertial / prior edges
// spread over the block, the 15x15 / 30x30 LDL^T in one wavefront with the pivot row broadcast by v_readlane.
#include "common.h"
#include "lba_math.h"
#include "ldlt_block.h"
#include "liba_math.h"
#include "liba_edges.h"
#include <cfloat>
#include <cstring>
#include <mutex>
#include <algorithm>
#include <vector>

namespace osh {

constexpr int kIT = 256;   // threads per frame block

struct PoseiDesc {
  LibaDesc cam;              // camera side only: Rcb tcb tbc cam kb8 rig (the edge functions of liba_edges.h read it)
  int mode, E, edge_off, rec_init;
  double P[24], s[9];        // current frame: pose record Rcw tcw Rwb twb, then v bg ba
  double pP[24], ps[9];      // previous state (Rcw / tcw unused)
  float rec[OSH_PREINT_FLOATS];
  double info[81], info_g[9], info_a[9];
  double prior_R[9], prior_t[3], prior_s[9], prior_H[225];
  double huber_mono, huber_stereo, huber_prior;
  float chi2_mono[4], chi2_stereo[4];
  int iters[4];
};
struct PoseiOut { double P[24], s[9], H[900]; int n_bad, n_inliers, rounds; };
struct PoseiView {
  const PoseiDesc* desc;
  PoseiOut* out;
  const double* X; const unsigned char* kind; const double* obs; const double* info; const unsigned char* close;
  double* chi2; unsigned char* level; unsigned char* outlier;
  int ecap;                   // edges per frame the block's dynamic LDS can hold (0: every access goes to global memory)
};

__device__ __forceinline__ double posei_block_sum(double v, double* sh) {
  v = dev::wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < kIT / 64; ++k) t += sh[k];
  return t;
}

// N block sums with TWO barriers: butterfly inside each wavefront, the wavefront partials parked in LDS ([kIT/64][N]) and added in
// wavefront order by every thread (deterministic).  27 single sums cost 54 barriers per Gauss-Newton iteration before.
template <int N>
__device__ __forceinline__ void posei_block_sum_n(double* v, double* shn) {
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = dev::wave_sum(v[k]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) shn[(threadIdx.x >> 6) * N + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kIT / 64; ++w) t += shn[w * N + k];
    v[k] = t;
  }
}

// A x = b for a symmetric n x n system (n <= NMAX = 16 or 32: 15 unknowns of the frame, 30 with the previous frame free) held row-major in LDS, by ONE wavefront: lane j keeps column j, the pivot row
// is broadcast with v_readlane; returns false unless every pivot is positive (Eigen::LDLT::isPositive).  U: 32 x 33 doubles of
// LDS scratch for the unit upper factor; x in LDS.
template <int NMAX>
__device__ bool posei_solve_wave(const double* A, const double* b, int n, double* U, double* x) {
  const int lane = threadIdx.x & 63;
  const int j = lane < NMAX ? lane : 0;
  double col[NMAX];
#pragma unroll
  for (int r = 0; r < NMAX; ++r) {
    const bool in = r < n && j < n && r <= j;
    const double a = A[in ? r * n + j : 0];
    col[r] = in ? a : (r == j ? 1.0 : 0.0);
  }
  double zr = lane < n ? b[lane] : 0.0;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    const double d = ldlt_readlane(col[k], k);
    bad |= !(d > 0.0);
    const double lk = col[k] / d;                       // l_kj in lane j (j > k)
    const double zk = ldlt_readlane(zr, k);             // z_k is final once the steps before k have been applied
    if (lane > k) zr -= lk * zk;
    if (lane < NMAX && lane > k) U[k * 33 + lane] = lk;
#pragma unroll
    for (int i = k + 1; i < NMAX; ++i) col[i] -= ldlt_readlane(lk, i) * col[k];
  }
  if (bad) return false;
  // w = z / d, then U x = w from the last column to the first: lane r holds its running entry, the solved one is broadcast
  double dl = 1.0;
#pragma unroll
  for (int r = 0; r < NMAX; ++r) { const double dr = ldlt_readlane(col[r], r); if (r == lane) dl = dr; }
  double t = zr / dl;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int c = NMAX - 1; c > 0; --c) {
    const double xc = ldlt_readlane(t, c);
    if (lane < c) t -= U[lane * 33 + c] * xc;
  }
  if (lane < n) x[lane] = t;
  return true;
}

__global__ __launch_bounds__(kIT) void k_posei(PoseiView v) {
  __shared__ double sh[kIT / 64];
  __shared__ double shn[(kIT / 64) * 27];
  __shared__ double shP[24], shs[9], shpP[24], shps[9];     // current / previous state
  __shared__ double shH[900], shb[30], shx[30];
  __shared__ double shJ[9 * 24], shWJ[15 * 30], shr[15], shJp[225];
  __shared__ double shU[32 * 33];
  __shared__ double shred[27];                 // the 27 visual sums, looked up by entry (a register array indexed at run time lives in scratch)
  __shared__ float shrec[OSH_PREINT_FLOATS];   // the preintegration record: read a dozen times per iteration by one thread
  __shared__ int sh_ok;
  const PoseiDesc& d = v.desc[blockIdx.x];
  PoseiOut& out = v.out[blockIdx.x];
  const int tid = threadIdx.x;
  const bool mode1 = d.mode == 1;
  const int n = mode1 ? 30 : 15;
  // The frame's edges (inputs, level, outlier flag, the chi2 computed last) live in LDS for the whole optimisation when they fit: the
  // 40 Gauss-Newton iterations otherwise pay global-memory round trips for data that never change.  Layout: a plane per field.
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  const int cap = v.ecap;
  const bool cached = cap > 0 && d.E <= cap;
  double* const sX = dyn; double* const sObs = dyn + 3 * (size_t)cap; double* const sInfo = dyn + 6 * (size_t)cap; double* const sChi = dyn + 7 * (size_t)cap;
  unsigned char* const sKind = reinterpret_cast<unsigned char*>(dyn + 8 * (size_t)cap);
  unsigned char* const sClose = sKind + cap; unsigned char* const sLevel = sClose + cap; unsigned char* const sOutlier = sLevel + cap;
  const size_t e0 = (size_t)d.edge_off;
  auto eX = [&](int e, int k) { return cached ? sX[k * cap + e] : v.X[(e0 + e) * 3 + k]; };
  auto eObs = [&](int e, int k) { return cached ? sObs[k * cap + e] : v.obs[(e0 + e) * 3 + k]; };
  auto eInfo = [&](int e) { return cached ? sInfo[e] : v.info[e0 + e]; };
  auto eKind = [&](int e) { return (int)(cached ? sKind[e] : v.kind[e0 + e]); };
  auto eClose = [&](int e) { return (cached ? sClose[e] : v.close[e0 + e]) != 0; };
  auto eLevel = [&](int e) { return (cached ? sLevel[e] : v.level[e0 + e]) != 0; };
  auto eOutlier = [&](int e) { return (cached ? sOutlier[e] : v.outlier[e0 + e]) != 0; };
  auto eChi = [&](int e) { return cached ? sChi[e] : v.chi2[e0 + e]; };
  auto setChi = [&](int e, double c) { if (cached) sChi[e] = c; else v.chi2[e0 + e] = c; };
  auto setLevel = [&](int e, bool l) { if (cached) sLevel[e] = l ? 1 : 0; else v.level[e0 + e] = l ? 1 : 0; };
  auto setOutlier = [&](int e, bool o) { if (cached) sOutlier[e] = o ? 1 : 0; else v.outlier[e0 + e] = o ? 1 : 0; };
  for (int e = tid; e < d.E; e += kIT) {
    if (cached) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { sX[k * cap + e] = v.X[(e0 + e) * 3 + k]; sObs[k * cap + e] = v.obs[(e0 + e) * 3 + k]; }
      sInfo[e] = v.info[e0 + e]; sKind[e] = v.kind[e0 + e]; sClose[e] = v.close[e0 + e];
    }
    setLevel(e, false); setOutlier(e, false); setChi(e, 0.0);
  }
  if (tid < 24) { shP[tid] = d.P[tid]; shpP[tid] = d.pP[tid]; }
  if (tid < 9) { shs[tid] = d.s[tid]; shps[tid] = d.ps[tid]; }
  if (tid < 30) shx[tid] = 0.0;
  if (tid < OSH_PREINT_FLOATS) shrec[tid] = d.rec[tid];
  __syncthreads();
  // column of the EdgeInertial Jacobian (P1 V1 G1 A1 P2 V2) -> unknown (-1: fixed vertex)
  auto unk_of = [&](int c) { return c >= 15 ? c - 15 : (mode1 ? 15 + c : -1); };
  // ---- one computeActiveErrors + buildSystem: shH / shb; the errors of the active visual edges stay in v.chi2
  auto build = [&](bool robust) {
    for (int k = tid; k < n * n; k += kIT) shH[k] = 0.0;
    if (tid < n) shb[tid] = 0.0;
    double pose[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) pose[k] = shP[k];
    double H[21], b[6];
#pragma unroll
    for (int k = 0; k < 21; ++k) H[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = 0.0;
    // the visual edges run on the first wavefronts while lane 0 of the last one forms the inertial edge (one long chain of
    // small matrix products): the two take about the same time
    constexpr int kVis = kIT - 64;
    for (int e = tid < kVis ? tid : d.E; e < d.E; e += kVis) {
      if (eLevel(e)) continue;
      const int kind = eKind(e);
      const double info = eInfo(e);
      double X[3], obs[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { X[k] = eX(e, k); obs[k] = eObs(e, k); }
      VisEval ev;
      vis_residual(d.cam, kind, pose, X, obs, info, ev);
      setChi(e, ev.chi2);
      double r0 = ev.chi2, r1 = 1.0, JX[9], Jp[18];
      if (robust) dev::huber(ev.chi2, kind == OSH_EDGE_STEREO ? d.huber_stereo : d.huber_mono, r0, r1);
      vis_jacobians(d.cam, kind, pose, ev.Xc, JX, Jp);
      const double ww = r1 * info;
      const double wr[3] = {-(info * ev.r[0]) * r1, -(info * ev.r[1]) * r1, -(info * ev.r[2]) * r1};
      int m = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int c = a; c < 6; ++c) { H[m] += (Jp[a] * ww) * Jp[c] + (Jp[6 + a] * ww) * Jp[6 + c] + (Jp[12 + a] * ww) * Jp[12 + c]; ++m; }
        b[a] += Jp[a] * wr[0] + Jp[6 + a] * wr[1] + Jp[12 + a] * wr[2];
      }
    }
    // inertial edge: residual and Jacobian by one thread, the quadratic form by the block
    if (tid == kVis) {
      inertial_residual_jacobian(shrec, shpP, shps, shP, shs, shr, shJ);
    }
    __syncthreads();
    for (int idx = tid; idx < 9 * 24; idx += kIT) {   // W J
      const int k = idx / 24, c = idx - k * 24;
      double t = 0.0;
      for (int m = 0; m < 9; ++m) t += d.info[k * 9 + m] * shJ[m * 24 + c];
      shWJ[idx] = t;
    }
    __syncthreads();
    // visual block sums (27 fixed-order reductions), then everything else is added by the threads that own an entry
    double red[27];
#pragma unroll
    for (int k = 0; k < 21; ++k) red[k] = H[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) red[21 + k] = b[k];
    posei_block_sum_n<27>(red, shn);
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 27; ++k) shred[k] = red[k];
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += kIT) {
      const int r = idx / n, c = idx - r * n;
      double acc = 0.0;
      if (r < 6 && c < 6) {   // visual edges: upper-triangle sums mirrored
        const int a = r < c ? r : c, bq = r < c ? c : r;
        acc += shred[a * 6 - a * (a - 1) / 2 + (bq - a)];
      }
      // random walks: r = b_cur - b_prev, J = [-I, I]
      for (int which = 0; which < 2; ++which) {
        const double* Og = which == 0 ? d.info_g : d.info_a;
        const int o2 = 9 + 3 * which, o1 = mode1 ? 24 + 3 * which : -100;
        const bool r2 = r >= o2 && r < o2 + 3, c2 = c >= o2 && c < o2 + 3, r1 = r >= o1 && r < o1 + 3, c1 = c >= o1 && c < o1 + 3;
        if (r2 && c2) acc += Og[(r - o2) * 3 + (c - o2)];
        if (r1 && c1) acc += Og[(r - o1) * 3 + (c - o1)];
        if (r1 && c2) acc -= Og[(r - o1) * 3 + (c - o2)];
        if (r2 && c1) acc -= Og[(c - o1) * 3 + (r - o2)];
      }
      shH[idx] = acc;
    }
    __syncthreads();
    // EdgeInertial: J^T W J, a thread per pair of Jacobian columns (a column maps to at most one unknown, so no two pairs meet)
    for (int idx = tid; idx < 24 * 24; idx += kIT) {
      const int ca = idx / 24, cb = idx - ca * 24;
      const int r = unk_of(ca), c = unk_of(cb);
      if (r < 0 || c < 0) continue;
      double t = 0.0;
      for (int k = 0; k < 9; ++k) t += shJ[k * 24 + ca] * shWJ[k * 24 + cb];
      shH[r * n + c] += t;
    }
    if (tid < n) {
      double acc = tid < 6 ? shred[21 + tid] : 0.0;
      for (int ca = 0; ca < 24; ++ca) {
        if (unk_of(ca) != tid) continue;
        double t = 0.0;
        for (int k = 0; k < 9; ++k) t += shWJ[k * 24 + ca] * shr[k];      // J^T (W r) = (W J)^T r, W symmetric
        acc -= t;
      }
      for (int which = 0; which < 2; ++which) {
        const double* Og = which == 0 ? d.info_g : d.info_a;
        const int o2 = 9 + 3 * which, o1 = mode1 ? 24 + 3 * which : -100;
        double rb[3];
        for (int k = 0; k < 3; ++k) rb[k] = shs[3 + 3 * which + k] - shps[3 + 3 * which + k];
        if (tid >= o2 && tid < o2 + 3) { const int i = tid - o2; acc -= Og[i * 3] * rb[0] + Og[i * 3 + 1] * rb[1] + Og[i * 3 + 2] * rb[2]; }
        if (tid >= o1 && tid < o1 + 3) { const int i = tid - o1; acc += Og[i * 3] * rb[0] + Og[i * 3 + 1] * rb[1] + Og[i * 3 + 2] * rb[2]; }
      }
      shb[tid] = acc;
    }
    __syncthreads();
    if (mode1) {
      // EdgePriorPoseImu (src/G2oTypes.cc:731-763), Huber(huber_prior): residual / Jacobian by one thread, J^T W J by the block
      if (tid == 0) {
        double T[9], dd[3], invJr[9];
        imu::m3_tmul(d.prior_R, shpP + 12, T);
        imu::log_so3(T, shr);
        for (int i = 0; i < 3; ++i) dd[i] = shpP[21 + i] - d.prior_t[i];
        imu::m3_tvec(d.prior_R, dd, shr + 3);
        for (int i = 0; i < 9; ++i) shr[6 + i] = shps[i] - d.prior_s[i];
        imu::inv_right_jac(shr, invJr);
        for (int i = 0; i < 225; ++i) shJp[i] = 0.0;
        for (int i = 0; i < 3; ++i) for (int jq = 0; jq < 3; ++jq) { shJp[i * 15 + jq] = invJr[i * 3 + jq]; shJp[(3 + i) * 15 + 3 + jq] = T[i * 3 + jq]; }
        for (int i = 6; i < 15; ++i) shJp[i * 15 + i] = 1.0;
      }
      __syncthreads();
      double chi = 0.0;
      if (tid < 15) { double t = 0.0; for (int m = 0; m < 15; ++m) t += d.prior_H[tid * 15 + m] * shr[m]; chi = shr[tid] * t; }
      chi = posei_block_sum(chi, sh);
      double r0, rho1;
      dev::huber(chi, d.huber_prior, r0, rho1);
      for (int idx = tid; idx < 225; idx += kIT) {   // W J with W = rho' Omega
        const int k = idx / 15, c = idx - k * 15;
        double t = 0.0;
        for (int m = 0; m < 15; ++m) t += d.prior_H[k * 15 + m] * shJp[m * 15 + c];
        shWJ[idx] = rho1 * t;
      }
      __syncthreads();
      for (int idx = tid; idx < 225; idx += kIT) {
        const int i = idx / 15, jq = idx - i * 15;
        double t = 0.0;
        for (int k = 0; k < 15; ++k) t += shJp[k * 15 + i] * shWJ[k * 15 + jq];
        shH[(15 + i) * n + 15 + jq] += t;
      }
      if (tid < 15) { double t = 0.0; for (int k = 0; k < 15; ++k) t += shWJ[k * 15 + tid] * shr[k]; shb[15 + tid] -= t; }
      __syncthreads();
    }
  };
  // ImuCamPose::Update (src/G2oTypes.cc:187-220) of a pose record in LDS
  auto pose_update = [&](double* P, const double* pu, bool cams) {
    double tw[3], E[9], Rwb[9], Rbw[9], tbw[3], tc[3];
    imu::m3_vec(P + 12, pu + 3, tw);
    for (int i = 0; i < 3; ++i) P[21 + i] += tw[i];
    imu::exp_so3(pu, E);
    imu::m3_mul(P + 12, E, Rwb);
    for (int i = 0; i < 9; ++i) P[12 + i] = Rwb[i];
    if (!cams) return;
    for (int i = 0; i < 3; ++i) for (int jq = 0; jq < 3; ++jq) Rbw[i * 3 + jq] = Rwb[jq * 3 + i];
    imu::m3_vec(Rbw, P + 21, tbw);
    tbw[0] = -tbw[0]; tbw[1] = -tbw[1]; tbw[2] = -tbw[2];
    imu::m3_mul(d.cam.Rcb, Rbw, P);
    imu::m3_vec(d.cam.Rcb, tbw, tc);
    for (int i = 0; i < 3; ++i) P[9 + i] = tc[i] + d.cam.tcb[i];
  };
  auto depth_positive = [&](int kind, const double* R, const double* X) {
    if (kind == OSH_EDGE_RIGHT) {
      double r2[3], t2 = d.cam.trl[2];
#pragma unroll
      for (int c = 0; c < 3; ++c) r2[c] = d.cam.Rrl[6] * R[c] + d.cam.Rrl[7] * R[3 + c] + d.cam.Rrl[8] * R[6 + c];
      t2 += d.cam.Rrl[6] * R[9] + d.cam.Rrl[7] * R[10] + d.cam.Rrl[8] * R[11];
      return (r2[0] * X[0] + r2[1] * X[1] + r2[2] * X[2] + t2) > 0.0;
    }
    return (R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + R[11]) > 0.0;
  };

  bool robust = true;
  int n_bad = 0, n_inl = 0, rounds = 0;
  const int n_graph_edges = d.E + (mode1 ? 4 : 3);
  for (int round = 0; round < 4; ++round) {
    bool ok = true;
    for (int it = 0; it < d.iters[round] && ok; ++it) {
      build(robust);
      if (tid < 64) {
        const bool good = mode1 ? posei_solve_wave<32>(shH, shb, n, shU, shx) : posei_solve_wave<16>(shH, shb, n, shU, shx);   // on failure x keeps the previous values and is still applied
        if (tid == 0) sh_ok = good ? 1 : 0;
      }
      __syncthreads();
      if (tid == 0) {
        pose_update(shP, shx, true);
        for (int i = 0; i < 9; ++i) shs[i] += shx[6 + i];
        if (mode1) {
          pose_update(shpP, shx + 15, false);
          for (int i = 0; i < 9; ++i) shps[i] += shx[21 + i];
        }
      }
      __syncthreads();
      ok = sh_ok != 0;
    }
    // ---- classification (:4747-4818 / :5139-5208)
    double pose[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) pose[k] = shP[k];
    const float chi2close = 1.5 * d.chi2_mono[round];
    int bad = 0, inl = 0;
    for (int e = tid; e < d.E; e += kIT) {
      const int kind = eKind(e);
      double X[3], obs[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { X[k] = eX(e, k); obs[k] = eObs(e, k); }
      if (eOutlier(e)) { VisEval ev; vis_residual(d.cam, kind, pose, X, obs, eInfo(e), ev); setChi(e, ev.chi2); }
      const float chi2 = (float)eChi(e);
      bool o;
      if (kind != OSH_EDGE_STEREO) {
        const bool bClose = eClose(e);
        o = (chi2 > d.chi2_mono[round] && !bClose) || (bClose && chi2 > chi2close) || !depth_positive(kind, pose, X);
      } else o = chi2 > d.chi2_stereo[round];
      setOutlier(e, o); setLevel(e, o);
      if (o) ++bad; else ++inl;
    }
    n_bad = (int)posei_block_sum((double)bad, sh);
    n_inl = (int)posei_block_sum((double)inl, sh);
    rounds = round + 1;
    if (round == 2) robust = false;
    __syncthreads();
    if (n_graph_edges < 10) break;
  }
  double pose[24];
#pragma unroll
  for (int k = 0; k < 24; ++k) pose[k] = shP[k];
  if (n_inl < 30 && !d.rec_init) {   // recovery (:4821-4848)
    int bad = 0;
    for (int e = tid; e < d.E; e += kIT) {
      const int kind = eKind(e);
      double X[3], obs[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { X[k] = eX(e, k); obs[k] = eObs(e, k); }
      VisEval ev;
      vis_residual(d.cam, kind, pose, X, obs, eInfo(e), ev);
      setChi(e, ev.chi2);
      if (ev.chi2 < (kind == OSH_EDGE_STEREO ? 24.f : 18.f)) setOutlier(e, false); else ++bad;
    }
    n_bad = (int)posei_block_sum((double)bad, sh);
  }
  // ---- Hessian of the frame's ConstraintPoseImu (:4858-4893 / :5252-5293): re-linearised at the final estimate, plain information
  {
    const int o2 = mode1 ? 15 : 0;
    if (tid == 0) inertial_jacobian(shrec, shpP, shps, shP, shs, shJ);
    double H[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) H[k] = 0.0;
    for (int e = tid; e < d.E; e += kIT) {
      if (eOutlier(e)) continue;
      const int kind = eKind(e);
      const double info = eInfo(e);
      double X[3], obs[3], JX[9], Jp[18];
#pragma unroll
      for (int k = 0; k < 3; ++k) { X[k] = eX(e, k); obs[k] = eObs(e, k); }
      VisEval ev;
      vis_residual(d.cam, kind, pose, X, obs, info, ev);
      vis_jacobians(d.cam, kind, pose, ev.Xc, JX, Jp);
      int m = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = a; c < 6; ++c) { H[m] += (Jp[a] * info) * Jp[c] + (Jp[6 + a] * info) * Jp[6 + c] + (Jp[12 + a] * info) * Jp[12 + c]; ++m; }
    }
    __syncthreads();
    for (int idx = tid; idx < 9 * 24; idx += kIT) {
      const int k = idx / 24, c = idx - k * 24;
      double t = 0.0;
      for (int m = 0; m < 9; ++m) t += d.info[k * 9 + m] * shJ[m * 24 + c];
      shWJ[idx] = t;
    }
    double red[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) red[k] = H[k];
    posei_block_sum_n<21>(red, shn);
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < 21; ++k) shred[k] = red[k];
    }
    if (mode1 && tid == 0) {
      double T[9], dd[3], rr[15], invJr[9];
      imu::m3_tmul(d.prior_R, shpP + 12, T);
      imu::log_so3(T, rr);
      (void)dd;
      imu::inv_right_jac(rr, invJr);
      for (int i = 0; i < 225; ++i) shJp[i] = 0.0;
      for (int i = 0; i < 3; ++i) for (int jq = 0; jq < 3; ++jq) { shJp[i * 15 + jq] = invJr[i * 3 + jq]; shJp[(3 + i) * 15 + 3 + jq] = T[i * 3 + jq]; }
      for (int i = 6; i < 15; ++i) shJp[i * 15 + i] = 1.0;
    }
    __syncthreads();
    double* WJp = shH;   // 15 x 15 scratch (the solver's matrix is no longer needed)
    if (mode1) {
      for (int idx = tid; idx < 225; idx += kIT) {
        const int k = idx / 15, c = idx - k * 15;
        double t = 0.0;
        for (int m = 0; m < 15; ++m) t += d.prior_H[k * 15 + m] * shJp[m * 15 + c];
        WJp[idx] = t;
      }
      __syncthreads();
    }
    // reference layout: mode 0 [P V G A] of the frame; mode 1 [previous 15 | current 15]
    auto ref_of = [&](int c) { return mode1 ? c : (c >= 15 ? c - 15 : -1); };   // EdgeInertial column -> row / column of H
    for (int idx = tid; idx < n * n; idx += kIT) {
      const int r = idx / n, c = idx - r * n;
      double acc = 0.0;
      if (r >= o2 && r < o2 + 6 && c >= o2 && c < o2 + 6) {
        const int a0 = r - o2, c0 = c - o2;
        const int a = a0 < c0 ? a0 : c0, bq = a0 < c0 ? c0 : a0;
        acc += shred[a * 6 - a * (a - 1) / 2 + (bq - a)];
      }
      for (int ca = 0; ca < 24; ++ca) {
        if (ref_of(ca) != r) continue;
        for (int cb = 0; cb < 24; ++cb) {
          if (ref_of(cb) != c) continue;
          double t = 0.0;
          for (int k = 0; k < 9; ++k) t += shJ[k * 24 + ca] * shWJ[k * 24 + cb];
          acc += t;
        }
      }
      for (int which = 0; which < 2; ++which) {
        const double* Og = which == 0 ? d.info_g : d.info_a;
        const int c2 = o2 + 9 + 3 * which, c1 = mode1 ? 9 + 3 * which : -100;
        const bool rr2 = r >= c2 && r < c2 + 3, cc2 = c >= c2 && c < c2 + 3, rr1 = r >= c1 && r < c1 + 3, cc1 = c >= c1 && c < c1 + 3;
        if (rr2 && cc2) acc += Og[(r - c2) * 3 + (c - c2)];
        if (rr1 && cc1) acc += Og[(r - c1) * 3 + (c - c1)];
        if (rr1 && cc2) acc -= Og[(r - c1) * 3 + (c - c2)];
        if (rr2 && cc1) acc -= Og[(r - c2) * 3 + (c - c1)];
      }
      if (mode1 && r < 15 && c < 15) {
        double t = 0.0;
        for (int k = 0; k < 15; ++k) t += shJp[k * 15 + r] * WJp[k * 15 + c];
        acc += t;
      }
      out.H[idx] = acc;
    }
  }
  if (tid < 24) out.P[tid] = shP[tid];
  if (tid < 9) out.s[tid] = shs[tid];
  if (tid == 0) { out.n_bad = n_bad; out.n_inliers = n_inl; out.rounds = rounds; }
  if (cached) {   // the caller's per-edge results
    __syncthreads();
    for (int e = tid; e < d.E; e += kIT) { v.outlier[e0 + e] = sOutlier[e]; v.chi2[e0 + e] = sChi[e]; v.level[e0 + e] = sLevel[e]; }
  }
}

struct PoseiPinned {
  void* p = nullptr;
  size_t cap = 0;
  ~PoseiPinned() { if (p) (void)hipHostFree(p); }
  void* reserve(size_t bytes) {
    if (bytes <= cap) return p;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&p, want) != hipSuccess) { p = nullptr; return nullptr; }
    cap = want;
    return p;
  }
};
// staging and device arena of osh_posei_optimize: kept with the context (released by osh_lba_destroy)
struct PoseiBuffers { PoseiPinned h_in, h_out; DevBuf arena; };

}  // namespace osh

using namespace osh;

#define OSH_TRY(expr) do { int _rc = (expr); if (_rc != OSH_OK) return _rc; } while (0)

extern "C" int osh_lba_stream(osh_lba_ctx* ctx, int* device, hipStream_t* stream);   // lba_device.hip
extern "C" void** osh_lba_attachment(osh_lba_ctx* ctx, int slot, void (*free_fn)(void*));   // lba_device.hip

extern "C" int osh_posei_optimize(osh_lba_ctx* ctx, int32_t n, const osh_posei_problem* pr, osh_posei_result* res) {
  if (!ctx || n <= 0 || !pr || !res) { set_error("osh_posei_optimize: bad arguments"); return OSH_ERR_INVALID; }
  int device = 0;
  hipStream_t s = nullptr;
  OSH_TRY(osh_lba_stream(ctx, &device, &s));
  OSH_HIP(hipSetDevice(device));
  std::vector<PoseiDesc> h_desc(n);
  size_t NE = 0;
  for (int f = 0; f < n; ++f) {
    const osh_posei_problem& p = pr[f];
    if (p.n_edges < 0 || (p.mode != 0 && p.mode != 1) || !p.Rcw || !p.tcw || !p.Rwb || !p.twb || !p.vel || !p.bias_g || !p.bias_a || !p.prev_Rwb ||
        !p.prev_twb || !p.prev_vel || !p.prev_bias_g || !p.prev_bias_a || !p.Rcb || !p.tcb || !p.tbc || !p.cam || !p.preint || !p.info_inertial ||
        !p.info_g || !p.info_a || (p.n_edges > 0 && (!p.points || !p.edge_kind || !p.edge_obs || !p.edge_info))) {
      set_error("frame %d: bad mode, negative size or NULL array", f); return OSH_ERR_INVALID;
    }
    if (p.mode == 1 && (!p.prior_Rwb || !p.prior_twb || !p.prior_vel || !p.prior_bg || !p.prior_ba || !p.prior_H)) {
      set_error("frame %d: PoseInertialOptimizationLastFrame needs the previous frame's ConstraintPoseImu", f); return OSH_ERR_INVALID;
    }
    PoseiDesc& d = h_desc[f];
    std::memset(&d, 0, sizeof(d));
    d.mode = p.mode; d.E = p.n_edges; d.edge_off = (int)NE; d.rec_init = p.rec_init;
    LibaDesc& c = d.cam;
    std::memcpy(c.Rcb, p.Rcb, 72); std::memcpy(c.tcb, p.tcb, 24); std::memcpy(c.tbc, p.tbc, 24); std::memcpy(c.cam, p.cam, 40);
    c.kb8_on = p.kb8 ? 1 : 0;
    for (int k = 0; k < 4; ++k) c.kb8[k] = p.kb8 ? p.kb8[k] : 0.0;
    c.rig_on = (p.kb8 && p.cam2 && p.trl) ? 1 : 0;
    if (c.rig_on) {
      double tcb1[3];
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) c.Rrl[i * 3 + j] = p.trl[i * 4 + j]; c.trl[i] = p.trl[i * 4 + 3]; }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { double a = 0.0; for (int k = 0; k < 3; ++k) a += c.Rrl[i * 3 + k] * c.Rcb[k * 3 + j]; c.Rcb1[i * 3 + j] = a; }
      for (int i = 0; i < 3; ++i) tcb1[i] = c.Rrl[i * 3] * c.tcb[0] + c.Rrl[i * 3 + 1] * c.tcb[1] + c.Rrl[i * 3 + 2] * c.tcb[2] + c.trl[i];
      for (int i = 0; i < 3; ++i) c.tbc1[i] = -(c.Rcb1[i] * tcb1[0] + c.Rcb1[3 + i] * tcb1[1] + c.Rcb1[6 + i] * tcb1[2]);
      std::memcpy(c.cam2, p.cam2, 64);
    }
    for (int e = 0; e < p.n_edges; ++e) {
      if (p.edge_kind[e] > OSH_EDGE_RIGHT) { set_error("frame %d edge %d: kind out of range", f, e); return OSH_ERR_INVALID; }
      if (p.edge_kind[e] == OSH_EDGE_RIGHT && !c.rig_on) { set_error("frame %d edge %d: a right-camera edge needs kb8, cam2 and trl", f, e); return OSH_ERR_INVALID; }
      if (p.edge_kind[e] == OSH_EDGE_STEREO && p.kb8) { set_error("frame %d: a KannalaBrandt8 frame takes monocular edges only (edge %d)", f, e); return OSH_ERR_UNSUPPORTED; }
    }
    std::memcpy(d.P, p.Rcw, 72); std::memcpy(d.P + 9, p.tcw, 24); std::memcpy(d.P + 12, p.Rwb, 72); std::memcpy(d.P + 21, p.twb, 24);
    std::memcpy(d.s, p.vel, 24); std::memcpy(d.s + 3, p.bias_g, 24); std::memcpy(d.s + 6, p.bias_a, 24);
    d.pP[0] = d.pP[4] = d.pP[8] = 1.0;
    std::memcpy(d.pP + 12, p.prev_Rwb, 72); std::memcpy(d.pP + 21, p.prev_twb, 24);
    std::memcpy(d.ps, p.prev_vel, 24); std::memcpy(d.ps + 3, p.prev_bias_g, 24); std::memcpy(d.ps + 6, p.prev_bias_a, 24);
    std::memcpy(d.rec, p.preint, OSH_PREINT_FLOATS * 4);
    std::memcpy(d.info, p.info_inertial, 81 * 8); std::memcpy(d.info_g, p.info_g, 72); std::memcpy(d.info_a, p.info_a, 72);
    if (p.mode == 1) {
      std::memcpy(d.prior_R, p.prior_Rwb, 72); std::memcpy(d.prior_t, p.prior_twb, 24);
      std::memcpy(d.prior_s, p.prior_vel, 24); std::memcpy(d.prior_s + 3, p.prior_bg, 24); std::memcpy(d.prior_s + 6, p.prior_ba, 24);
      std::memcpy(d.prior_H, p.prior_H, 225 * 8);
    }
    d.huber_mono = p.huber_mono; d.huber_stereo = p.huber_stereo; d.huber_prior = p.huber_prior;
    for (int k = 0; k < 4; ++k) { d.chi2_mono[k] = p.chi2_mono[k]; d.chi2_stereo[k] = p.chi2_stereo[k]; d.iters[k] = p.iterations[k]; }
    NE += (size_t)p.n_edges;
  }
  if (NE > 0x7fffff00u) { set_error("batch too large for 32-bit offsets"); return OSH_ERR_UNSUPPORTED; }
  // one pinned staging buffer, one device arena, one copy each way (the call of a single frame was a dozen copies)
  void** slot = osh_lba_attachment(ctx, 2, [](void* q) { delete static_cast<PoseiBuffers*>(q); });
  if (!slot) { set_error("osh_posei_optimize: no context"); return OSH_ERR_INVALID; }
  if (!*slot) *slot = new PoseiBuffers();
  PoseiBuffers& B = *static_cast<PoseiBuffers*>(*slot);
  size_t in_bytes = 0, out_bytes = 0;
  auto take = [](size_t& total, size_t bytes) { const size_t o = total; total = (total + std::max<size_t>(bytes, 8) + 255) & ~(size_t)255; return o; };
  const size_t i_desc = take(in_bytes, n * sizeof(PoseiDesc)), i_X = take(in_bytes, NE * 24), i_obs = take(in_bytes, NE * 24), i_info = take(in_bytes, NE * 8),
               i_kind = take(in_bytes, NE), i_close = take(in_bytes, NE);
  const size_t o_out = take(out_bytes, n * sizeof(PoseiOut)), o_outlier = take(out_bytes, NE), o_chi2 = take(out_bytes, NE * 8), o_level = take(out_bytes, NE);
  char* hs = static_cast<char*>(B.h_in.reserve(in_bytes));
  char* hr = static_cast<char*>(B.h_out.reserve(out_bytes));
  if (!hs || !hr) { set_error("osh_posei_optimize: pinned staging allocation failed"); return OSH_ERR_DEVICE; }
  std::memcpy(hs + i_desc, h_desc.data(), n * sizeof(PoseiDesc));
  for (int f = 0; f < n; ++f) {
    const osh_posei_problem& p = pr[f];
    const size_t o = (size_t)h_desc[f].edge_off;
    if (p.n_edges > 0) {
      std::memcpy(hs + i_X + o * 24, p.points, (size_t)p.n_edges * 24);
      std::memcpy(hs + i_obs + o * 24, p.edge_obs, (size_t)p.n_edges * 24);
      std::memcpy(hs + i_info + o * 8, p.edge_info, (size_t)p.n_edges * 8);
      std::memcpy(hs + i_kind + o, p.edge_kind, (size_t)p.n_edges);
      if (p.edge_close) std::memcpy(hs + i_close + o, p.edge_close, (size_t)p.n_edges); else std::memset(hs + i_close + o, 0, (size_t)p.n_edges);
    }
  }
  OSH_TRY(B.arena.reserve(in_bytes + out_bytes));
  char* din = B.arena.as<char>();
  char* dout = din + in_bytes;
  OSH_HIP(hipMemcpyAsync(din, hs, in_bytes, hipMemcpyHostToDevice, s));
  PoseiView v;
  v.desc = reinterpret_cast<const PoseiDesc*>(din + i_desc); v.out = reinterpret_cast<PoseiOut*>(dout + o_out);
  v.X = reinterpret_cast<const double*>(din + i_X); v.kind = reinterpret_cast<const unsigned char*>(din + i_kind);
  v.obs = reinterpret_cast<const double*>(din + i_obs); v.info = reinterpret_cast<const double*>(din + i_info);
  v.close = reinterpret_cast<const unsigned char*>(din + i_close); v.chi2 = reinterpret_cast<double*>(dout + o_chi2);
  v.level = reinterpret_cast<unsigned char*>(dout + o_level); v.outlier = reinterpret_cast<unsigned char*>(dout + o_outlier);
  // dynamic LDS for the edges of a frame: 8 doubles + 4 bytes each; frames of more than kPoseiMaxCached edges read global memory instead
  int e_max = 0;
  for (int f = 0; f < n; ++f) e_max = std::max(e_max, h_desc[f].E);
  constexpr int kPoseiMaxCached = 1400;
  v.ecap = (e_max > 0 && e_max <= kPoseiMaxCached) ? ((e_max + 63) & ~63) : 0;
  const size_t dyn_bytes = (size_t)v.ecap * (8 * 8 + 4) + 16;
  {
    static std::mutex attr_mu;
    static std::vector<int> attr_devices;
    std::lock_guard<std::mutex> attr_lock(attr_mu);
    if (std::find(attr_devices.begin(), attr_devices.end(), device) == attr_devices.end()) {
      OSH_HIP(hipFuncSetAttribute((const void*)k_posei, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      attr_devices.push_back(device);
    }
  }
  hipLaunchKernelGGL(k_posei, dim3((unsigned)n), dim3(kIT), dyn_bytes, s, v);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { set_error("k_posei launch failed: %s", hipGetErrorString(le)); return OSH_ERR_DEVICE; }
  OSH_HIP(hipMemcpyAsync(hr, dout, out_bytes, hipMemcpyDeviceToHost, s));
  OSH_HIP(hipStreamSynchronize(s));
  const PoseiOut* h_out = reinterpret_cast<const PoseiOut*>(hr + o_out);
  for (int f = 0; f < n; ++f) {
    const PoseiDesc& d = h_desc[f];
    if (res[f].outlier && d.E) std::memcpy(res[f].outlier, hr + o_outlier + d.edge_off, (size_t)d.E);
    if (res[f].edge_chi2 && d.E) std::memcpy(res[f].edge_chi2, hr + o_chi2 + (size_t)d.edge_off * 8, (size_t)d.E * 8);
  }
  for (int f = 0; f < n; ++f) {
    const PoseiOut& o = h_out[f];
    osh_posei_result& r = res[f];
    std::memcpy(r.Rcw, o.P, 72); std::memcpy(r.tcw, o.P + 9, 24); std::memcpy(r.Rwb, o.P + 12, 72); std::memcpy(r.twb, o.P + 21, 24);
    std::memcpy(r.vel, o.s, 24); std::memcpy(r.bias_g, o.s + 3, 24); std::memcpy(r.bias_a, o.s + 6, 24);
    r.n_bad = o.n_bad; r.n_inliers = o.n_inliers; r.rounds = o.rounds; r.status = OSH_OK;
    std::memcpy(r.H, o.H, sizeof(r.H));
  }
  return OSH_OK;
}
