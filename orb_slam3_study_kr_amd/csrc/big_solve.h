// big_solve.h -- LDL^T + solve of ONE reduced camera system that does not fit a block's LDS: the global bundle adjustment
// of a map beyond ~230 keyframes (Optimizer::GlobalBundleAdjustemnt / BundleAdjustment, src/Optimizer.cc:53-392, whose
// LinearSolverEigen::solve factors the whole-map system, Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:94-124).
//
// Same factorisation as ldlt_block.h (no pivoting, upper storage, A = U^T D U with U unit upper triangular; fails only on an
// exactly-zero pivot), right-looking with 32-wide panels, the matrix in global memory and the whole chip on the trailing update:
//   k_big_diag    one wavefront: factors the 32x32 diagonal block in registers (pivot rows broadcast with v_readlane),
//                 forward-substitutes the panel's rhs entries            z1 = U11^-T b1,  w1 = D1^-1 z1
//   k_big_panel   thread per trailing column: V12 = U11^-T A12 (= D1 U12), U12 = D1^-1 V12, rhs  b2 -= U12^T z1
//   k_big_update  64x64 tiles of the trailing block on the FP64 matrix cores:  A22 -= U12^T V12  (upper tiles only)
//   k_big_back    one block: x = U^-1 w, panels in reverse
//   k_big_finish  pose update T <- exp(x) T and the pose part of computeScale (the tail of k_solve)
// A: n x n row-major (upper triangle valid, overwritten by U with D on the diagonal), b: rhs (overwritten by the solution).
// The trailing update moves 16 bytes per 64 flops (K = 32): it is HBM bound, ~n^3 / 12 bytes per factorisation.
#pragma once
#include <hip/hip_runtime.h>
#include "ldlt_block.h"

namespace osh {

constexpr int kBigNB = 32;       // panel width
constexpr int kBigTile = 64;     // tile edge of the trailing update (one block of 4 wavefronts)
constexpr int kBigMaxPoses = 4000;  // dense S: 24000^2 doubles = 4.6 GB

struct BigSolve {
  double* A; double* b;          // system of the window being solved
  double* V;                     // [kBigNB][n] unscaled panel rows D1 U12
  double* z;                     // [n] unscaled forward-substituted rhs
  int* fail;                     // set when a pivot is exactly zero
  const int* active;             // the window's LmState::active
  int n;
};

__global__ __launch_bounds__(64) void k_big_diag(BigSolve g, int k0) {
  if (!*g.active || *g.fail) return;
  const int lane = threadIdx.x, n = g.n;
  const int kb = min(kBigNB, n - k0);
  const int j = lane < kBigNB ? lane : 0;
  // lane j keeps column j of the block; rows / columns beyond kb are the identity
  double col[kBigNB];
#pragma unroll
  for (int r = 0; r < kBigNB; ++r) {
    const bool in = r < kb && j < kb && r <= j;
    const double a = g.A[in ? (size_t)(k0 + r) * n + k0 + j : 0];
    col[r] = in ? a : (r == j ? 1.0 : 0.0);
  }
  double zr = (lane < kb) ? g.b[k0 + lane] : 0.0;
  bool zero_pivot = false;
#pragma unroll
  for (int k = 0; k < kBigNB; ++k) {
    const double d = ldlt_readlane(col[k], k);
    zero_pivot |= (d == 0.0);
    const double lk = col[k] / d;                       // l_kj in lane j (j > k)
    const double zk = ldlt_readlane(zr, k);             // z_k is final once the steps before k have been applied
    if (lane > k) zr -= lk * zk;
#pragma unroll
    for (int i = k + 1; i < kBigNB; ++i) col[i] -= ldlt_readlane(lk, i) * col[k];
  }
  if (zero_pivot) { if (lane == 0) *g.fail = 1; return; }
  // pivots and scaled rows back into A; z (unscaled) and w = z / d
  double dd[kBigNB];
#pragma unroll
  for (int r = 0; r < kBigNB; ++r) dd[r] = ldlt_readlane(col[r], r);
  if (lane < kb) {
#pragma unroll
    for (int r = 0; r < kBigNB; ++r)
      if (r <= lane && r < kb) g.A[(size_t)(k0 + r) * n + k0 + lane] = (r == lane) ? dd[r] : col[r] / dd[r];
    double dl = 1.0;
#pragma unroll
    for (int r = 0; r < kBigNB; ++r) if (r == lane) dl = dd[r];
    g.z[k0 + lane] = zr;
    g.b[k0 + lane] = zr / dl;
  }
}

__global__ __launch_bounds__(256) void k_big_panel(BigSolve g, int k0) {
  if (!*g.active || *g.fail) return;
  __shared__ double U11[kBigNB][kBigNB + 1];
  __shared__ double d1[kBigNB], z1[kBigNB];
  const int n = g.n, tid = threadIdx.x;
  for (int idx = tid; idx < kBigNB * kBigNB; idx += 256) {
    const int r = idx / kBigNB, c = idx - r * kBigNB;
    const double a = g.A[(size_t)(k0 + r) * n + k0 + (c >= r ? c : r)];
    if (c == r) d1[r] = a;
    U11[r][c] = c > r ? a : 0.0;
  }
  if (tid < kBigNB) z1[tid] = g.z[k0 + tid];
  __syncthreads();
  const int j = k0 + kBigNB + blockIdx.x * 256 + tid;
  if (j >= n) return;
  double v[kBigNB];
#pragma unroll
  for (int r = 0; r < kBigNB; ++r) v[r] = g.A[(size_t)(k0 + r) * n + j];
#pragma unroll
  for (int r = 1; r < kBigNB; ++r) {
    double s = v[r];
#pragma unroll
    for (int k = 0; k < r; ++k) s -= U11[k][r] * v[k];
    v[r] = s;
  }
  double acc = 0.0;
#pragma unroll
  for (int r = 0; r < kBigNB; ++r) {
    const double u = v[r] / d1[r];
    g.V[(size_t)r * n + j] = v[r];
    g.A[(size_t)(k0 + r) * n + j] = u;
    acc += u * z1[r];
  }
  g.b[j] -= acc;
}

// A22 -= U12^T V12 on 64x64 tiles; block (ti, tj) with ti <= tj, wavefront wv takes rows 16 wv .. 16 wv + 15 of the tile.
// MFMA operand layout as in ldlt_block.h: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D[row = (lane >> 4) + 4 reg][col = lane & 15].
__global__ __launch_bounds__(256) void k_big_update(BigSolve g, int k0) {
  if (blockIdx.x > blockIdx.y) return;
  if (!*g.active || *g.fail) return;
  typedef double f64x4 __attribute__((ext_vector_type(4)));
  __shared__ double Us[kBigNB][kBigTile + 4], Vs[kBigNB][kBigTile + 4];
  const int n = g.n, tid = threadIdx.x;
  const int t0 = k0 + kBigNB;
  const int i0 = t0 + blockIdx.x * kBigTile, j0 = t0 + blockIdx.y * kBigTile;
  for (int idx = tid; idx < kBigNB * kBigTile; idx += 256) {
    const int r = idx / kBigTile, c = idx - r * kBigTile;
    Us[r][c] = (i0 + c < n) ? g.A[(size_t)(k0 + r) * n + i0 + c] : 0.0;
    Vs[r][c] = (j0 + c < n) ? g.V[(size_t)r * n + j0 + c] : 0.0;
  }
  __syncthreads();
  const int wv = tid >> 6, lane = tid & 63, lrow = lane >> 4, lcol = lane & 15;
  double a[kBigNB / 4];
#pragma unroll
  for (int q = 0; q < kBigNB / 4; ++q) a[q] = Us[4 * q + lrow][16 * wv + lcol];
#pragma unroll
  for (int jt = 0; jt < kBigTile / 16; ++jt) {
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < kBigNB / 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], Vs[4 * q + lrow][16 * jt + lcol], acc, 0, 0, 0);
    const int col = j0 + 16 * jt + lcol;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = i0 + 16 * wv + lrow + 4 * reg;
      if (row < n && col < n && col >= row) g.A[(size_t)row * n + col] -= acc[reg];
    }
  }
}

__global__ __launch_bounds__(1024) void k_big_back(BigSolve g) {
  if (!*g.active || *g.fail) return;
  __shared__ double U11[kBigNB][kBigNB + 1];
  __shared__ double srow[kBigNB];
  const int n = g.n, tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int npanel = (n + kBigNB - 1) / kBigNB;
  for (int p = npanel - 1; p >= 0; --p) {
    const int k0 = p * kBigNB, kb = min(kBigNB, n - k0), t0 = k0 + kb;
    for (int idx = tid; idx < kBigNB * kBigNB; idx += 1024) {
      const int r = idx / kBigNB, c = idx - r * kBigNB;
      U11[r][c] = (r < kb && c < kb && c > r) ? g.A[(size_t)(k0 + r) * n + k0 + c] : 0.0;
    }
    // s_r = sum_{j >= t0} u_rj x_j : two rows per wavefront, lanes stride the row
    for (int r = wv; r < kBigNB; r += 16) {
      double s = 0.0;
      if (r < kb) {
        // eight loads of the row and of x in flight per lane (one pair per iteration left every L2 round trip exposed: 45 k cycles
        // per panel); the partial sums are added in a fixed order
        const double* row = g.A + (size_t)(k0 + r) * n;
        double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        int jj = t0 + lane;
        for (; jj + 7 * 64 < n; jj += 8 * 64) {
          double a[8], x[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { a[u] = row[jj + 64 * u]; x[u] = g.b[jj + 64 * u]; }
#pragma unroll
          for (int u = 0; u < 8; ++u) p[u] += a[u] * x[u];
        }
        for (; jj < n; jj += 64) p[0] += row[jj] * g.b[jj];
        s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
      }
      s = dev::wave_sum(s);
      if (lane == 0) srow[r] = s;
    }
    __syncthreads();
    if (wv == 0) {
      // U11 x1 = w1 - s, columns right to left: lane r holds its running entry, the solved one is broadcast
      double t = (lane < kb) ? g.b[k0 + lane] - srow[lane] : 0.0;
#pragma unroll
      for (int c = kBigNB - 1; c > 0; --c) {
        const double xc = ldlt_readlane(t, c);
        if (lane < c) t -= U11[lane < kBigNB ? lane : 0][c] * xc;
      }
      if (lane < kb) g.b[k0 + lane] = t;
    }
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace osh
