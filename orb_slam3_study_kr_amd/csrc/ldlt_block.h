// ldlt_block.h -- dense blocked LDL^T (no pivoting, upper storage, fails only on an exactly-zero pivot) + solve,
// executed cooperatively by one thread block.  Stand-in for Eigen::SimplicialLDLT behind LinearSolverEigen::solve
// (Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:94-124).
//   A   : n x n row-major in global memory, upper triangle valid; overwritten by the factor
//   rhs : n, overwritten by the forward-substituted right-hand side
//   sh  : dynamic LDS scratch, ldlt_lds_doubles(NB, W, NT) doubles, W >= n + 8
//   xs  : returns a pointer into `sh` holding the solution (n doubles) when the result is true
#pragma once
#include <hip/hip_runtime.h>
#include "lba_math.h"

namespace osh {

__host__ __device__ constexpr size_t ldlt_lds_doubles(int nb, int W, int nthreads) {
  return (size_t)2 * nb * W + W + 2 * nb + nthreads / 64 + 8;
}

template <int NB, int NT>
__device__ bool ldlt_solve_block(double* __restrict__ A, double* __restrict__ rhs, const int n, const int W, double* sh,
                                 double*& xs_out, double*& shw_out) {
  constexpr int nb = NB;
  constexpr int kSolveThreads = NT;
  const int tid = threadIdx.x;
  double* U = sh;                    // [nb][W]  unscaled panel rows  (d_k * l_jk)
  double* Lp = sh + (size_t)nb * W;  // [nb][W]  scaled panel rows    (l_jk)
  double* xs = Lp + (size_t)nb * W;  // [W]
  double* dd = xs + W;               // [nb]
  double* part = dd + nb;            // [nb]
  double* shw = part + nb;           // [NT/64] cross-wave scratch for the caller
  int& sh_ok = *reinterpret_cast<int*>(shw + kSolveThreads / 64);
  xs_out = xs;
  shw_out = shw;
  if (tid == 0) sh_ok = 1;
  __syncthreads();

  for (int k0 = 0; k0 < n; k0 += nb) {
    const int kb = min(nb, n - k0);
    const int m = n - k0;  // local columns 0..m-1, rhs at local column m
    // ---- 1. diagonal block: load, then factor it with ONE wavefront (LDS ops of a wave stay in order)
    for (int idx = tid; idx < kb * kb; idx += kSolveThreads) {
      const int r = idx / kb, c2 = idx - r * kb;
      U[r * W + c2] = (c2 >= r) ? A[(size_t)(k0 + r) * n + k0 + c2] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
      for (int k = 0; k < kb; ++k) {
        const double d = U[k * W + k];
        if (d == 0.0 && tid == 0) sh_ok = 0;
        const int rows = kb - k - 1;
        for (int idx = tid; idx < rows * kb; idx += 64) {
          const int ii = k + 1 + idx / kb, jj = idx - (idx / kb) * kb;
          if (jj >= ii) U[ii * W + jj] -= (U[k * W + ii] / d) * U[k * W + jj];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      for (int idx = tid; idx < kb * kb; idx += 64) {
        const int r = idx / kb, c2 = idx - r * kb;
        const double d = U[r * W + r];
        Lp[r * W + c2] = (c2 > r) ? U[r * W + c2] / d : 0.0;
        if (c2 == r) dd[r] = d;
      }
    }
    __syncthreads();
    if (!sh_ok) break;
    // ---- 2. row panel: every thread forward-substitutes whole columns (incl. the rhs column m) in registers
    for (int jj = kb + tid; jj <= m; jj += kSolveThreads) {
      double wv[NB];
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        wv[r] = 0.0;
        if (r < kb) wv[r] = (jj == m) ? rhs[k0 + r] : A[(size_t)(k0 + r) * n + k0 + jj];
      }
#pragma unroll
      for (int r = 1; r < NB; ++r) {
        if (r < kb) {
          double acc = wv[r];
#pragma unroll
          for (int k = 0; k < r; ++k) acc -= Lp[k * W + r] * wv[k];
          wv[r] = acc;
        }
      }
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        if (r < kb) { U[r * W + jj] = wv[r]; Lp[r * W + jj] = (jj < m) ? wv[r] / dd[r] : 0.0; }
      }
    }
    // zero padding so the 4-wide tiles below may over-read
    for (int idx = tid; idx < kb * 4; idx += kSolveThreads) {
      const int r = idx >> 2, jj = m + 1 + (idx & 3);
      U[r * W + jj] = 0.0; Lp[r * W + jj] = 0.0;
    }
    __syncthreads();
    // ---- 3. trailing update of rows k0+kb .. n-1 (upper part) and of the rhs column: 4x4 register tiles
    const int tr = m - kb;  // trailing rows
    if (tr > 0) {
      const int Tr = (tr + 3) >> 2, Tc = (tr + 1 + 3) >> 2;  // column tiles include the rhs column
      const int ntile = Tr * Tc - Tr * (Tr - 1) / 2;
      for (int t = tid; t < ntile; t += kSolveThreads) {
        const float bq = (float)(2 * Tc + 1);
        int ti = (int)((bq - sqrtf(fmaxf(bq * bq - 8.0f * (float)t, 0.0f))) * 0.5f);
        ti = max(0, min(ti, Tr - 1));
        while (ti > 0 && ti * Tc - ti * (ti - 1) / 2 > t) --ti;
        while (ti + 1 < Tr && (ti + 1) * Tc - (ti + 1) * ti / 2 <= t) ++ti;
        const int tj = ti + (t - (ti * Tc - ti * (ti - 1) / 2));
        const int i0 = kb + 4 * ti, j0 = kb + 4 * tj;
        // old values first: their latency hides under the FMAs
        double old[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c2 = 0; c2 < 4; ++c2) {
            const int ii = i0 + r, jj = j0 + c2;
            old[r][c2] = 0.0;
            if (ii < m && jj >= ii && jj <= m) old[r][c2] = (jj == m) ? rhs[k0 + ii] : A[(size_t)(k0 + ii) * n + k0 + jj];
          }
        double acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c2 = 0; c2 < 4; ++c2) acc[r][c2] = 0.0;
        for (int k = 0; k < kb; ++k) {
          double a[4], b[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { a[r] = Lp[k * W + i0 + r]; b[r] = U[k * W + j0 + r]; }
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) acc[r][c2] += a[r] * b[c2];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ii = i0 + r;
          if (ii >= m) continue;
#pragma unroll
          for (int c2 = 0; c2 < 4; ++c2) {
            const int jj = j0 + c2;
            if (jj < ii || jj > m) continue;
            if (jj == m) rhs[k0 + ii] = old[r][c2] - acc[r][c2];
            else A[(size_t)(k0 + ii) * n + k0 + jj] = old[r][c2] - acc[r][c2];
          }
        }
      }
    }
    // ---- 4. write the factor back: L rows, pivots on the diagonal, forward-substituted rhs
    for (int idx = tid; idx < kb * (m + 1); idx += kSolveThreads) {
      const int r = idx / (m + 1), jj = idx - r * (m + 1);
      if (jj == m) rhs[k0 + r] = U[r * W + m];
      else if (jj == r) A[(size_t)(k0 + r) * n + k0 + r] = dd[r];
      else if (jj > r) A[(size_t)(k0 + r) * n + k0 + jj] = Lp[r * W + jj];
    }
    __syncthreads();
  }
  const int ok = sh_ok;
  if (ok) {
    // back substitution  L^T x = D^-1 y, panels in reverse
    const int npanel = (n + nb - 1) / nb;
    const int wv = tid >> 6, lane = tid & 63;
    for (int pi = npanel - 1; pi >= 0; --pi) {
      const int k0 = pi * nb;
      const int kb = min(nb, n - k0);
      const int tail0 = k0 + kb;  // x known for indices >= tail0
      // part[r] = sum_{j>=tail0} L[k0+r][j] x[j] : one wavefront per group of rows
      for (int r = wv; r < kb; r += kSolveThreads / 64) {
        double sacc = 0.0;
        const double* row = A + (size_t)(k0 + r) * n;
        for (int j = tail0 + lane; j < n; j += 64) sacc += row[j] * xs[j];
        sacc = dev::wave_sum(sacc);
        if (lane == 0) part[r] = sacc;
      }
      // diagonal block of the factor (L above the diagonal, pivots on it) and the rhs into LDS
      for (int idx = tid; idx < kb * kb; idx += kSolveThreads) {
        const int r = idx / kb, c2 = idx - r * kb;
        U[r * W + c2] = (c2 >= r) ? A[(size_t)(k0 + r) * n + k0 + c2] : 0.0;
      }
      if (tid < kb) dd[tid] = rhs[k0 + tid];
      __syncthreads();
      if (wv == 0) {
        // lane r holds s_r; columns are eliminated right to left
        double sv = 0.0;
        if (lane < kb) sv = dd[lane] / U[lane * W + lane] - part[lane];
        for (int c2 = kb - 1; c2 >= 0; --c2) {
          const double xc = __shfl(sv, c2, 64);
          if (lane < c2) sv -= U[lane * W + c2] * xc;
        }
        if (lane < kb) xs[k0 + lane] = sv;
      }
      __syncthreads();
    }
    
  } else {
    // zero pivot: LinearSolverEigen::solve returns false; the step is rejected by the controller
    for (int k = tid; k < n; k += kSolveThreads) xs[k] = 0.0;
  }
  __syncthreads();
  return ok != 0;
}

}  // namespace osh
