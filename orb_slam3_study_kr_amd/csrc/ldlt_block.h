// ldlt_block.h -- dense blocked LDL^T (no pivoting, upper storage, fails only on an exactly-zero pivot) + solve,
// executed cooperatively by one thread block.  Stand-in for Eigen::SimplicialLDLT behind LinearSolverEigen::solve
// (Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:94-124).
//   A   : n x n row-major in global memory, upper triangle valid; overwritten by the factor
//   rhs : n, overwritten by the forward-substituted right-hand side
//   sh  : dynamic LDS scratch, ldlt_lds_doubles(NB, W, NT) doubles, W = ldlt_row_stride(n_max)
//   xs  : returns a pointer into `sh` holding the solution (n doubles) when the result is true
#pragma once
#include <hip/hip_runtime.h>
#include "lba_math.h"

namespace osh {

// row stride of the LDS panels for systems of up to n unknowns: n + rhs column + 16 columns of zero padding, even
__host__ __device__ constexpr int ldlt_row_stride(int n) { return (n + 24 + 1) & ~1; }

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

#ifdef OSH_LDLT_TRACE
  long long tr_t[6] = {0, 0, 0, 0, 0, 0};
  long long tr_last = clock64();
#define OSH_TR(i) do { const long long _n = clock64(); tr_t[i] += _n - tr_last; tr_last = _n; } while (0)
#else
#define OSH_TR(i) do {} while (0)
#endif
  for (int k0 = 0; k0 < n; k0 += nb) {
    const int kb = min(nb, n - k0);
    const int m = n - k0;  // local columns 0..m-1, rhs at local column m
    // ---- 1. diagonal block: load, then factor it with ONE wavefront (LDS ops of a wave stay in order)
    for (int idx = tid; idx < kb * kb; idx += kSolveThreads) {
      const int r = idx / kb, c2 = idx - r * kb;
      U[r * W + c2] = (c2 >= r) ? A[(size_t)(k0 + r) * n + k0 + c2] : 0.0;
    }
    __syncthreads();
    OSH_TR(0);
    if (tid < 64) {
      for (int k = 0; k < kb; ++k) {
        const double d = U[k * W + k];
        if (d == 0.0 && tid == 0) sh_ok = 0;
        const int rows = kb - k - 1;
        // multipliers l_ii = u_k,ii / d once per row (the products are the same numbers as with a division per element), then
        // two rows x 32 columns per pass: no FP64 division and no integer division inside the element loop
        if (tid < rows) part[tid] = U[k * W + k + 1 + tid] / d;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (kb <= 32) {
          const int jj = tid & 31, rsub = tid >> 5;
          const double ukj = (jj < kb) ? U[k * W + jj] : 0.0;
          for (int ri = rsub; ri < rows; ri += 2) {
            const int ii = k + 1 + ri;
            if (jj < kb && jj >= ii) U[ii * W + jj] -= part[ri] * ukj;
          }
        } else {
          for (int idx = tid; idx < rows * kb; idx += 64) {
            const int ri = idx / kb, jj = idx - ri * kb;
            const int ii = k + 1 + ri;
            if (jj >= ii) U[ii * W + jj] -= part[ri] * U[k * W + jj];
          }
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
    OSH_TR(1);
    if (!sh_ok) break;
    // ---- 2. row panel: every thread forward-substitutes whole columns (incl. the rhs column m) in registers
    for (int jj = kb + tid; jj <= m; jj += kSolveThreads) {
      double wv[NB];
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        wv[r] = 0.0;
        if (r < kb) wv[r] = (jj == m) ? rhs[k0 + r] : A[(size_t)(k0 + r) * n + k0 + jj];
      }
      // the multipliers of row r are fetched from LDS as one batch BEFORE the dependent chain of FMAs (one wait per row
      // instead of one per multiplier: the chain itself cannot hide an LDS round trip)
#pragma unroll
      for (int r = 1; r < NB; ++r) {
        if (r < kb) {
          double lrow[NB];
#pragma unroll
          for (int k = 0; k < r; ++k) lrow[k] = Lp[k * W + r];
          double acc = wv[r];
#pragma unroll
          for (int k = 0; k < r; ++k) acc -= lrow[k] * wv[k];
          wv[r] = acc;
        }
      }
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        if (r < kb) { U[r * W + jj] = wv[r]; Lp[r * W + jj] = (jj < m) ? wv[r] / dd[r] : 0.0; }
      }
    }
    // zero padding so the 16-wide tiles below may over-read (W >= n + 17, see ldlt_row_stride)
    for (int idx = tid; idx < kb * 16; idx += kSolveThreads) {
      const int r = idx >> 4, jj = m + 1 + (idx & 15);
      U[r * W + jj] = 0.0; Lp[r * W + jj] = 0.0;
    }
    __syncthreads();
    OSH_TR(2);
    // ---- 3. trailing update of rows k0+kb .. n-1 (upper part) and of the rhs column on the FP64 matrix cores: one
    // wavefront per 16x16 tile of the trailing block, C -= L[16 x kb] U[kb x 16] as kb/4 v_mfma_f64_16x16x4_f64.
    // Operand lanes read straight from the panels in LDS: A[i = lane & 15][k = lane >> 4] = Lp[k][i0 + i], B[k][j] = U[k][j0 + j];
    // lane holds D[row = (lane >> 4) + 4 reg][col = lane & 15].  (A 4x4 register-tiled VALU version spent 3 of 4 issue
    // slots on operand traffic: 44 k cycles per panel on one CU against ~10 k here.)
    const int tr = m - kb;  // trailing rows
    if (tr > 0) {
      typedef double ldlt_f64x4 __attribute__((ext_vector_type(4)));
      const int Tr = (tr + 15) >> 4, Tc = (tr + 1 + 15) >> 4;  // column tiles include the rhs column (local column m)
      const int ntile = Tr * Tc - Tr * (Tr - 1) / 2;
      const int wave = tid >> 6, lane = tid & 63;
      const int lrow = lane >> 4, lcol = lane & 15;
      // tile t -> (ti, tj), its 4 old values per lane (unconditional loads: an out-of-range entry reads A[0] and is discarded);
      // the loads of the wave's NEXT tile are issued before the current tile is multiplied (an L2 round trip is longer than
      // the six MFMAs of a tile)
      struct Tile { int i0, jj; double old[4]; };
      auto prep = [&](int t, Tile& T) {
        int ti = 0, rem = t;
        while (rem >= Tc - ti) { rem -= Tc - ti; ++ti; }
        T.i0 = kb + 16 * ti;
        T.jj = kb + 16 * (ti + rem) + lcol;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int ii = T.i0 + lrow + 4 * reg;
          const bool in = ii < m && T.jj >= ii && T.jj <= m;
          const double* src = !in ? A : ((T.jj == m) ? rhs + k0 + ii : A + (size_t)(k0 + ii) * n + k0 + T.jj);
          T.old[reg] = *src;
        }
      };
      // two tiles per step: their MFMA chains are independent, so the matrix core issues back to back instead of waiting
      // for each accumulator; all operand reads of a step are issued before its first MFMA
      auto finish = [&](const Tile& T, const ldlt_f64x4& acc) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int ii = T.i0 + lrow + 4 * reg;
          if (!(ii < m && T.jj >= ii && T.jj <= m)) continue;
          if (T.jj == m) rhs[k0 + ii] = T.old[reg] - acc[reg];
          else A[(size_t)(k0 + ii) * n + k0 + T.jj] = T.old[reg] - acc[reg];
        }
      };
      auto run2 = [&](const Tile& T0, const Tile& T1, bool two) {
        constexpr int KS = (NB + 3) / 4;
        double a0[KS], b0[KS], a1[KS], b1[KS];
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          const int k = 4 * q + lrow;
          const bool kin = k < kb;
          const int kc = kin ? k : 0;
          a0[q] = Lp[kc * W + T0.i0 + lcol]; b0[q] = U[kc * W + T0.jj];
          a1[q] = Lp[kc * W + T1.i0 + lcol]; b1[q] = U[kc * W + T1.jj];
          if (!kin) { a0[q] = 0.0; b0[q] = 0.0; a1[q] = 0.0; b1[q] = 0.0; }
        }
        ldlt_f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          if (4 * q < kb) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], acc1, 0, 0, 0);
          }
        }
        finish(T0, acc0);
        if (two) finish(T1, acc1);
      };
      constexpr int nwaves = kSolveThreads / 64;
      // wave w takes tiles w, w + nwaves, ... two at a time; the loads of the next pair are in flight during the current one
      Tile TA0, TA1, TB0, TB1;
      auto prep_pair = [&](int t, Tile& X, Tile& Y) {
        prep(t, X);
        if (t + nwaves < ntile) prep(t + nwaves, Y); else Y = X;
      };
      if (wave < ntile) prep_pair(wave, TA0, TA1);
      for (int t = wave; t < ntile; t += 4 * nwaves) {
        const bool moreB = t + 2 * nwaves < ntile;
        if (moreB) prep_pair(t + 2 * nwaves, TB0, TB1);
        run2(TA0, TA1, t + nwaves < ntile);
        if (moreB) {
          if (t + 4 * nwaves < ntile) prep_pair(t + 4 * nwaves, TA0, TA1);
          run2(TB0, TB1, t + 3 * nwaves < ntile);
        }
      }
    }
    OSH_TR(3);
    // ---- 4. write the factor back: L rows, pivots on the diagonal, forward-substituted rhs
    for (int idx = tid; idx < kb * (m + 1); idx += kSolveThreads) {
      const int r = idx / (m + 1), jj = idx - r * (m + 1);
      if (jj == m) rhs[k0 + r] = U[r * W + m];
      else if (jj == r) A[(size_t)(k0 + r) * n + k0 + r] = dd[r];
      else if (jj > r) A[(size_t)(k0 + r) * n + k0 + jj] = Lp[r * W + jj];
    }
    __syncthreads();
    OSH_TR(4);
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
#ifdef OSH_LDLT_TRACE
  OSH_TR(5);
  if (tid == 0 && blockIdx.x == 0) printf("ldlt n=%d: load_diag %lld  factor %lld  row_panel %lld  trailing %lld  writeback %lld  backsub %lld cycles\n", n, tr_t[0], tr_t[1], tr_t[2], tr_t[3], tr_t[4], tr_t[5]);
#endif
  return ok != 0;
}

}  // namespace osh
