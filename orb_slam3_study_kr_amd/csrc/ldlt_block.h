// ldlt_block.h -- dense blocked LDL^T (no pivoting, upper storage, fails only on an exactly-zero pivot) + solve,
// executed cooperatively by one thread block.  Stand-in for Eigen::SimplicialLDLT behind LinearSolverEigen::solve
// (Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:94-124).
//   A   : n x n row-major in global memory, upper triangle valid; overwritten by the factor (the strict lower triangle is
//         scratch: never read, partly overwritten)
//   rhs : n, read once (the forward substitution runs on a copy in LDS)
//   sh  : dynamic LDS scratch, ldlt_lds_doubles(NB, W, NT) doubles, W = ldlt_row_stride(n_max)
//   xs  : returns a pointer into `sh` holding the solution (n doubles) when the result is true
//   lo_pose (optional): the sparsity the caller knows.  The unknowns come in blocks of six (one keyframe); lo_pose[p] is the first
//         block row with a non-zero in block column p of the upper triangle (<= p).  No pivoting => the factor keeps this column
//         envelope, so a panel only has to touch the columns whose envelope reaches up to it: the row panel and the trailing update
//         run over the 16-wide tile columns that hold such a column and skip the others, whose entries are exact zeros and stay
//         zeros (the skipped operations are 0 -= l * 0).  nullptr: dense.
#pragma once
#include <hip/hip_runtime.h>
#include "lba_math.h"

namespace osh {

// row stride of the LDS panels for systems of up to n unknowns: n + rhs column + 16 columns of zero padding, even
__host__ __device__ constexpr int ldlt_row_stride(int n) { return (n + 24 + 1) & ~1; }

__host__ __device__ constexpr size_t ldlt_lds_doubles(int nb, int W, int nthreads) {
  // two nb x nb blocks (factor rows, inverse of L); then ints: live tile columns of a panel, envelope of the block columns, last live column per panel
  return (size_t)nb * W + W + 3 * nb + 2 * (size_t)nb * nb + nthreads / 64 + 8 + (W / 32 + 4) + (W / 12 + 2) + (W / (2 * nb) + 2);
}

// value of `v` in lane `lane` (compile-time constant after unrolling) as a wave-uniform scalar
__device__ __forceinline__ double ldlt_readlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Block barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL access of the wavefront
// (s_waitcnt vmcnt(0)), i.e. for the stores of the phase just finished and for loads requested ahead of time.  Used between the
// phases of a panel whose hand-over is through LDS; the phase that writes the trailing matrix ends with a full __syncthreads().
__device__ __forceinline__ void ldlt_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ONE wavefront factors an NB x NB diagonal block (threads 0..63 of the block call this).  Lane j keeps column j in registers:
// l_k,j = u_k,j / d_k is formed by lane j itself, then u_i,j -= l_k,i * u_k,j for every row i > k with the multiplier l_k,i broadcast
// (below).  Lanes j < i compute values of the unused lower triangle; nothing valid reads them.  Same operations in the same order as
// the textbook loop; a padded pivot is 1 with zero multipliers.  Lanes NB .. 2 NB - 1 carry the columns of the identity through the
// same row operations (they execute the instruction stream anyway): what they hold at the end is L^-1, the operator of the row panel.
//   Ld  : in, the block padded with the identity to NB x NB (upper triangle); out, the scaled rows l_kj (k < j), zeros left of the diagonal
//   Mi  : out, L^-1       dd, ddi : out, pivots and their reciprocals (NB each)       part : NB doubles of scratch
template <int NB>
__device__ __forceinline__ void ldlt_factor_diag(double* Ld, double* Mi, double* dd, double* ddi, double* part, const int tid, int& sh_ok) {
  static_assert(2 * NB <= 64, "the identity columns ride in the upper lanes of the wavefront");
  double col[NB];
  const int cj = tid < NB ? tid : 0;
  const bool idl = tid >= NB && tid < 2 * NB;
#pragma unroll
  for (int r = 0; r < NB; ++r) { const double v = Ld[r * NB + cj]; col[r] = idl ? (r == tid - NB ? 1.0 : 0.0) : v; }
  double* lout = tid < NB ? Ld + tid : part;  // lanes beyond the block write to a slot nobody reads in this phase
  bool zero_pivot = false;
  // Per pivot k: l_k,j = u_k,j / d_k by lane j, stored as row k of the block's factor.  Only the NEXT row takes its multiplier through
  // v_readlane (its diagonal entry is the next pivot, whose reciprocal -- hardware seed + two Newton steps -- is started at once);
  // the other rows take theirs from the row just stored, as broadcast LDS reads (ds_read_b128: two multipliers per instruction)
  // requested now and used one step later, between the start of the next reciprocal and its first use.  A single wavefront
  // issues one FP64 instruction per ~10 cycles whatever it depends on (profiles/ubench/pivot_chain.hip: 402 cycles per pivot with two
  // v_readlane_b32 + v_fma_f64 per row, 297 this way, same bits), so the step is as long as its instruction count.
  double d = ldlt_readlane(col[0], 0);
  double rd = dev::rcp_nr(d);
  double mprev[NB];     // multipliers of the previous pivot for rows k + 1 .. NB - 1
  double uprev = 0.0;   // the previous pivot's row entry of this lane's column
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    zero_pivot |= d == 0.0;
    const double lk = col[k] * rd;
    const double d_cur = d, rd_cur = rd;
    lout[k * NB] = tid > k ? lk : 0.0;  // scaled row k of the block (l_kj)
    if (k + 1 < NB) {
      if (k > 0) col[k + 1] -= mprev[k + 1] * uprev;   // the older update first, as in the textbook loop
      col[k + 1] -= ldlt_readlane(lk, k + 1) * col[k];
      asm volatile("" : "+v"(col[k + 1]));
      d = ldlt_readlane(col[k + 1], k + 1);
      rd = dev::rcp_nr(d);
      asm volatile("" : "+v"(rd));   // computed HERE (the optimiser otherwise sinks the chain to its first use in the next step)
    }
    if (k > 0) {
#pragma unroll
      for (int ii = k + 2; ii < NB; ++ii) {
        col[ii] -= mprev[ii] * uprev;
        asm volatile("" : "+v"(col[ii]));
      }
    }
#pragma unroll
    for (int ii = k + 2; ii < NB; ++ii) mprev[ii] = Ld[k * NB + ii];
    uprev = col[k];
    if (tid == 0) { dd[k] = d_cur; ddi[k] = rd_cur; }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (idl) {
#pragma unroll
    for (int r = 0; r < NB; ++r) Mi[r * NB + (tid - NB)] = col[r];
  }
  if (zero_pivot && tid == 0) sh_ok = 0;
}

template <int NB, int NT>
__device__ bool ldlt_solve_block(double* __restrict__ A, const double* __restrict__ rhs, const int n, const int W, double* sh,
                                 double*& xs_out, double*& shw_out, const int* __restrict__ lo_pose = nullptr) {
  constexpr int nb = NB;
  constexpr int kSolveThreads = NT;
  const int tid = threadIdx.x;
  double* U = sh;                    // [nb][W]  unscaled panel rows  (d_k * l_jk); the scaled rows l_jk = u_jk / d_k are formed from them
                                     // on the way into the matrix cores (one multiply per operand): ONE panel in LDS, so two windows of
                                     // a batch can share a CU
  double* xs = U + (size_t)nb * W;   // [W]
  double* dd = xs + W;               // [nb]
  double* ddi = dd + nb;             // [nb] reciprocals of the pivots
  double* part = ddi + nb;           // [nb]
  double* Ld = part + nb;            // [nb][nb] diagonal block: padded input, then its scaled factor rows
  double* Mi = Ld + nb * nb;         // [nb][nb] inverse of the block's unit lower-triangular factor (row panel = Mi x A_panel on the matrix cores)
  double* shw = Mi + nb * nb;        // [NT/64] cross-wave scratch for the caller
  int& sh_ok = *reinterpret_cast<int*>(shw + kSolveThreads / 64);
  int* const tcl = reinterpret_cast<int*>(shw + kSolveThreads / 64 + 2);   // [0]: number of live tile columns of the panel, [1..]: their indices
  int* const lo_sh = tcl + 2 * (W / 32 + 4);                               // lo_pose staged (the list is rebuilt for every panel)
  int* const jhi = lo_sh + 2 * (W / 12 + 2);                               // per panel: one past its last live column (global index), for the back substitution
  if (lo_pose) for (int k = tid; k < (n + 5) / 6; k += kSolveThreads) lo_sh[k] = lo_pose[k];
  xs_out = xs;
  shw_out = shw;
  if (tid == 0) sh_ok = 1;
  // the right-hand side lives in LDS (in `xs`) from here on: forward-substituted in place, then overwritten by the solution
  for (int k = tid; k < n; k += kSolveThreads) xs[k] = rhs[k];
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
    // ---- 1. diagonal block, padded with the identity to a full NB x NB block so the factor below is branch-free
    for (int idx = tid; idx < NB * NB; idx += kSolveThreads) {
      const int r = idx / NB, c2 = idx - r * NB;
      const bool in = r < kb && c2 < kb;
      const double a = A[in && c2 >= r ? (size_t)(k0 + r) * n + k0 + c2 : 0];
      Ld[idx] = in ? (c2 >= r ? a : 0.0) : (r == c2 ? 1.0 : 0.0);
    }
    {
      // live tile columns of this panel's trailing part (local tile t covers global columns k0 + kb + 16 t ...): those holding a
      // column whose envelope starts above the panel's last row.  The last wavefront builds the list (the first one factors the block).
      const int Trp = (m - kb + 15) >> 4;
      if (tid >= kSolveThreads - 64) {
        const int ln = tid - (kSolveThreads - 64);
        int base = 0;
        for (int c0 = 0; c0 < Trp; c0 += 64) {
          const int ti = c0 + ln;
          bool act = false;
          if (ti < Trp) {
            act = lo_pose == nullptr;
            if (!act) {
              const int g0 = k0 + kb + 16 * ti, g1 = min(g0 + 16, n) - 1;
              for (int pz = g0 / 6; pz <= g1 / 6; ++pz) act |= 6 * lo_sh[pz] < k0 + kb;
            }
          }
          const unsigned long long mk = __ballot(act);
          if (act) tcl[1 + base + __popcll(mk & ((1ull << ln) - 1ull))] = ti;
          base += __popcll(mk);
        }
        if (ln == 0) tcl[0] = base;
        if (ln == 63) jhi[k0 / nb] = base ? min(n, k0 + kb + 16 * (tcl[base] + 1)) : k0 + kb;   // (lane 63 wrote or saw the list's last entry: same wavefront)
      }
    }
    ldlt_lds_barrier();
    OSH_TR(0);
    // The panel rows of the first live tile of every wavefront but the first are requested now: the loads are in flight while
    // wavefront 0 factors the block (they do not depend on it), instead of after the next barrier.
    constexpr int KSp0 = (NB + 3) / 4;
    double bq0[KSp0];
    bool pre0 = false;
    {
      const int wv0 = __builtin_amdgcn_readfirstlane(tid >> 6);
      if (wv0 != 0 && wv0 < tcl[0]) {
        const int jl = kb + 16 * tcl[1 + wv0] + (tid & 15);
        const char* Ab0 = reinterpret_cast<const char*>(A + (size_t)k0 * n + k0);
        const unsigned n8 = (unsigned)n * 8u, j8 = (unsigned)min(jl, m - 1) * 8u;
#pragma unroll
        for (int ks = 0; ks < KSp0; ++ks) {
          const int k = 4 * ks + ((tid & 63) >> 4);
          bq0[ks] = *reinterpret_cast<const double*>(Ab0 + ((unsigned)min(k, kb - 1) * n8 + j8));
        }
        pre0 = true;
      }
    }
    if (tid < 64) ldlt_factor_diag<NB>(Ld, Mi, dd, ddi, part, tid, sh_ok);
    ldlt_lds_barrier();
    OSH_TR(1);
    if (!sh_ok) break;
    // ---- 2. row panel on the matrix cores: W = L^-1 A[panel rows][live columns], one wavefront per live 16-column tile;
    // C[16 tr + i][j] = sum_k Mi[16 tr + i][k] A[k][j] as v_mfma_f64_16x16x4_f64 steps (Mi is lower triangular: tile row tr needs
    // k < 16 (tr + 1) only).  The unscaled rows go to the LDS panel U for the trailing update, the scaled ones (the factor's rows)
    // to global memory.  (Until round 3 every thread forward-substituted one column with 276 dependent-sweep FMAs and broadcast LDS
    // reads: 12.7 k cycles per panel of a 300 x 300 system against ~2 k for this product.)  The rhs column is Mi y, one lane per row.
    typedef double ldlt_f64x4 __attribute__((ext_vector_type(4)));
    const int na = tcl[0];
    {
      const int wave_p = __builtin_amdgcn_readfirstlane(tid >> 6), lane_p = tid & 63;
      const int prow = lane_p >> 4, pcol = lane_p & 15;
      constexpr int nwaves_p = kSolveThreads / 64;
      constexpr int KSp = (NB + 3) / 4, TRp = (NB + 15) / 16;
      if (wave_p == nwaves_p - 1 && lane_p < NB) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) acc += Mi[lane_p * NB + k] * (k < kb ? xs[k0 + k] : 0.0);
        U[lane_p * W + m] = acc;
        if (lane_p < kb) xs[k0 + lane_p] = acc;
      }
      if (na > 0) {
        double am[TRp][KSp];
#pragma unroll
        for (int tr = 0; tr < TRp; ++tr)
#pragma unroll
          for (int ks = 0; ks < KSp; ++ks) {
            const int i = 16 * tr + pcol, k = 4 * ks + prow;
            const bool in = i < NB && k < NB;
            am[tr][ks] = Mi[in ? i * NB + k : 0];
            if (!in) am[tr][ks] = 0.0;
          }
        char* const Ab = reinterpret_cast<char*>(A + (size_t)k0 * n + k0);
        const unsigned n8 = (unsigned)n * 8u;
        for (int ai = wave_p; ai < na; ai += nwaves_p) {
          const int jl = kb + 16 * tcl[1 + ai] + pcol;     // local column of this lane
          const bool jin = jl < m;
          const unsigned j8 = (unsigned)min(jl, m - 1) * 8u;
          double bq[KSp];
#pragma unroll
          for (int ks = 0; ks < KSp; ++ks) {
            const int k = 4 * ks + prow;
            double v;
            if (pre0 && ai == wave_p) v = bq0[ks];   // requested before the block factorisation
            else v = *reinterpret_cast<const double*>(Ab + ((unsigned)min(k, kb - 1) * n8 + j8));
            bq[ks] = k < kb ? v : 0.0;
          }
          ldlt_f64x4 acc[TRp];
#pragma unroll
          for (int tr = 0; tr < TRp; ++tr) {
            acc[tr] = (ldlt_f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KSp; ++ks)
              if (4 * ks < 16 * (tr + 1)) acc[tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[tr][ks], bq[ks], acc[tr], 0, 0, 0);
          }
#pragma unroll
          for (int tr = 0; tr < TRp; ++tr)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const int r = 16 * tr + prow + 4 * reg;
              if (r < NB && jin) {
                const double wv = acc[tr][reg];
                U[r * W + jl] = wv;
                if (r < kb) *reinterpret_cast<double*>(Ab + ((unsigned)r * n8 + j8)) = wv * ddi[r];
              }
            }
        }
      }
    }
    // the factored diagonal block: pivots on the diagonal, l_rj to their right
    for (int idx = tid; idx < NB * NB; idx += kSolveThreads) {
      const int r = idx / NB, c2 = idx - r * NB;
      if (r < kb && c2 < kb && c2 >= r) A[(size_t)(k0 + r) * n + k0 + c2] = c2 == r ? dd[r] : Ld[idx];
    }
    // zero padding so the 16-wide tiles below may over-read (W >= n + 17, see ldlt_row_stride)
    for (int idx = tid; idx < kb * 16; idx += kSolveThreads) {
      const int r = idx >> 4, jj = m + 1 + (idx & 15);
      U[r * W + jj] = 0.0;
    }
    const int tr = m - kb;  // trailing rows
    const int Tr = na;                    // live tile columns only: a tile (a, b) is touched when both of its tile columns are live
    const int ntile = Tr * (Tr + 1) / 2;  // upper triangle of 16x16 tiles, diagonal tiles included
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // wave index as a scalar: tile decode on the SALU
    const int lrow = lane >> 4, lcol = lane & 15;
    // All addressing is 32-bit byte offsets from the panel origin (uniform base in SGPRs), every load and store is
    // unconditional: an out-of-range row or column is clamped for the load and its store goes to `sink_off`, an entry of the
    // never-read lower triangle; lanes below the diagonal of a diagonal tile update their own (unused) lower-triangle entry.
    // The per-element cost is what bounded this phase before: ~600 VALU/SALU instructions per pair of tiles against 12 MFMAs.
    char* const Ab = reinterpret_cast<char*>(A + (size_t)k0 * n + k0);
    const unsigned n8 = (unsigned)n * 8u;
    const unsigned sink_off = (unsigned)kb * n8;
    const int cA = lrow * W + lcol;
    struct Tile { int i0, j0; unsigned off[4]; double old[4]; };
    auto prep = [&](int t, Tile& T) {
      int ti = 0, rem = t;
      while (rem >= Tr - ti) { rem -= Tr - ti; ++ti; }
      T.i0 = kb + 16 * tcl[1 + ti];
      T.j0 = kb + 16 * tcl[1 + ti + rem];
      const unsigned j8 = (unsigned)min(T.j0 + lcol, m - 1) * 8u;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ii = min(T.i0 + lrow + 4 * reg, m - 1);
        T.off[reg] = __umul24((unsigned)ii, n8) + j8;
        T.old[reg] = *reinterpret_cast<const double*>(Ab + T.off[reg]);
      }
    };
    auto finish = [&](const Tile& T, const ldlt_f64x4& acc) {
      const bool jin = T.j0 + lcol < m;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const bool in = jin && T.i0 + lrow + 4 * reg < m;
        unsigned o = in ? T.off[reg] : sink_off;
        asm volatile("" : "+v"(o));  // one store through a selected offset, not two stores behind a branch
        *reinterpret_cast<double*>(Ab + o) = T.old[reg] - acc[reg];
      }
    };
    // two tiles per step: their MFMA chains are independent, so the matrix core issues back to back instead of waiting
    // for each accumulator; all operand reads of a step are issued before its first MFMA
    constexpr int KSq = (NB + 3) / 4;
    double dq[KSq];   // 1 / d_k of the lane's operand rows k = 4 q + lrow: the A operand is l_ki = u_ki / d_k
#pragma unroll
    for (int q = 0; q < KSq; ++q) dq[q] = (4 * q + lrow < NB) ? ddi[4 * q + lrow] : 0.0;
    auto run2 = [&](const Tile& T0, const Tile& T1) {
      constexpr int KS = (NB + 3) / 4;
      double a0[KS], b0[KS], a1[KS], b1[KS];
      const double* pa0 = U + cA + T0.i0; const double* pb0 = U + cA + T0.j0;
      const double* pa1 = U + cA + T1.i0; const double* pb1 = U + cA + T1.j0;
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        if (4 * q + 3 < NB) {
          a0[q] = pa0[4 * q * W] * dq[q]; b0[q] = pb0[4 * q * W];
          a1[q] = pa1[4 * q * W] * dq[q]; b1[q] = pb1[4 * q * W];
        } else {
          // last k step of a panel whose width is not a multiple of 4: rows >= NB do not exist, their lanes multiply zeros
          const bool kin = 4 * q + lrow < NB;
          const int back = kin ? 0 : lrow * W;
          a0[q] = pa0[4 * q * W - back] * dq[q]; b0[q] = pb0[4 * q * W - back];
          a1[q] = pa1[4 * q * W - back] * dq[q]; b1[q] = pb1[4 * q * W - back];
          if (!kin) { a0[q] = 0.0; b0[q] = 0.0; a1[q] = 0.0; b1[q] = 0.0; }
        }
      }
      ldlt_f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], acc1, 0, 0, 0);
      }
      finish(T0, acc0);
      finish(T1, acc1);
    };
    constexpr int nwaves = kSolveThreads / 64;
    // wave w takes tiles w, w + nwaves, ... two at a time; the loads of the next pair are in flight during the current one.
    // An odd tile out is paired with a copy of itself (both copies were loaded before either is stored, so the second
    // store repeats the first), and past the last pair the prefetch re-reads the current one and is dropped.
    Tile TA0, TA1, TB0, TB1;
    auto prep_pair = [&](int t, Tile& X, Tile& Y) {
      prep(t, X);
      prep(t + nwaves < ntile ? t + nwaves : t, Y);
    };
    // the old values of every wavefront's first pair of trailing tiles are requested before the barrier (the row panel does not
    // write them): in flight while the slower wavefronts finish the panel
    const bool have_tiles = tr > 0 && wave < ntile;
    if (have_tiles) prep_pair(wave, TA0, TA1);
    ldlt_lds_barrier();   // (the factor rows stored above are read again only by the back substitution)
    OSH_TR(2);
    // ---- 3. trailing update of rows k0+kb .. n-1 (upper part) on the FP64 matrix cores: one wavefront per 16x16 tile of
    // the trailing block, C -= L[16 x NB] U[NB x 16] as NB/4 v_mfma_f64_16x16x4_f64 (a partial panel is the last one and has
    // no trailing block).  Operand lanes read straight from the panel in LDS: A[i = lane & 15][k = lane >> 4] = U[k][i0 + i] / d_k,
    // B[k][j] = U[k][j0 + j]; lane holds D[row = (lane >> 4) + 4 reg][col = lane & 15].  (A 4x4 register-tiled VALU version
    // spent 3 of 4 issue slots on operand traffic: 44 k cycles per panel on one CU against ~10 k for the first MFMA version.)
    if (tr > 0) {
      // right-hand side: y_i -= sum_k l_ki y_k, one thread per row, panels read conflict-free
      for (int q = tid; q < 16 * na; q += kSolveThreads) {
        const int ii = kb + 16 * tcl[1 + (q >> 4)] + (q & 15);
        if (ii >= m) continue;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NB; ++k) acc += (U[k * W + ii] * ddi[k]) * U[k * W + m];
        xs[k0 + ii] -= acc;
      }
      if (have_tiles) {
        int t = wave;
        while (true) {
          const int tb = t + 2 * nwaves;
          const bool hasB = tb < ntile;
          prep_pair(hasB ? tb : t, TB0, TB1);
          run2(TA0, TA1);
          if (!hasB) break;
          const int ta = tb + 2 * nwaves;
          const bool hasA = ta < ntile;
          prep_pair(hasA ? ta : tb, TA0, TA1);
          run2(TB0, TB1);
          if (!hasA) break;
          t = ta;
        }
      }
    }
    __syncthreads();
    OSH_TR(3);
  }
  const int ok = sh_ok;
  if (ok) {
    // back substitution  L^T x = D^-1 y, panels in reverse.  Per panel: the global loads of the diagonal block and of every
    // row's tail are issued together (one L2 round trip, not one per row), then one wavefront eliminates the block's columns
    // right to left with the row of each lane in registers and the solved entry broadcast by v_readlane.
    const int npanel = (n + nb - 1) / nb;
    const int wv = tid >> 6, lane = tid & 63;
    constexpr int nwaves = kSolveThreads / 64;
    constexpr int kRowsPerWave = (NB + nwaves - 1) / nwaves;
    constexpr int kDiagPerThread = (NB * NB + kSolveThreads - 1) / kSolveThreads;
    for (int pi = npanel - 1; pi >= 0; --pi) {
      const int k0 = pi * nb;
      const int kb = min(nb, n - k0);
      const int tail0 = k0 + kb;  // x known for indices >= tail0
      const int jend = jhi[pi];   // the rows of this panel are zero from here on (outside the envelope)
      // diagonal block of the factor, padded: row r right of the diagonal, zero elsewhere
      double dg[kDiagPerThread];
#pragma unroll
      for (int q = 0; q < kDiagPerThread; ++q) {
        const int idx = tid + q * kSolveThreads;
        const int r = idx / NB, c2 = idx - r * NB;
        const bool in = idx < NB * NB && r < kb && c2 < kb && c2 >= r;
        dg[q] = A[in ? (size_t)(k0 + r) * n + k0 + c2 : 0];
        if (!in) dg[q] = 0.0;
      }
      // part[r] = sum_{j>=tail0} L[k0+r][j] x[j] : one wavefront per group of rows, lanes stride the columns
      // (the tails of ALL rows of the wavefront, four chunks of 64 columns at a time, are requested before the first is used: written
      // row by row -- `for (j ...) sacc += row[j] * xs[j]` inside the loop over the rows -- every row and every 64 columns of it was a
      // round trip to L2 of its own, 7 to 15 per panel; the sums run over j in the same order)
      constexpr int kTailChunks = 4;
      double sacc[kRowsPerWave];
      unsigned rowb[kRowsPerWave];
#pragma unroll
      for (int q = 0; q < kRowsPerWave; ++q) {
        const int r = wv + q * nwaves;
        rowb[q] = (unsigned)(k0 + (r < kb ? r : 0)) * (unsigned)n;
        sacc[q] = 0.0;
      }
      for (int j0 = tail0 + lane; j0 < jend; j0 += 64 * kTailChunks) {
        double av[kRowsPerWave][kTailChunks];
#pragma unroll
        for (int q = 0; q < kRowsPerWave; ++q)
#pragma unroll
          for (int u = 0; u < kTailChunks; ++u) av[q][u] = A[rowb[q] + (unsigned)min(j0 + 64 * u, jend - 1)];
#pragma unroll
        for (int u = 0; u < kTailChunks; ++u) {
          const int j = j0 + 64 * u;
          if (j < jend) {
            const double xj = xs[j];
#pragma unroll
            for (int q = 0; q < kRowsPerWave; ++q) sacc[q] += av[q][u] * xj;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < kRowsPerWave; ++q) {
        const int r = wv + q * nwaves;
        const double sred = dev::wave_sum(sacc[q]);
        if (lane == 0 && r < NB) part[r] = r < kb ? sred : 0.0;
      }
#pragma unroll
      for (int q = 0; q < kDiagPerThread; ++q) {
        const int idx = tid + q * kSolveThreads;
        if (idx < NB * NB) Ld[idx] = dg[q];
      }
      if (tid < NB) dd[tid] = tid < kb ? xs[k0 + tid] : 0.0;
      __syncthreads();
      if (wv == 0) {
        // lane r holds s_r = y_r / d_r - part_r and row r of the block; lanes beyond the block hold zeros
        const int lr = lane < NB ? lane : 0;
        double urow[NB];
#pragma unroll
        for (int c2 = 0; c2 < NB; ++c2) urow[c2] = Ld[lr * NB + c2];
        const double piv = Ld[lr * NB + lr];
        double sv = (lane < kb) ? dd[lr] / piv - part[lr] : 0.0;
#pragma unroll
        for (int c2 = NB - 1; c2 >= 1; --c2) {
          const double xc = ldlt_readlane(sv, c2);
          const double nv = sv - urow[c2] * xc;
          sv = lane < c2 ? nv : sv;
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
