// Micro-benchmark (not product code): what bounds the diagonal-block factorisation of ldlt_block.h -- ONE wavefront, lane j holds
// column j of a 24 x 24 block, 24 pivots, row k broadcast to the other rows?
//   hipcc --offload-arch=gfx950 -O3 -o pivot_chain pivot_chain.hip && ./pivot_chain
// V0: per row two v_readlane_b32 into the same scalar pair + v_fma_f64 (round 2/3)      V1: multipliers of seven rows first, then seven FMAs
// V2: only the NEXT row through v_readlane; the other rows take their multipliers from the LDS row the step stores anyway (broadcast
//     ds_read_b128), applied one step later, between the start and the use of the next pivot's reciprocal
// V3: the dependent part alone (next row + pivot + reciprocal)                          V4: 24 x 11 independent v_fma_f64 alone
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

constexpr int NB = 24;

__device__ __forceinline__ double rl(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rcp_nr(double z) {
  double x = __builtin_amdgcn_rcp(z);
  x = fma(x, fma(-z, x, 1.0), x);
  x = fma(x, fma(-z, x, 1.0), x);
  return x;
}

template <int V>
__device__ __forceinline__ void factor(const double* Ain, double* Ld, double* Mi, double* dd, double* ddi, double* part, int tid) {
  double col[NB];
  const int cj = tid < NB ? tid : 0;
  const bool idl = tid >= NB && tid < 2 * NB;
#pragma unroll
  for (int r = 0; r < NB; ++r) { const double v = Ain[r * NB + cj]; col[r] = idl ? (r == tid - NB ? 1.0 : 0.0) : v; }
  double* lout = tid < NB ? Ld + tid : part + tid;
  const int ls = tid < NB ? NB : 0;
  double d = rl(col[0], 0);
  double rd = rcp_nr(d);
  if constexpr (V == 0 || V == 1 || V == 3 || V == 4) {
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const double lk = col[k] * rd;
      const double d_cur = d, rd_cur = rd;
      if (V != 4 && k + 1 < NB) {
        col[k + 1] -= rl(lk, k + 1) * col[k];
        asm volatile("" : "+v"(col[k + 1]));
        d = rl(col[k + 1], k + 1);
        rd = rcp_nr(d);
        asm volatile("" : "+v"(rd));
      }
      if constexpr (V == 0) {
#pragma unroll
        for (int ii = k + 2; ii < NB; ++ii) {
          col[ii] -= rl(lk, ii) * col[k];
          asm volatile("" : "+v"(col[ii]));
        }
      } else if constexpr (V == 1) {
#pragma unroll
        for (int i0 = k + 2; i0 < NB; i0 += 7) {
          double mq[7];
#pragma unroll
          for (int q = 0; q < 7; ++q) mq[q] = rl(lk, i0 + q < NB ? i0 + q : NB - 1);
          asm volatile("" : "+s"(mq[0]), "+s"(mq[1]), "+s"(mq[2]), "+s"(mq[3]), "+s"(mq[4]), "+s"(mq[5]), "+s"(mq[6]));
#pragma unroll
          for (int q = 0; q < 7; ++q)
            if (i0 + q < NB) {
              col[i0 + q] -= mq[q] * col[k];
              asm volatile("" : "+v"(col[i0 + q]));
            }
        }
      } else if constexpr (V == 4) {
#pragma unroll
        for (int ii = k + 2; ii < NB; ++ii) {
          col[ii] -= lk * col[k];
          asm volatile("" : "+v"(col[ii]));
        }
      }
      lout[k * ls] = tid > k ? lk : 0.0;
      if (tid == 0) { dd[k] = d_cur; ddi[k] = rd_cur; }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // V2
    double mprev[NB];     // multipliers of the previous pivot (rows k + 1 .. NB - 1 of step k - 1), from LDS
    double uprev = 0.0;   // that pivot's row entry of this lane's column
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const double lk = col[k] * rd;
      const double d_cur = d, rd_cur = rd;
      lout[k * ls] = tid > k ? lk : 0.0;
      // next row: previous pivot's deferred update first (it is the oldest), then this pivot's
      if (k + 1 < NB) {
        if (k > 0) col[k + 1] -= mprev[k + 1] * uprev;
        col[k + 1] -= rl(lk, k + 1) * col[k];
        asm volatile("" : "+v"(col[k + 1]));
        d = rl(col[k + 1], k + 1);
        rd = rcp_nr(d);
        asm volatile("" : "+v"(rd));
      }
      // the previous pivot's other rows
      if (k > 0) {
#pragma unroll
        for (int ii = k + 2; ii < NB; ++ii) {
          col[ii] -= mprev[ii] * uprev;
          asm volatile("" : "+v"(col[ii]));
        }
      }
      // this pivot's multipliers for rows k + 2 ..: requested now, used in the next step
#pragma unroll
      for (int ii = k + 2; ii < NB; ++ii) mprev[ii] = Ld[k * NB + ii];
      uprev = col[k];
      if (tid == 0) { dd[k] = d_cur; ddi[k] = rd_cur; }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (idl) {
#pragma unroll
    for (int r = 0; r < NB; ++r) Mi[r * NB + (tid - NB)] = col[r];
  }
}

// V7: V2 without the per-step bookkeeping (pivot and reciprocal stored by thread 0 behind an exec-mask branch, zero fill of the row's left
// part): lane j keeps its own pivot (a select per step) and stores it with its reciprocal once, after the loop.  V8: + Newton's second
// step from the square of the first residual (one level less on the chain).
template <int V>
__device__ __forceinline__ void factor3(const double* Ain, double* Ld, double* Mi, double* dd, double* ddi, double* part, int tid) {
  double col[NB];
  const int cj = tid < NB ? tid : 0;
  const bool idl = tid >= NB && tid < 2 * NB;
#pragma unroll
  for (int r = 0; r < NB; ++r) { const double v = Ain[r * NB + cj]; col[r] = idl ? (r == tid - NB ? 1.0 : 0.0) : v; }
  double* lout = tid < NB ? Ld + tid : part + tid;
  const int ls = tid < NB ? NB : 0;
  auto rcp2 = [](double z) {
    if constexpr (V == 8) {
      const double x0 = __builtin_amdgcn_rcp(z);
      const double e = fma(-z, x0, 1.0);
      const double x1 = fma(x0, e, x0), e2 = e * e;
      return fma(x1, e2, x1);
    } else {
      return rcp_nr(z);
    }
  };
  double d = rl(col[0], 0);
  double rd = rcp2(d);
  double mine = d, mine_r = rd;
  double mprev[NB];
  double uprev = 0.0;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const double lk = col[k] * rd;
    lout[k * ls] = lk;
    if (k + 1 < NB) {
      if (k > 0) col[k + 1] -= mprev[k + 1] * uprev;
      col[k + 1] -= rl(lk, k + 1) * col[k];
      asm volatile("" : "+v"(col[k + 1]));
      d = rl(col[k + 1], k + 1);
      rd = rcp2(d);
      asm volatile("" : "+v"(rd));
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
    if (k + 1 < NB) { mine = tid == k + 1 ? d : mine; mine_r = tid == k + 1 ? rd : mine_r; }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (tid < NB) { dd[tid] = mine; ddi[tid] = mine_r; }
  if (idl) {
#pragma unroll
    for (int r = 0; r < NB; ++r) Mi[r * NB + (tid - NB)] = col[r];
  }
}

// V5 / V6: the pivot recurrence off the row updates.  d_{k+1} = u_{k+1,k+1} - (u_{k,k+1} / d_k) u_{k,k+1} is formed by EVERY lane from
// two wave-uniform values read (v_readlane) one step ahead, so the chain from one reciprocal to the next is mul, fma, rcp, Newton --
// no v_readlane, no row update on it -- and bit-identical to what lane k + 1 computes for its own diagonal entry.
template <int V>
__device__ __forceinline__ void factor2(const double* Ain, double* Ld, double* Mi, double* dd, double* ddi, double* part, int tid) {
  double col[NB];
  const int cj = tid < NB ? tid : 0;
  const bool idl = tid >= NB && tid < 2 * NB;
#pragma unroll
  for (int r = 0; r < NB; ++r) { const double v = Ain[r * NB + cj]; col[r] = idl ? (r == tid - NB ? 1.0 : 0.0) : v; }
  double* lout = tid < NB ? Ld + tid : part + tid;
  const int ls = tid < NB ? NB : 0;
  double d = rl(col[0], 0);
  double rd = rcp_nr(d);
  double su = rl(col[0], 1), sc = rl(col[1], 1);
  double mine = d;
  double mprev[NB];
  double uprev = 0.0;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const double lk = col[k] * rd;
    lout[k * ls] = lk;
    double dn = d, rdn = rd;
    if (k + 1 < NB) {
      const double t = su * rd;
      dn = fma(-t, su, sc);
      rdn = rcp_nr(dn);
    }
    if constexpr (V == 5) {
#pragma unroll
      for (int i0 = k + 1; i0 < NB; i0 += 7) {
        double mq[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) mq[q] = rl(lk, i0 + q < NB ? i0 + q : NB - 1);
        asm volatile("" : "+s"(mq[0]), "+s"(mq[1]), "+s"(mq[2]), "+s"(mq[3]), "+s"(mq[4]), "+s"(mq[5]), "+s"(mq[6]));
#pragma unroll
        for (int q = 0; q < 7; ++q)
          if (i0 + q < NB) {
            col[i0 + q] -= mq[q] * col[k];
            asm volatile("" : "+v"(col[i0 + q]));
          }
        if (i0 == k + 1 && k + 2 < NB) { su = rl(col[k + 1], k + 2); sc = rl(col[k + 2], k + 2); }
      }
    } else {
      // previous pivot's rows k + 2 .. (multipliers from LDS, requested a step ago), oldest update first
      if (k > 0) {
#pragma unroll
        for (int ii = k + 2; ii < NB; ++ii) {
          col[ii] -= mprev[ii] * uprev;
          asm volatile("" : "+v"(col[ii]));
        }
      }
      if (k + 1 < NB) { col[k + 1] -= rl(lk, k + 1) * col[k]; asm volatile("" : "+v"(col[k + 1])); }
      if (k + 2 < NB) {
        col[k + 2] -= rl(lk, k + 2) * col[k]; asm volatile("" : "+v"(col[k + 2]));
        su = rl(col[k + 1], k + 2); sc = rl(col[k + 2], k + 2);
      }
#pragma unroll
      for (int ii = k + 3; ii < NB; ++ii) mprev[ii] = Ld[k * NB + ii];
      uprev = col[k];
    }
    if (k + 1 < NB) { mine = tid == k + 1 ? dn : mine; }
    d = dn; rd = rdn;
  }
  if (tid < NB) { dd[tid] = mine; ddi[tid] = rcp_nr(mine); }
  if (idl) {
#pragma unroll
    for (int r = 0; r < NB; ++r) Mi[r * NB + (tid - NB)] = col[r];
  }
}

template <int V>
__global__ __launch_bounds__(64) void k(const double* A, double* out, long long* cyc, int reps) {
  __shared__ double Ain[NB * NB], Ld[NB * NB], Mi[NB * NB], dd[NB], ddi[NB], part[64];
  const int tid = threadIdx.x;
  for (int i = tid; i < NB * NB; i += 64) Ain[i] = A[i];
  __syncthreads();
  if constexpr (V >= 7) factor3<V>(Ain, Ld, Mi, dd, ddi, part, tid); else if constexpr (V >= 5) factor2<V>(Ain, Ld, Mi, dd, ddi, part, tid); else factor<V>(Ain, Ld, Mi, dd, ddi, part, tid);   // warm (instruction cache)
  __syncthreads();
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    if constexpr (V >= 7) factor3<V>(Ain, Ld, Mi, dd, ddi, part, tid); else if constexpr (V >= 5) factor2<V>(Ain, Ld, Mi, dd, ddi, part, tid); else factor<V>(Ain, Ld, Mi, dd, ddi, part, tid);
    __syncthreads();
  }
  const long long t1 = clock64();
  if (tid == 0) cyc[0] = (t1 - t0) / reps;
  for (int i = tid; i < NB * NB; i += 64) { out[i] = Ld[i]; out[NB * NB + i] = Mi[i]; }
  if (tid < NB) { out[2 * NB * NB + tid] = dd[tid]; out[2 * NB * NB + NB + tid] = ddi[tid]; }
}

template <int V>
void run(const double* dA, std::vector<double>& res, const char* what) {
  double* dout; long long* dc;
  hipMalloc(&dout, sizeof(double) * (2 * NB * NB + 2 * NB));
  hipMalloc(&dc, sizeof(long long));
  hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, dA, dout, dc, 200);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, dc, sizeof(c), hipMemcpyDeviceToHost);
  res.resize(2 * NB * NB + 2 * NB);
  hipMemcpy(res.data(), dout, sizeof(double) * res.size(), hipMemcpyDeviceToHost);
  printf("V%d %-58s %7lld cycles per block = %5.0f per pivot\n", V, what, c, (double)c / NB);
  hipFree(dout); hipFree(dc);
}

int main() {
  std::vector<double> A(NB * NB), B(NB * NB);
  unsigned s = 12345;
  for (auto& v : B) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / (1 << 24) - 0.5; }
  for (int i = 0; i < NB; ++i)
    for (int j = 0; j < NB; ++j) {
      double a = i == j ? NB : 0.0;
      for (int q = 0; q < NB; ++q) a += B[i * NB + q] * B[j * NB + q];
      A[i * NB + j] = j >= i ? a : 0.0;   // upper triangle, as ldlt_block.h stages it
    }
  double* dA;
  hipMalloc(&dA, sizeof(double) * NB * NB);
  hipMemcpy(dA, A.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
  std::vector<double> r0, r1, r2, r3, r4, r5, r6, r7, r8;
  run<0>(dA, r0, "row by row (2 readlane + fma)");
  run<1>(dA, r1, "seven rows' multipliers, then seven fma");
  run<2>(dA, r2, "next row by readlane, others from the LDS row, one step late");
  run<3>(dA, r3, "dependent part alone");
  run<4>(dA, r4, "the independent fma alone (wrong results)");
  run<5>(dA, r5, "pivot recurrence off the row updates, readlane rows");
  run<6>(dA, r6, "pivot recurrence off the row updates, LDS rows");
  run<7>(dA, r7, "V2 without per-step bookkeeping");
  run<8>(dA, r8, "V7 + shorter Newton");
  auto diff = [&](const std::vector<double>& a, const std::vector<double>& b) {
    double m = 0;
    for (int r = 0; r < NB; ++r)
      for (int c = r + 1; c < NB; ++c) m = fmax(m, fabs(a[r * NB + c] - b[r * NB + c]));             // factor rows right of the diagonal
    for (int i = 0; i < NB * NB; ++i) m = fmax(m, fabs(a[NB * NB + i] - b[NB * NB + i]));            // L^-1
    for (int i = 0; i < 2 * NB; ++i) m = fmax(m, fabs(a[2 * NB * NB + i] - b[2 * NB * NB + i]) / fabs(a[2 * NB * NB + i]));
    return m;
  };
  printf("max |V1 - V0| = %.3g   max |V2 - V0| = %.3g   max |V5 - V0| = %.3g   max |V6 - V0| = %.3g\n", diff(r1, r0), diff(r2, r0), diff(r5, r0), diff(r6, r0));
  printf("max |V7 - V0| = %.3g   max |V8 - V0| = %.3g\n", diff(r7, r0), diff(r8, r0));
  return 0;
}
