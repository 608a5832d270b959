// liba_device.hip -- Optimizer::LocalInertialBA's optimisation (src/Optimizer.cc:2843-2848) on MI355X.
//
// An inertial window is small (N <= 10/25 keyframes x 15 dof, O(10^3) landmarks, O(10^4) edges) and the reference runs exactly
// one at a time, so the whole Levenberg-Marquardt loop of a window runs inside ONE launch with no host round trip: a GROUP of
// up to 32 thread blocks -- one XCD -- works on the window (the tracker's single window: 32 blocks; a large batch: one block per
// window fills the chip).  The phases are separated by a barrier of the group (arrival counter in global memory); every sum is
// taken in a fixed order (lanes, wavefronts, blocks), so a run is bitwise reproducible and every block takes the same controller
// decisions from the same numbers.  Phases (each a function of its own, see the note above liba_landmark_pass):
//   errors      computeActiveErrors + activeRobustChi2: an edge per thread and round, the links on the blocks' last wavefronts
//   linearise   buildSystem: an edge per thread, walked pose by pose: its Hpl block and its terms of Hll, b_l (a row of `eh`) and
//               of Hpp, b_p (a column of `ep`); EdgeInertial Jacobians one lane per link, J^T W J by the link's block
//   sums        per landmark (a team of 1/2/4 lanes): Hll, b_l, Dinv = (Hll + lambda I)^-1, B Dinv; per pose: Hpp, b_p in chunks;
//               the links' blocks into H, links of one colour (no shared keyframe) at a time
//   Schur       S = H + lambda I - sum_l (B Dinv)_i B_i2^T, one wavefront per pose pair; rhs
//   solve       blocked LDL^T (ldlt_block.h) by block 0
//   update      landmark back-substitution (teams), ImuCamPose::Update, computeScale; then the trial's errors and the gain ratio
// Vertex order of the reduced system = g2o's: the 6-dof poses of the N temporal keyframes, then (v, bg, ba).
#include "common.h"
#include "lba_math.h"
#include "ldlt_block.h"
#include "liba_math.h"
#include "liba_edges.h"
#include <algorithm>
#include <cfloat>
#include <cstring>
#include <mutex>
#include <vector>

namespace osh {

constexpr int kLT = 256;      // threads of a block: one wavefront per SIMD, so a phase may use all 512 registers (with 512 threads the
                              // per-edge code spilled: 1.2 KB of scratch per lane)
constexpr int kLNB = 24;      // LDL^T panel width (12 or 6 for windows whose 24-wide panels do not fit LDS: k_liba<NB>)
constexpr int kLG = 32;       // blocks per window at most (one XCD's worth of a group)
constexpr int kPB = 16;       // pivots per step of the group factorisation (liba_solve_group)
constexpr int kPoseChunks = 8;   // a pose row's edges are summed in at most this many chunks
constexpr int kLinkQ = 832;   // per link: J^T W J (24x24), -J^T W r (24), then J (9x24), -W r (9), rho'
// LDS scratch: the LDL^T panels of the reduced system when they fit one block's LDS (NB = 24: up to 51 keyframes, every LocalInertialBA /
// MergeInertialBA window, liba_solve); beyond that the group factorises in global memory (NB = 6 names that variant, liba_solve_group)
// and LDS holds its 16 x 16 blocks and two vectors only.  600 keyframes = a dense 9000 x 9000 system (H and S: 1.3 GB).
constexpr int kLibaMaxKeyframes = 1200;
__host__ __device__ constexpr size_t liba_scratch_doubles(int NB, int W) {
  const size_t need = NB == kLNB ? ldlt_lds_doubles(NB, W, kLT) : (size_t)(6 * 256 + 64 + W + 32);
  return need > 512 ? need : 512;
}
struct LibaOut {
  double chi2_initial, chi2_final;
  int iterations, trials, n_trace, sel;
  double chi2_trace[OSH_LBA_MAX_TRACE], lambda_trace[OSH_LBA_MAX_TRACE];
  int trials_trace[OSH_LBA_MAX_TRACE];
  long long prof[8];   // shader-clock cycles of block 0 per phase: linearise, assembly, Dinv, Schur, LDL^T, back-substitution, errors, outputs
};

struct LibaView {
  const LibaDesc* desc;
  LibaOut* out;
  double* pose[2];            // [K*24] per buffer: Rcw(9) tcw(3) Rwb(9) twb(3)
  double* vba[2];             // [NV*9]: v(3) bg(3) ba(3)
  double* pts[2];             // [L*3]
  const int* e_pose; const int* e_point; const unsigned char* e_kind; const double* e_obs; const double* e_info; const int* e_orig;
  const int* lm_off;          // L+1 per window (sorted edges)
  const int* pel_off; const int* pel_edge;   // pel_off [N+1]: the optimisable poses' ranges in pel_edge [E]: their edges pose by pose (landmark order inside), then the fixed keyframes' edges
  const int* lm_pose_edge;    // [L*N] place of the block of (landmark, optimisable pose) in the pose-by-pose order (pel_edge), or -1
  const int* link_prev; const int* link_cur; const int* link_bias; const float* link_preint; const double* link_info; const double* link_info_g;
  const double* link_info_a; const unsigned char* link_robust;
  double* Hpl; double* Hll; double* bl; double* dinv;   // [18][EF_total] (a column per block, pose-by-pose order) [L*6] [L*3] [L*9]
  double* BD;                 // [18][EF_total] B Dinv of every optimisable-pose block (Schur step), same layout as Hpl
  double* eh; double* ep;     // [9][E_total], [27][EF_total] per-edge terms of the landmark rows / pose rows (linearisation)
  size_t E_total, EF_total;
  const int* link_colour;     // [NL] links of one colour share no keyframe
  double* bfull;              // [n] b with the pose rows' visual part
  double* H; double* b; double* S; double* bs; double* x;   // [n*n] [n] [n*n] [n] [n]
  double* linkQ;              // [NL*kLinkQ]
  double* ppart;              // [sum N][kPoseChunks][27] pose-row chunk sums: Hpp upper (21), b_p (6)
  unsigned* bar; int* abort_flag; double* red; double* ctrl;   // group barrier counters [nw], abort word, published sums [nw*4*kLG*2], [nw*4]
  int nw;
  int force_heavy;            // diagnostic: keep the agent-scope fences even when a group shares an XCD (OSH_LIBA_HEAVY_BARRIER=1)
  double* out_chi2; unsigned char* out_depth;   // result arena: per edge, caller's order
  int* res_abort; double* res_pose; double* res_vba; double* res_pts;   // result arena: abort word, final [sum N][24], [sum N][9], [sum L][3]
  int test_abort;   // OSH_LIBA_TEST_ABORT: make the first group barrier of the launch give up (exercises the one-block retry)
};

// deterministic block reductions over kLT threads
__device__ __forceinline__ double blk_sum(double v, double* shw) {
  v = dev::wave_sum_dpp(v);
  if ((threadIdx.x & 63) == 0) shw[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < kLT / 64; ++k) t += shw[k];
  __syncthreads();
  return t;
}
__device__ __forceinline__ double blk_max(double v, double* shw) {
  v = dev::wave_max(v);
  if ((threadIdx.x & 63) == 0) shw[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < kLT / 64; ++k) t = fmax(t, shw[k]);
  __syncthreads();
  return t;
}

// place of keyframe i's pose (6 unknowns) / velocity + biases (9 unknowns) in the reduced system, see LibaDesc::il
__device__ __forceinline__ int off_pose(const LibaDesc& d, int i) { return d.il ? 15 * i : 6 * i; }
__device__ __forceinline__ int off_vba(const LibaDesc& d, int i) { return d.il ? 15 * i + 6 : 6 * d.N + 9 * i; }
// reduced-state offset of vertex v (0..5) of link (a -> c), or -1 when the vertex is fixed
__device__ __forceinline__ int link_vertex_offset(const LibaDesc& d, int v, int a, int c, int ab) {
  const int kf = (v == 2 || v == 3) ? ab : (v < 4) ? a : c;   // ab: the keyframe that stores the edge's bias vertices (osh_liba_problem.link_bias)
  if (kf >= d.N) return -1;
  if (v == 0 || v == 4) return off_pose(d, kf);
  const int base = off_vba(d, kf);
  return (v == 1 || v == 5) ? base : (v == 2 ? base + 3 : base + 6);
}

// ---------------------------------------------------------------------------------------------------------------------
// A window is worked on by a GROUP of G thread blocks (G = 1 when many windows are in flight, 8 for the tracker's single
// window): the edge / landmark / pose-pair loops are strided over the whole group, the phases are separated by a barrier
// of the group (an arrival counter in global memory, agent-scope fences either side), sums are published per block and
// added in block order by every block, so each block takes the same controller decisions from the same numbers.
// ---------------------------------------------------------------------------------------------------------------------
struct Grp {
  unsigned* bar;        // arrival counter of this window
  int* abort_flag;      // one word per launch: a barrier that does not complete sets it and every block leaves
  int* res_abort;       // the copy of it that travels back with the results
  double* red;          // [4][kLG][2] published partial sums
  int G, m;             // blocks in the group, this block's rank
  int light;            // 1: every block of the group runs on the same XCD (checked at start), so they share one L2: a barrier then only
                        // has to drain this block's stores to L2 and drop this CU's L1, not write back and invalidate the L2
  unsigned gen, nred;
  int test_abort;       // test hook (OSH_LIBA_TEST_ABORT): the first barrier of the launch gives up at once
};
constexpr unsigned kSpinLimit = 1u << 22;   // polls (about a microsecond each) before a barrier gives up

__device__ __forceinline__ bool grp_sync(Grp& g, int* lds_flag) {
  // every wavefront drains its own stores to L2 first (the counter is per wavefront): the arrival below must not overtake them
  if (g.G > 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (g.G == 1) return true;
  ++g.gen;
  if (threadIdx.x == 0) {
    if (g.light) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else __threadfence();
    atomicAdd(g.bar, 1u);
    const unsigned target = g.gen * (unsigned)g.G;
    int good = g.test_abort ? 0 : 1;
    unsigned spins = 0;
    while (good && __hip_atomic_load(g.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0 && (spins > kSpinLimit || __hip_atomic_load(g.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { good = 0; break; }
    }
    if (!good) { __hip_atomic_store(g.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *g.res_abort = 1; }
    if (g.light) asm volatile("buffer_inv sc1" ::: "memory");
    else __threadfence();
    *lds_flag = good;
  }
  __syncthreads();
  return *lds_flag != 0;
}

// sums of two per-thread values over the whole group (fixed order: lanes, wavefronts, blocks)
__device__ __forceinline__ bool grp_sum2(Grp& g, double& a, double& b, double* shw, int* lds_flag) {
  a = blk_sum(a, shw);
  b = blk_sum(b, shw);
  if (g.G == 1) return true;
  double* slot = g.red + (size_t)(g.nred++ & 3u) * kLG * 2;
  if (threadIdx.x == 0) { slot[g.m * 2] = a; slot[g.m * 2 + 1] = b; }
  if (!grp_sync(g, lds_flag)) return false;
  double sa = 0.0, sb = 0.0;
  for (int k = 0; k < g.G; ++k) { sa += slot[k * 2]; sb += slot[k * 2 + 1]; }
  a = sa; b = sb;
  return true;
}

// Inputs of NE visual edges (clamped to the last edge when past the end) read with every load in flight at once: first the
// indices, then the camera rows of the poses, the landmarks, the observations.  (Left to itself the scheduler of this large
// kernel waits for each load before it issues the next one.)
template <int NE>
struct EdgeIn { int kind[NE], ip[NE], j[NE]; double pose[NE][12], X[NE][3], obs[NE][3], info[NE]; };
template <int NE>
__device__ __forceinline__ void load_edges(const LibaView& v, const LibaDesc& d, const int* e, const double* poses, const double* pts, EdgeIn<NE>& o) {
  size_t ge[NE];
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    ge[u] = (size_t)d.edge_off + min(e[u], d.E - 1);
    o.kind[u] = v.e_kind[ge[u]]; o.ip[u] = v.e_pose[ge[u]]; o.j[u] = v.e_point[ge[u]];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const double* P = poses + 24 * (size_t)o.ip[u];
    const double* X = pts + 3 * (size_t)o.j[u];
#pragma unroll
    for (int k = 0; k < 12; ++k) o.pose[u][k] = P[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o.X[u][k] = X[k]; o.obs[u][k] = v.e_obs[ge[u] * 3 + k]; }
    o.info[u] = v.e_info[ge[u]];
  }
  __builtin_amdgcn_sched_barrier(0);
}

// this thread's share of the robust chi2 of the state in buffer `sel` (computeActiveErrors + activeRobustChi2): one edge per
// thread and round, the inertial links on the last wavefront of the blocks
__device__ __noinline__ double eval_partial(const LibaView& v, const LibaDesc& d, int sel, const Grp& g) {
  const int tid = threadIdx.x;
  const double* poses = v.pose[sel] + (size_t)d.pose_off * 24;
  const double* vba = v.vba[sel] + (size_t)d.vel_off * 9;
  const double* pts = v.pts[sel] + (size_t)d.pt_off * 3;
  double acc = 0.0;
  if ((tid >> 6) == kLT / 64 - 1) {
    for (int l = (tid & 63) * g.G + g.m; l < d.NL; l += 64 * g.G) {
      const int gl = d.link_off + l;
      const int a = v.link_prev[gl], c = v.link_cur[gl], ab = v.link_bias[gl];
      double r[9], s1[9];   // (v, bg, ba) of vertices 1..3: the earlier keyframe's velocity, the biases of keyframe ab
      for (int i = 0; i < 3; ++i) s1[i] = vba[9 * a + i];
      for (int i = 3; i < 9; ++i) s1[i] = vba[9 * ab + i];
      inertial_residual(v.link_preint + (size_t)gl * OSH_PREINT_FLOATS, poses + 24 * a, s1, poses + 24 * c, vba + 9 * c, r);
      const double* Om = v.link_info + (size_t)gl * 81;
      double chi = 0.0;
      for (int i = 0; i < 9; ++i) { double t = 0; for (int j = 0; j < 9; ++j) t += Om[i * 9 + j] * r[j]; chi += r[i] * t; }
      if (v.link_robust[gl]) { double r0, r1; dev::huber(chi, d.huber_inertial, r0, r1); chi = r0; }
      acc += chi;
      for (int which = 0; which < 2; ++which) {
        const double* Og = (which == 0 ? v.link_info_g : v.link_info_a) + (size_t)gl * 9;
        double rb[3];
        for (int i = 0; i < 3; ++i) rb[i] = vba[9 * c + 3 + 3 * which + i] - vba[9 * a + 3 + 3 * which + i];
        for (int i = 0; i < 3; ++i) acc += rb[i] * (Og[i * 3] * rb[0] + Og[i * 3 + 1] * rb[1] + Og[i * 3 + 2] * rb[2]);
      }
    }
  }
  const int GT = g.G * kLT;
  for (int e0 = g.m * kLT + tid; e0 < d.E; e0 += 3 * GT) {   // three edges per thread and step
    const int e[3] = {e0, e0 + GT, e0 + 2 * GT};
    EdgeIn<3> in;
    load_edges<3>(v, d, e, poses, pts, in);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (e[u] >= d.E) continue;
      VisEval ev;
      vis_residual(d, in.kind[u], in.pose[u], in.X[u], in.obs[u], in.info[u], ev);
      double r0, r1;
      dev::huber(ev.chi2, in.kind[u] == OSH_EDGE_STEREO ? d.huber_stereo : d.huber_mono, r0, r1);
      acc += r0;
    }
  }
  return acc;
}

#define OSH_GSYNC() do { if (!grp_sync(g, lds_flag)) return; } while (0)
#define OSH_PROF(i) do { if (prof_on) { const long long _n = clock64(); prof[i] += _n - prof_last; prof_last = _n; } } while (0)

// linearisation of one visual edge: robust weight, weighted residual, JX (3x3), Jp (3x6)
struct EdgeLin { double ww, wr[3], JX[9], Jp[18]; };
__device__ __forceinline__ void lin_edge(const LibaDesc& d, int kind, const double* pose, const double* X, const double* obs, double info, EdgeLin& o) {
  VisEval ev;
  vis_residual(d, kind, pose, X, obs, info, ev);
  double r0, r1;
  dev::huber(ev.chi2, kind == OSH_EDGE_STEREO ? d.huber_stereo : d.huber_mono, r0, r1);
  vis_jacobians(d, kind, pose, ev.Xc, o.JX, o.Jp);
  o.ww = r1 * info;
  o.wr[0] = -(info * ev.r[0]) * r1; o.wr[1] = -(info * ev.r[1]) * r1; o.wr[2] = -(info * ev.r[2]) * r1;
}

// sum over the T = 1, 2 or 4 neighbouring lanes of a landmark's team
__device__ __forceinline__ double team_sum(double v, int T) {
  if (T >= 2) v += __shfl_xor(v, 1, 64);
  if (T >= 4) v += __shfl_xor(v, 2, 64);
  return v;
}

struct LibaCtx { const LibaView* v; const LibaDesc* d; double* sh; double* shw; int G, m, C, win, W; };
// eh [E][9]: per-edge terms of Hll (6) and b_l (3), a row per edge (writer and reader both take whole rows);
// ep [27][EF_total]: per-edge terms of Hpp (21, upper) and b_p (6), a column per edge at its place in its pose's list (written by
// consecutive lanes: the linearisation walks the optimisable edges in that order; read the same way)
#define OSH_LIBA_LOCALS \
  [[maybe_unused]] const LibaView& v = *c.v; [[maybe_unused]] const LibaDesc& d = *c.d; \
  [[maybe_unused]] const int G = c.G, m = c.m, tid = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6; \
  [[maybe_unused]] const int GT = G * kLT, gt = m * kLT + tid, GW = G * (kLT / 64); \
  [[maybe_unused]] const int N = d.N, n = d.n, L = d.L, E = d.E, n6 = 6 * d.N, C = c.C; \
  [[maybe_unused]] double* const sh = c.sh; [[maybe_unused]] double* const shw = c.shw; \
  [[maybe_unused]] double* const H = v.H + d.H_off; [[maybe_unused]] double* const S = v.S + d.H_off; \
  [[maybe_unused]] double* const b = v.b + d.b_off; [[maybe_unused]] double* const bs = v.bs + d.b_off; \
  [[maybe_unused]] double* const xg = v.x + d.b_off; [[maybe_unused]] double* const bfull = v.bfull + d.b_off; \
  [[maybe_unused]] double* const Hpl = v.Hpl + d.pel_off; [[maybe_unused]] double* const BD = v.BD + d.pel_off; \
  [[maybe_unused]] double* const Hll = v.Hll + (size_t)d.pt_off * 6; [[maybe_unused]] double* const bl = v.bl + (size_t)d.pt_off * 3; \
  [[maybe_unused]] double* const dinv = v.dinv + (size_t)d.pt_off * 9; \
  [[maybe_unused]] double* const eh = v.eh + (size_t)d.edge_off * 9; [[maybe_unused]] double* const ep = v.ep + d.pel_off; \
  [[maybe_unused]] const size_t ES = v.E_total, PS = v.EF_total; \
  [[maybe_unused]] const int* const lmo = v.lm_off + d.lmoff_off; [[maybe_unused]] const int* const po = v.pel_off + d.peloff_off; \
  [[maybe_unused]] const int* const lmpe = v.lm_pose_edge + d.lmpose_off; \
  [[maybe_unused]] double* const linkQ = v.linkQ + (size_t)d.link_off * kLinkQ; \
  [[maybe_unused]] double* const ppart = v.ppart + (size_t)(d.b_off / 15) * kPoseChunks * 27; \
  [[maybe_unused]] double* const ctrl = v.ctrl + (size_t)c.win * 4; \
  (void)0;
__device__ __forceinline__ int up21(int r, int c) { const int lo = r < c ? r : c, hi = r < c ? c : r; return lo * 6 - lo * (lo - 1) / 2 + (hi - lo); }

// Each phase is a function of its own (not inlined): the register allocation and the instruction scheduling of a phase are then its
// own business.  Inlined into one kernel body, the pressure of the widest phase put the scheduler into its register-saving mode
// everywhere, which waits for every load before it issues the next one.

// ---- landmark pass: each landmark is worked on by a team of T neighbouring lanes of the blocks [first, first + nblk).
// from_edges: Hll, b_l = sums of the per-edge terms (else read back); with_dinv: Dinv = (Hll + lambda I)^-1, Dinv b_l and
// B Dinv of the landmark's optimisable-pose edges (setLambda on Hll + the first product of block_solver.hpp:381-432)
__device__ __noinline__ void liba_landmark_pass(const LibaCtx& c, bool from_edges, bool with_dinv, double lambda, int first, int nblk) {
  OSH_LIBA_LOCALS
    if (m < first || m >= first + nblk) return;
    const int lanes = nblk * kLT;
    const int T = (4 * L <= lanes) ? 4 : ((2 * L <= lanes) ? 2 : 1);
    const int q = tid & (T - 1);
    for (int j = ((m - first) * kLT + tid) / T; j < L; j += lanes / T) {
      double h[9];
      if (from_edges) {
#pragma unroll
        for (int k = 0; k < 9; ++k) h[k] = 0.0;
        const int e_end = lmo[j + 1];
        for (int e = lmo[j] + q; e < e_end; e += 4 * T) {   // four edges in flight: the loads of one are a full L2 round trip
          double t[4][9];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int eu = min(e + u * T, e_end - 1);
#pragma unroll
            for (int k = 0; k < 9; ++k) t[u][k] = eh[(size_t)eu * 9 + k];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool in = e + u * T < e_end;
#pragma unroll
            for (int k = 0; k < 9; ++k) h[k] += in ? t[u][k] : 0.0;
          }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) h[k] = team_sum(h[k], T);
        if (q == 0) {
#pragma unroll
          for (int k = 0; k < 6; ++k) Hll[(size_t)j * 6 + k] = h[k];
#pragma unroll
          for (int k = 0; k < 3; ++k) bl[(size_t)j * 3 + k] = h[6 + k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) h[k] = Hll[(size_t)j * 6 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) h[6 + k] = bl[(size_t)j * 3 + k];
      }
      if (!with_dinv) continue;
      double Di[9];
      dev::inv3_sym(h[0] + lambda, h[1], h[2], h[3] + lambda, h[4], h[5] + lambda, Di);
      if (q == 0) {
        double* o = dinv + (size_t)j * 9;
        o[0] = Di[0]; o[1] = Di[1]; o[2] = Di[2]; o[3] = Di[4]; o[4] = Di[5]; o[5] = Di[8];
        o[6] = Di[0] * h[6] + Di[1] * h[7] + Di[2] * h[8]; o[7] = Di[3] * h[6] + Di[4] * h[7] + Di[5] * h[8]; o[8] = Di[6] * h[6] + Di[7] * h[7] + Di[8] * h[8];
      }
      for (int i0 = q; i0 < N; i0 += 2 * T) {   // two blocks in flight
        int eu[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) eu[u] = (i0 + u * T < N) ? lmpe[(size_t)j * N + i0 + u * T] : -1;
        __builtin_amdgcn_sched_barrier(0);
        double Bu[2][18];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const double* Be = Hpl + (eu[u] < 0 ? 0 : eu[u]);
#pragma unroll
          for (int k = 0; k < 18; ++k) Bu[u][k] = Be[k * PS];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (eu[u] < 0) continue;
          double* od = BD + eu[u];
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            const double x0 = Bu[u][r * 3], x1 = Bu[u][r * 3 + 1], x2 = Bu[u][r * 3 + 2];
            od[(r * 3 + 0) * PS] = x0 * Di[0] + x1 * Di[1] + x2 * Di[2];
            od[(r * 3 + 1) * PS] = x0 * Di[1] + x1 * Di[4] + x2 * Di[5];
            od[(r * 3 + 2) * PS] = x0 * Di[2] + x1 * Di[5] + x2 * Di[8];
          }
        }
      }
    }
}

// ---- pose rows: Hpp (upper) and b_p of pose i = sum of its edges' terms, one wavefront per (pose, chunk) on blocks [first, first + nblk)
__device__ __noinline__ void liba_pose_pass(const LibaCtx& c, int first, int nblk) {
  OSH_LIBA_LOCALS
    if (m < first || m >= first + nblk) return;
    for (int item = (kLT / 64 - 1 - wave) * nblk + (m - first); item < N * C; item += nblk * (kLT / 64)) {
      const int i = item / C, ch = item - i * C;
      const int cnt = po[i + 1] - po[i];
      const int per = ((cnt + C - 1) / C + 63) / 64 * 64;
      const int lo = po[i] + ch * per, hi = min(po[i + 1], lo + per);
      double a[27];
#pragma unroll
      for (int k = 0; k < 27; ++k) a[k] = 0.0;
      for (int idx = lo + lane; idx < hi; idx += 64) {
        // all 27 loads are issued before the first add (left alone, the scheduler waits for each load in turn)
        double t[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) t[k] = ep[k * PS + idx];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 27; ++k) a[k] += t[k];
      }
#pragma unroll
      for (int k = 0; k < 27; ++k) a[k] = dev::wave_sum_dpp(a[k]);
      if (lane == 0) {
        double* o = ppart + ((size_t)i * kPoseChunks + ch) * 27;
#pragma unroll
        for (int k = 0; k < 27; ++k) o[k] = a[k];
      }
    }
}

// ---- the inertial links into H and b (which the linearisation zeroed): links of one colour share no keyframe and are added
// together by the whole group, one colour per phase; EdgeGyroRW / EdgeAccRW (r = b2 - b1, J = [-I, I], plain information) ride along
__device__ __noinline__ void liba_assemble_links(const LibaCtx& c, int sel, int col) {
  OSH_LIBA_LOCALS
  const double* vba = v.vba[sel] + (size_t)d.vel_off * 9;
    auto vert_of = [](int col) { return col < 6 ? 0 : col < 9 ? 1 : col < 12 ? 2 : col < 15 ? 3 : col < 21 ? 4 : 5; };
    const int vbase[6] = {0, 6, 9, 12, 15, 21};
    {
      for (int idx = gt; idx < d.NL * 648; idx += GT) {
        const int l = idx / 648, sl = idx - l * 648;
        const int gl = d.link_off + l;
        if (v.link_colour[gl] != col) continue;
        const int a = v.link_prev[gl], c = v.link_cur[gl], ab = v.link_bias[gl];
        const double* Q = linkQ + (size_t)l * kLinkQ;
        if (sl < 576) {
          const int ca = sl / 24, cb = sl - ca * 24;
          const int va = vert_of(ca), vb = vert_of(cb);
          if (vb < va) continue;   // upper blocks + mirrored below
          const int oa = link_vertex_offset(d, va, a, c, ab), ob = link_vertex_offset(d, vb, a, c, ab);
          if (oa < 0 || ob < 0) continue;
          double val = Q[sl];
          if (va == vb && (va == 2 || va == 3)) val += (va == 2 ? v.link_info_g : v.link_info_a)[(size_t)gl * 9 + (ca - vbase[va]) * 3 + (cb - vbase[vb])];
          const int ra = oa + (ca - vbase[va]), rb = ob + (cb - vbase[vb]);
          H[(size_t)ra * n + rb] += val;
          if (va != vb) H[(size_t)rb * n + ra] += val;
        } else if (sl < 600) {
          const int ca = sl - 576;
          const int va = vert_of(ca);
          const int oa = link_vertex_offset(d, va, a, c, ab);
          if (oa < 0) continue;
          double val = Q[sl];
          if (va == 2 || va == 3) {
            const int which = va - 2, i = ca - vbase[va];
            const double* Og = (which == 0 ? v.link_info_g : v.link_info_a) + (size_t)gl * 9;
            double rb[3];
            for (int k = 0; k < 3; ++k) rb[k] = vba[9 * c + 3 + 3 * which + k] - vba[9 * a + 3 + 3 * which + k];
            val += Og[i * 3] * rb[0] + Og[i * 3 + 1] * rb[1] + Og[i * 3 + 2] * rb[2];
          }
          b[oa + (ca - vbase[va])] += val;
        } else if (sl < 642) {
          const int t = sl - 600, which = t / 21, u = t - which * 21;
          const double* Og = (which == 0 ? v.link_info_g : v.link_info_a) + (size_t)gl * 9;
          const int o1 = (a < N) ? off_vba(d, a) + 3 + 3 * which : -1;
          const int o2 = off_vba(d, c) + 3 + 3 * which;
          if (u < 9) {
            const int i = u / 3, j = u - i * 3;
            if (o1 >= 0) { H[(size_t)(o1 + i) * n + o2 + j] += -Og[u]; H[(size_t)(o2 + j) * n + o1 + i] += -Og[u]; }
          } else if (u < 18) {
            const int i = (u - 9) / 3, j = (u - 9) - i * 3;
            H[(size_t)(o2 + i) * n + o2 + j] += Og[u - 9];
          } else {
            const int i = u - 18;
            double rb[3];
            for (int k = 0; k < 3; ++k) rb[k] = vba[9 * c + 3 + 3 * which + k] - vba[9 * a + 3 + 3 * which + k];
            b[o2 + i] += -(Og[i * 3] * rb[0] + Og[i * 3 + 1] * rb[1] + Og[i * 3 + 2] * rb[2]);
          }
        }
      }
    }
}

// ---- linearise (buildSystem): zero H and b, Jacobians and quadratic forms of the inertial links, per-edge terms of the visual edges
__device__ __noinline__ void liba_linearise(const LibaCtx& c, int sel) {
  OSH_LIBA_LOCALS
  const double* poses = v.pose[sel] + (size_t)d.pose_off * 24;
  const double* vba = v.vba[sel] + (size_t)d.vel_off * 9;
  const double* pts = v.pts[sel] + (size_t)d.pt_off * 3;
    if (d.il) {
      // banded layout: only the entries within the band exist (both triangles of H are used by the link assembly)
      const long long wrow = 2ll * d.bw + 1, tot = (long long)n * wrow;
      for (long long k = gt; k < tot; k += GT) {
        const int r = (int)(k / wrow), cc = r - d.bw + (int)(k - (long long)r * wrow);
        if (cc >= 0 && cc < n) H[(size_t)r * n + cc] = 0.0;
      }
    } else {
      for (int k = gt; k < n * n; k += GT) H[k] = 0.0;
    }
    for (int k = gt; k < n; k += GT) b[k] = 0.0;
    // inertial links, Jacobians: link l belongs to block l mod G, one lane of that block's last wavefront each
    if (wave == kLT / 64 - 1) {
      for (int l = lane * G + m; l < d.NL; l += 64 * G) {
        const int gl = d.link_off + l;
        const int a = v.link_prev[gl], c = v.link_cur[gl], ab = v.link_bias[gl];
        double* Q = linkQ + (size_t)l * kLinkQ;
        double* J = Q + 600;
        double r[9], s1[9];
        for (int i = 0; i < 3; ++i) s1[i] = vba[9 * a + i];
        for (int i = 3; i < 9; ++i) s1[i] = vba[9 * ab + i];
        const float* rec = v.link_preint + (size_t)gl * OSH_PREINT_FLOATS;
        inertial_residual_jacobian(rec, poses + 24 * a, s1, poses + 24 * c, vba + 9 * c, r, J);
        const double* Om = v.link_info + (size_t)gl * 81;
        double rho1 = 1.0;
        if (v.link_robust[gl]) {
          double chi = 0.0, r0;
          for (int i = 0; i < 9; ++i) { double t = 0; for (int j = 0; j < 9; ++j) t += Om[i * 9 + j] * r[j]; chi += r[i] * t; }
          dev::huber(chi, d.huber_inertial, r0, rho1);
        }
        double* wr = J + 216;
        wr[9] = rho1;
        for (int i = 0; i < 9; ++i) { double t = 0; for (int j = 0; j < 9; ++j) t += Om[i * 9 + j] * r[j]; wr[i] = -t * rho1; }
      }
    }
    // visual edges, one per thread and round: the edge's terms of Hll, b_l (summed per landmark in the next phase), of Hpp, b_p
    // (summed per pose there), and its Hpl block
    // (in a group the wavefronts that hold a link's Jacobian take no edges: the Jacobian is as long as four edge rounds)
    const int nlw = G > 1 ? min(G, d.NL) : 0;                                  // blocks 0 .. nlw-1: their last wavefront is a link wavefront
    const int wpb = kLT / 64;
    const int nfree = nlw * (wpb - 1) + (G - nlw) * wpb;
    const bool link_wave = m < nlw && wave == wpb - 1;
    const int frank = m < nlw ? m * (wpb - 1) + wave : nlw * (wpb - 1) + (m - nlw) * wpb + wave;
    for (int idx = link_wave ? E : frank * 64 + lane; idx < E; idx += nfree * 64) {
      // the optimisable keyframes' edges first, pose by pose (so the columns of ep are written by consecutive lanes), then the rest
      const int e = v.pel_edge[(size_t)d.edge_off + idx];
      const size_t ge = (size_t)d.edge_off + e;
      EdgeIn<1> in;
      load_edges<1>(v, d, &e, poses, pts, in);
      const int kind = in.kind[0], ip = in.ip[0], j = in.j[0];
      const double* X = in.X[0];
      const double* pose = in.pose[0];
      EdgeLin ln;
      lin_edge(d, kind, pose, X, in.obs[0], in.info[0], ln);
      const double* JX = ln.JX; const double* Jp = ln.Jp;
      const double ww = ln.ww;
      eh[(size_t)e * 9 + 0] = (JX[0] * ww) * JX[0] + (JX[3] * ww) * JX[3] + (JX[6] * ww) * JX[6];
      eh[(size_t)e * 9 + 1] = (JX[0] * ww) * JX[1] + (JX[3] * ww) * JX[4] + (JX[6] * ww) * JX[7];
      eh[(size_t)e * 9 + 2] = (JX[0] * ww) * JX[2] + (JX[3] * ww) * JX[5] + (JX[6] * ww) * JX[8];
      eh[(size_t)e * 9 + 3] = (JX[1] * ww) * JX[1] + (JX[4] * ww) * JX[4] + (JX[7] * ww) * JX[7];
      eh[(size_t)e * 9 + 4] = (JX[1] * ww) * JX[2] + (JX[4] * ww) * JX[5] + (JX[7] * ww) * JX[8];
      eh[(size_t)e * 9 + 5] = (JX[2] * ww) * JX[2] + (JX[5] * ww) * JX[5] + (JX[8] * ww) * JX[8];
#pragma unroll
      for (int i = 0; i < 3; ++i) eh[(size_t)e * 9 + 6 + i] = JX[i] * ln.wr[0] + JX[3 + i] * ln.wr[1] + JX[6 + i] * ln.wr[2];
      if (ip < N) {
        const int pp = idx;   // = the edge's place in its pose's list
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = a; c < 6; ++c) { ep[q * PS + pp] = (Jp[a] * ww) * Jp[c] + (Jp[6 + a] * ww) * Jp[6 + c] + (Jp[12 + a] * ww) * Jp[12 + c]; ++q; }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) ep[(21 + a) * PS + pp] = Jp[a] * ln.wr[0] + Jp[6 + a] * ln.wr[1] + Jp[12 + a] * ln.wr[2];
        // Hpl: the right-camera edge of a (keyframe, landmark) pair shares the block of the left edge sorted just before it;
        // the left edge's thread forms both terms
        const bool second = e > 0 && v.e_point[ge - 1] == j && v.e_pose[ge - 1] == ip;
        if (!second) {
          double hv[18];
#pragma unroll
          for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) hv[i * 3 + jj] = (Jp[i] * ww) * JX[jj] + (Jp[6 + i] * ww) * JX[3 + jj] + (Jp[12 + i] * ww) * JX[6 + jj];
          if (d.rig_on && e + 1 < E && v.e_point[ge + 1] == j && v.e_pose[ge + 1] == ip) {
            EdgeLin l2;
            lin_edge(d, v.e_kind[ge + 1], pose, X, v.e_obs + (ge + 1) * 3, v.e_info[ge + 1], l2);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
              for (int jj = 0; jj < 3; ++jj)
                hv[i * 3 + jj] += (l2.Jp[i] * l2.ww) * l2.JX[jj] + (l2.Jp[6 + i] * l2.ww) * l2.JX[3 + jj] + (l2.Jp[12 + i] * l2.ww) * l2.JX[6 + jj];
          }
#pragma unroll
          for (int k = 0; k < 18; ++k) Hpl[k * PS + idx] = hv[k];
        }
      }
    }
    __syncthreads();
    // quadratic forms of this block's links (BaseMultiEdge::constructQuadraticForm): W J first, then J^T (W J)
    for (int l = m; l < d.NL; l += G) {
      const int gl = d.link_off + l;
      double* Q = linkQ + (size_t)l * kLinkQ;
      const double* J = Q + 600;
      const double rho1 = J[216 + 9];
      const double* Om = v.link_info + (size_t)gl * 81;
      extern __shared__ __attribute__((aligned(16))) double lds_lin[];   // the block's dynamic LDS (the LDL^T scratch, free here)
      double* WJ = lds_lin;       // [9][24]
      double* Js = lds_lin + 216; // [9][24]
      for (int idx = tid; idx < 216; idx += kLT) {
        const int k = idx / 24, cb = idx - k * 24;
        double t = 0.0;
        for (int q = 0; q < 9; ++q) t += (rho1 * Om[k * 9 + q]) * J[q * 24 + cb];
        WJ[idx] = t;
        Js[idx] = J[idx];
      }
      __syncthreads();
      for (int idx = tid; idx < 600; idx += kLT) {
        double acc = 0.0;
        if (idx < 576) {
          const int ca = idx / 24, cb = idx - ca * 24;
          for (int k = 0; k < 9; ++k) acc += Js[k * 24 + ca] * WJ[k * 24 + cb];
        } else {
          const int ca = idx - 576;
          for (int k = 0; k < 9; ++k) acc += Js[k * 24 + ca] * J[216 + k];
        }
        Q[idx] = acc;
      }
      __syncthreads();
    }
}

// ---- the same for the banded layout of a map-sized problem (LibaDesc::il): S = H + lambda I only inside the band, the Schur products
// only for keyframe pairs at most bw_kf apart, and for such a pair only over the landmarks keyframe i observes (its edge list)
// instead of over all landmarks: O(N bw_kf deg) instead of O(N^2 L).
__device__ __noinline__ void liba_schur_banded(const LibaCtx& c, double lambda) {
  OSH_LIBA_LOCALS
  {
    // (kPB - 1 columns past the band too: the row panel of a 16-pivot step of liba_solve_group reads its first row that far -- structural
    // zeros, but zeros that have to be there: left unwritten they were whatever an earlier, larger call had put in the arena, and a
    // 400-keyframe map solved after a 128-window batch in the same context ended on another cost with 20 trials instead of 5)
    const long long wrow = (long long)d.bw + kPB, tot = (long long)n * wrow;
    for (long long k = gt; k < tot; k += GT) {
      const int r = (int)(k / wrow), cc = r + (int)(k - (long long)r * wrow);
      if (cc >= n) continue;
      if (cc - r > d.bw) { S[(size_t)r * n + cc] = 0.0; continue; }
      if ((r % 15) < 6 && (cc % 15) < 6) {               // pose-pose blocks: written below for keyframes at most bw_kf apart
        if (cc / 15 - r / 15 > d.bw_kf) S[(size_t)r * n + cc] = 0.0;
        continue;
      }
      S[(size_t)r * n + cc] = H[(size_t)r * n + cc] + ((r == cc) ? lambda : 0.0);
    }
    for (int k = gt; k < n; k += GT) if ((k % 15) >= 6) { bfull[k] = b[k]; bs[k] = b[k]; }
  }
  const int wp = d.bw_kf + 1;
  const long long npairs = (long long)N * wp;
  for (long long pr = wave * G + m; pr < npairs; pr += GW) {
    const int i = (int)(pr / wp), i2 = i + (int)(pr - (long long)i * wp);
    if (i2 >= N) continue;
    double acc[36], ci[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0.0;
    const int p_lo = po[i], p_hi = po[i + 1];
    for (int p0 = p_lo + lane; p0 < p_hi; p0 += 64) {
      const int e = v.pel_edge[(size_t)d.edge_off + p0];
      const int j = v.e_point[(size_t)d.edge_off + e];
      if (lmpe[(size_t)j * N + i] != p0) continue;       // the second edge of a (keyframe, landmark) pair shares the first one's block
      const int e2 = (i2 == i) ? p0 : lmpe[(size_t)j * N + i2];
      if (e2 < 0) continue;
      double a1[18], b2[18];
#pragma unroll
      for (int k = 0; k < 18; ++k) { a1[k] = BD[k * PS + p0]; b2[k] = Hpl[k * PS + e2]; }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double a0 = a1[r * 3], a1v = a1[r * 3 + 1], a2 = a1[r * 3 + 2];
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) acc[r * 6 + cc] += a0 * b2[cc * 3] + a1v * b2[cc * 3 + 1] + a2 * b2[cc * 3 + 2];
      }
      if (i == i2) {
        const double* Dj = dinv + (size_t)j * 9;
        const double d0 = Dj[6], d1 = Dj[7], d2 = Dj[8];
#pragma unroll
        for (int r = 0; r < 6; ++r) ci[r] += b2[r * 3] * d0 + b2[r * 3 + 1] * d1 + b2[r * 3 + 2] * d2;
      }
    }
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = dev::wave_sum_dpp(acc[k]);
    const int oi = off_pose(d, i), oi2 = off_pose(d, i2);
    if (lane < 36) {
      double val = acc[0];
#pragma unroll
      for (int k = 1; k < 36; ++k) val = (lane == k) ? acc[k] : val;
      const int r = lane / 6, cc = lane - r * 6;
      double base = H[(size_t)(oi + r) * n + oi2 + cc];
      if (i == i2) {
        const int qq = up21(r, cc);
        for (int ch = 0; ch < C; ++ch) base += ppart[((size_t)i * kPoseChunks + ch) * 27 + qq];
        if (r == cc) base += lambda;
      }
      S[(size_t)(oi + r) * n + oi2 + cc] = base - val;
    }
    if (i == i2) {
#pragma unroll
      for (int r = 0; r < 6; ++r) ci[r] = dev::wave_sum_dpp(ci[r]);
      if (lane < 6) {
        double cv = ci[0];
        if (lane == 1) cv = ci[1]; else if (lane == 2) cv = ci[2]; else if (lane == 3) cv = ci[3];
        else if (lane == 4) cv = ci[4]; else if (lane == 5) cv = ci[5];
        double bf = b[oi + lane];
        for (int ch = 0; ch < C; ++ch) bf += ppart[((size_t)i * kPoseChunks + ch) * 27 + 21 + lane];
        bfull[oi + lane] = bf;
        bs[oi + lane] = bf - cv;
      }
    }
  }
}

// ---- S = H + lambda I, Schur complement of the landmarks, right-hand side
__device__ __noinline__ void liba_schur(const LibaCtx& c, double lambda) {
  OSH_LIBA_LOCALS
  if (d.il) { liba_schur_banded(c, lambda); return; }
      // S = H + lambda I and the rhs outside the pose-pose blocks (upper triangle)
      for (int k = gt; k < n * n; k += GT) {
        const int r = k / n, c = k - r * n;
        if (c >= r && c >= n6) S[k] = H[k] + ((r == c) ? lambda : 0.0);
      }
      for (int k = n6 + gt; k < n; k += GT) { bfull[k] = b[k]; bs[k] = b[k]; }
      // Schur complement (block_solver.hpp:381-432), landmark-parallel: one wavefront per pose PAIR (i <= i2), the lanes stride
      // the landmarks, a landmark seen by both poses adds (B Dinv)_i B_i2^T to the lane's 6x6 partial, the 36 partials are summed
      // by a fixed butterfly; S(i, i2) = H(i, i2) [links] + Hpp_i + lambda I [i = i2] - that sum.  The diagonal pairs also form
      // their pose's rhs  b_p - B (Dinv b_l).
      const int npairs = N * (N + 1) / 2;
      for (int pr = wave * G + m; pr < npairs; pr += GW) {
        int i = 0, rem = pr;
        while (rem >= N - i) { rem -= N - i; ++i; }
        const int i2 = i + rem;
        double acc[36], ci[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = 0.0;
        for (int j0 = lane; j0 < L; j0 += 128) {   // two landmarks per lane and step, every load of a step issued before its first use
          int e1[2], e2[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int j = min(j0 + 64 * u, L - 1);
            e1[u] = lmpe[(size_t)j * N + i]; e2[u] = lmpe[(size_t)j * N + i2];
            if (j0 + 64 * u >= L) e1[u] = -1;
          }
          __builtin_amdgcn_sched_barrier(0);
          double a1[2][18], b2[2][18], dj[2][3];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const bool on = e1[u] >= 0 && e2[u] >= 0;
            const double* A1 = BD + (on ? e1[u] : 0);
            const double* B2 = Hpl + (on ? e2[u] : 0);
            const double* Dj = dinv + (size_t)min(j0 + 64 * u, L - 1) * 9;
#pragma unroll
            for (int k = 0; k < 18; ++k) { a1[u][k] = A1[k * PS]; b2[u][k] = B2[k * PS]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) dj[u][k] = Dj[6 + k];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (e1[u] < 0 || e2[u] < 0) continue;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
              const double a0 = a1[u][r * 3], a1v = a1[u][r * 3 + 1], a2 = a1[u][r * 3 + 2];
#pragma unroll
              for (int c = 0; c < 6; ++c) acc[r * 6 + c] += a0 * b2[u][c * 3] + a1v * b2[u][c * 3 + 1] + a2 * b2[u][c * 3 + 2];
            }
            if (i == i2) {
#pragma unroll
              for (int r = 0; r < 6; ++r) ci[r] += b2[u][r * 3] * dj[u][0] + b2[u][r * 3 + 1] * dj[u][1] + b2[u][r * 3 + 2] * dj[u][2];
            }
          }
        }
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = dev::wave_sum_dpp(acc[k]);
        if (lane < 36) {
          double val = acc[0];
#pragma unroll
          for (int k = 1; k < 36; ++k) val = (lane == k) ? acc[k] : val;
          const int r = lane / 6, c = lane - r * 6;
          double base = H[(size_t)(6 * i + r) * n + 6 * i2 + c];   // (il = 0 here: poses first)
          if (i == i2) {
            const int qq = up21(r, c);
            for (int ch = 0; ch < C; ++ch) base += ppart[((size_t)i * kPoseChunks + ch) * 27 + qq];
            if (r == c) base += lambda;
          }
          S[(size_t)(6 * i + r) * n + 6 * i2 + c] = base - val;
        }
        if (i == i2) {
#pragma unroll
          for (int r = 0; r < 6; ++r) ci[r] = dev::wave_sum_dpp(ci[r]);
          if (lane < 6) {
            double c = ci[0];
            if (lane == 1) c = ci[1]; else if (lane == 2) c = ci[2]; else if (lane == 3) c = ci[3];
            else if (lane == 4) c = ci[4]; else if (lane == 5) c = ci[5];
            double bf = b[6 * i + lane];
            for (int ch = 0; ch < C; ++ch) bf += ppart[((size_t)i * kPoseChunks + ch) * 27 + 21 + lane];
            bfull[6 * i + lane] = bf;
            bs[6 * i + lane] = bf - c;
          }
        }
      }
}

// ---- LDL^T + solve of the reduced system by block 0 of the group.  A function of its own, NOT inlined: inside k_liba the sixty-odd
// pointers of the view are live across the whole kernel, and inlined there the factorisation's inner loops reloaded spilled scalar
// registers at every step (s_or_saveexec / v_accvgpr_read / v_readlane: 75 instructions per pivot for 45, 15.7 k cycles per diagonal
// block against 8.8 k for the same source in k_solve).  The LDS scratch travels as an address_space(3) pointer so that the body still
// addresses it with ds_ instructions.
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(1))) double glb_double;
template <int NB>
__device__ __noinline__ void liba_solve_fn(glb_double* S1, const glb_double* bs1, const int n, const int W, lds_double* sh3, glb_double* xg1, glb_double* ctrl1) {
  double* const sh_lds = (double*)sh3;
  double* const S = (double*)S1; const double* const bs = (const double*)bs1; double* const xg = (double*)xg1; double* const ctrl = (double*)ctrl1;
  const int tid = threadIdx.x;
  double *xs, *shw2;
  const bool okb = ldlt_solve_block<NB, kLT>(S, bs, n, W, sh_lds, xs, shw2);
  for (int k = tid; k < n; k += kLT) xg[k] = xs[k];
  if (tid == 0) ctrl[1] = okb ? 1.0 : 0.0;
}
template <int NB>
__device__ __forceinline__ void liba_solve(const LibaCtx& c, double* sh_lds) {
  OSH_LIBA_LOCALS
  liba_solve_fn<NB>((glb_double*)S, (const glb_double*)bs, n, c.W, (lds_double*)sh_lds, (glb_double*)xg, (glb_double*)ctrl);
}

// ---- LDL^T + solve of a reduced system too wide for the LDS panels above (FullInertialBA over a map: up to 2880 unknowns), by the WHOLE
// group: right-looking, 16 pivots per step, the matrix stays in global memory (upper storage; A = U^T D^-1 U with the unscaled rows
// u_pj = d_p l_jp kept in place of A, as in ldlt_block.h).  Per step: every block factors the 16 x 16 diagonal block for itself (one
// wavefront, the columns in registers, pivots and multipliers passed by lane shuffles), the row panel is divided among all threads of
// the group, a group barrier, the trailing update A_ij -= sum_p u_pi u_pj / d_p in tiles of 16 rows x 64 columns divided among all
// wavefronts of the group, a group barrier.  Forward and back substitution by block 0, 16 rows at a time, the vectors in LDS.
__device__ __noinline__ bool liba_solve_group(const LibaCtx& c, Grp& g, int* lds_flag) {
  OSH_LIBA_LOCALS
  double* du = sh;                       // [kPB][kPB] row q: u_qp for p >= q (u_qq = d_q)
  double* ddi = du + kPB * kPB;          // [kPB] 1 / d_q
  double* wsh = ddi + kPB;               // [kLT/64][kPB*kPB] per wavefront: u_pi / d_p of the row tile it works on
  double* xs = wsh + (kLT / 64) * kPB * kPB;   // [n] right-hand side -> solution (block 0)
  double* red = xs + ((n + 15) & ~15);   // [kLT/64][kPB] partial sums of the back substitution
  bool ok = true;
#ifdef OSH_LIBA_LDLT_TRACE
  long long gt_t[6] = {0, 0, 0, 0, 0, 0}, gt_last = clock64();
#define OSH_GT(i) do { const long long _n = clock64(); gt_t[i] += _n - gt_last; gt_last = _n; } while (0)
#else
#define OSH_GT(i) do {} while (0)
#endif
  for (int k0 = 0; k0 < n; k0 += kPB) {
    const int kb = min(kPB, n - k0);
    // ---- the diagonal block, by wavefront 0 of every block: lane t holds column k0 + t (rows 0..t)
    if (wave == 0) {
      double col[kPB];
#pragma unroll
      for (int p = 0; p < kPB; ++p) col[p] = (lane < kb && p <= lane) ? S[(size_t)(k0 + p) * n + k0 + lane] : (p == lane ? 1.0 : 0.0);
#pragma unroll
      for (int q = 0; q < kPB; ++q) {
        const double dq = ldlt_readlane(col[q], q);     // pivot: row q of column q, final after the earlier pivots (v_readlane: the lane is a
        const double cq = col[q] * (1.0 / dq);           // compile-time constant; a shuffle is an LDS round trip, 1.4 k cycles per pivot before)
#pragma unroll
        for (int p = q + 1; p < kPB; ++p) {
          const double uqp = ldlt_readlane(col[q], p);    // u_qp: row q of column p
          if (p <= lane) col[p] -= uqp * cq;
        }
      }
      if (lane < kPB) {
#pragma unroll
        for (int p = 0; p < kPB; ++p) du[p * kPB + lane] = col[p];      // column `lane`: u_p,lane (garbage below the diagonal, never read)
        ddi[lane] = 1.0 / col[lane];
      }
    }
    __syncthreads();
    OSH_GT(0);
    for (int q = 0; q < kb; ++q) if (du[q * kPB + q] == 0.0) ok = false;   // Eigen's LDLT fails on an exactly-zero pivot only
    // banded layout: the pivot rows end bw columns right of the diagonal; what lies beyond is not stored
    const int jend = d.il ? min(n, k0 + kb + d.bw) : n;
    // ---- the row panel: column j of rows k0 .. k0+kb, one column per thread of the group
    for (int j = k0 + kb + gt; j < jend; j += GT) {
      double col[kPB];
#pragma unroll
      for (int p = 0; p < kPB; ++p) col[p] = p < kb ? S[(size_t)(k0 + p) * n + j] : 0.0;
#pragma unroll
      for (int q = 0; q < kPB; ++q) {
        const double cq = col[q] * ddi[q];
#pragma unroll
        for (int p = q + 1; p < kPB; ++p) col[p] -= du[q * kPB + p] * cq;
      }
#pragma unroll
      for (int p = 0; p < kPB; ++p) if (p < kb) S[(size_t)(k0 + p) * n + j] = col[p];
    }
    // the right-hand side is one more column of the panel (forward substitution inside the factorisation, as ldlt_block.h does: a pass of
    // its own by block 0 afterwards cost more than the factorisation): its entries of the pivot rows, by the last thread of the group
    if (gt == GT - 1) {
      double col[kPB];
#pragma unroll
      for (int p = 0; p < kPB; ++p) col[p] = p < kb ? bs[k0 + p] : 0.0;
#pragma unroll
      for (int q = 0; q < kPB; ++q) {
        const double cq = col[q] * ddi[q];
#pragma unroll
        for (int p = q + 1; p < kPB; ++p) col[p] -= du[q * kPB + p] * cq;
      }
#pragma unroll
      for (int p = 0; p < kPB; ++p) if (p < kb) bs[k0 + p] = col[p];
    }
    OSH_GT(1);
    if (!grp_sync(g, lds_flag)) return false;
    OSH_GT(2);
    // the factored diagonal block goes back only now: until the barrier the other blocks were still reading the unfactored one
    if (m == 0) {
      const int pp = tid >> 4, tt = tid & 15;
      if (tid < kPB * kPB && pp <= tt && tt < kb) S[(size_t)(k0 + pp) * n + k0 + tt] = du[tid];
    }
    // ---- trailing update, tiles of 16 rows x 64 columns (upper part) over the wavefronts of the group
    const int t0 = k0 + kb;
    if (t0 < n && kb == kPB && m == G - 1) {
      // ... and its entries below the panel: z_i -= sum_p (u_pi / d_p) z_p, by the last block
      double zs[kPB];
#pragma unroll
      for (int p = 0; p < kPB; ++p) zs[p] = bs[k0 + p] * ddi[p];
      for (int i = t0 + tid; i < jend; i += kLT) {
        double acc = 0.0;
#pragma unroll
        for (int p = 0; p < kPB; ++p) acc += S[(size_t)(k0 + p) * n + i] * zs[p];
        bs[i] -= acc;
      }
    }
    if (t0 < n && kb == kPB) {
      double* wt = wsh + wave * kPB * kPB;
      const int gw = m * (kLT / 64) + wave;
      int base = 0;
      for (int i0 = t0; i0 < jend; i0 += kPB) {
        const int nc = (jend - i0 + 63) >> 6;
        int first = (gw - base) % GW; if (first < 0) first += GW;
        base = (base + nc) % GW;
        if (first >= nc) continue;
        // this wavefront has chunks of row tile i0: the multipliers u_pi / d_p of its 16 rows
        for (int idx = lane; idx < kPB * kPB; idx += 64) {
          const int pp = idx >> 4, ii = idx & 15;
          wt[idx] = (i0 + ii < jend) ? S[(size_t)(k0 + pp) * n + i0 + ii] * ddi[pp] : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
        for (int jc = first; jc < nc; jc += GW) {
          const int j = i0 + (jc << 6) + lane;
          if (j >= jend) continue;
          double uj[kPB];
#pragma unroll
          for (int p = 0; p < kPB; ++p) uj[p] = S[(size_t)(k0 + p) * n + j];
#pragma unroll
          for (int ii = 0; ii < kPB; ++ii) {
            const int i = i0 + ii;
            if (i >= jend || j < i) continue;
            double acc = 0.0;
#pragma unroll
            for (int p = 0; p < kPB; ++p) acc += wt[p * kPB + ii] * uj[p];
            S[(size_t)i * n + j] -= acc;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    OSH_GT(3);
    if (!grp_sync(g, lds_flag)) return false;
    OSH_GT(4);
  }
#ifdef OSH_LIBA_LDLT_TRACE
  if (m == 0 && tid == 0) printf("group ldlt n=%d bw=%d: diag %lld  row_panel %lld  barrier1 %lld  trailing %lld  barrier2 %lld cycles\n", n, d.bw, gt_t[0], gt_t[1], gt_t[2], gt_t[3], gt_t[4]);
#endif
  // ---- substitutions by block 0 (the other blocks wait at the caller's barrier)
  if (m == 0) {
    for (int k = tid; k < n; k += kLT) xs[k] = bs[k];
    __syncthreads();
    // (the forward substitution z_j = b_j - sum_{p<j} (u_pj / d_p) z_p happened inside the factorisation: bs holds z)
    // backward: x_p = z_p / d_p - (sum_{j>p} u_pj x_j) / d_p, the last rows first
    for (int k0 = ((n - 1) / kPB) * kPB; k0 >= 0; k0 -= kPB) {
      const int kb = min(kPB, n - k0);
      double part[kPB];
#pragma unroll
      for (int p = 0; p < kPB; ++p) part[p] = 0.0;
      const int jend = d.il ? min(n, k0 + kb + d.bw) : n;
      for (int j = k0 + kb + tid; j < jend; j += kLT) {
        const double xj = xs[j];
#pragma unroll
        for (int p = 0; p < kPB; ++p) if (p < kb) part[p] += S[(size_t)(k0 + p) * n + j] * xj;
      }
#pragma unroll
      for (int p = 0; p < kPB; ++p) { const double t = dev::wave_sum_dpp(part[p]); if (lane == 0) red[wave * kPB + p] = t; }
      for (int idx = tid; idx < kPB * kPB; idx += kLT) {
        const int pp = idx >> 4, jj = idx & 15;
        du[idx] = (pp < kb && jj < kb && jj >= pp) ? S[(size_t)(k0 + pp) * n + k0 + jj] : (pp == jj ? 1.0 : 0.0);
      }
      __syncthreads();
      if (tid < kPB) {   // lane p holds row p of the block: x_j of the higher rows arrives by shuffle, the last row first
        double sum = 0.0, r[kPB];
        for (int w = 0; w < kLT / 64; ++w) sum += red[w * kPB + tid];
#pragma unroll
        for (int j = 0; j < kPB; ++j) r[j] = du[tid * kPB + j];
        const double z = tid < kb ? xs[k0 + tid] : 0.0, dinvp = 1.0 / du[tid * kPB + tid];
        double x = 0.0;
#pragma unroll
        for (int j = kPB - 1; j >= 0; --j) {
          if (tid == j) x = (z - sum) * dinvp;
          const double xj = ldlt_readlane(x, j);
          if (tid < j) sum += r[j] * xj;
        }
        if (tid < kb) xs[k0 + tid] = x;
      }
      __syncthreads();
    }
    for (int k = tid; k < n; k += kLT) xg[k] = xs[k];
    if (tid == 0) ctrl[1] = ok ? 1.0 : 0.0;
#ifdef OSH_LIBA_LDLT_TRACE
    OSH_GT(5);
    if (tid == 0) printf("group ldlt n=%d: back substitution %lld cycles\n", n, gt_t[5]);
#endif
  }
  return true;
}

// ---- landmark back-substitution and the update of every vertex into the trial buffers; returns this thread's share of computeScale
__device__ __noinline__ double liba_backsub(const LibaCtx& c, int sel, double lambda, bool ok2) {
  OSH_LIBA_LOCALS
  const double* poses = v.pose[sel] + (size_t)d.pose_off * 24;
  const double* vba = v.vba[sel] + (size_t)d.vel_off * 9;
  const double* pts = v.pts[sel] + (size_t)d.pt_off * 3;
  const int trs = sel ^ 1;
      // landmark back-substitution (team per landmark), point update, landmark part of computeScale
      double sc = 0.0;
      double* pts_t = v.pts[trs] + (size_t)d.pt_off * 3;
      {
        const int T = (4 * L <= GT) ? 4 : ((2 * L <= GT) ? 2 : 1);
        const int q = tid & (T - 1);
        for (int j = gt / T; j < L; j += GT / T) {
          double c0 = 0, c1 = 0, c2 = 0;
          if (d.il) {
            // map-sized problem: walk the landmark's own edges (a handful) instead of every keyframe's slot in lm_pose_edge
            for (int x = lmo[j] + q; x < lmo[j + 1]; x += T) {
              const int ip = v.e_pose[(size_t)d.edge_off + x];
              if (ip >= N) continue;
              const int pos = lmpe[(size_t)j * N + ip];
              if (pos < 0 || v.pel_edge[(size_t)d.edge_off + pos] != x) continue;   // the second edge of a pair shares the first one's block
              const double* xp = xg + off_pose(d, ip);
#pragma unroll
              for (int r = 0; r < 6; ++r) { const double mx = -xp[r]; c0 += Hpl[(r * 3) * PS + pos] * mx; c1 += Hpl[(r * 3 + 1) * PS + pos] * mx; c2 += Hpl[(r * 3 + 2) * PS + pos] * mx; }
            }
          } else
          for (int i0 = q; i0 < N; i0 += 2 * T) {   // two blocks in flight
            int eu[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) eu[u] = (i0 + u * T < N) ? lmpe[(size_t)j * N + i0 + u * T] : -1;
            __builtin_amdgcn_sched_barrier(0);
            double Bu[2][18], xu[2][6];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const double* B = Hpl + (eu[u] < 0 ? 0 : eu[u]);
              const double* xp = xg + 6 * min(i0 + u * T, N - 1);
#pragma unroll
              for (int k = 0; k < 18; ++k) Bu[u][k] = B[k * PS];
#pragma unroll
              for (int k = 0; k < 6; ++k) xu[u][k] = xp[k];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              if (eu[u] < 0) continue;
#pragma unroll
              for (int r = 0; r < 6; ++r) { const double mx = -xu[u][r]; c0 += Bu[u][r * 3] * mx; c1 += Bu[u][r * 3 + 1] * mx; c2 += Bu[u][r * 3 + 2] * mx; }
            }
          }
          c0 = team_sum(c0, T); c1 = team_sum(c1, T); c2 = team_sum(c2, T);
          if (q != 0) continue;
          const double* Dj = dinv + (size_t)j * 9;
          const double b0 = bl[(size_t)j * 3], b1 = bl[(size_t)j * 3 + 1], b2 = bl[(size_t)j * 3 + 2];
          c0 += b0; c1 += b1; c2 += b2;
          double xl[3] = {0, 0, 0};
          if (ok2) {
            xl[0] = Dj[0] * c0 + Dj[1] * c1 + Dj[2] * c2;
            xl[1] = Dj[1] * c0 + Dj[3] * c1 + Dj[4] * c2;
            xl[2] = Dj[2] * c0 + Dj[4] * c1 + Dj[5] * c2;
          }
          pts_t[3 * (size_t)j] = pts[3 * (size_t)j] + xl[0]; pts_t[3 * (size_t)j + 1] = pts[3 * (size_t)j + 1] + xl[1];
          pts_t[3 * (size_t)j + 2] = pts[3 * (size_t)j + 2] + xl[2];
          sc += xl[0] * (lambda * xl[0] + b0) + xl[1] * (lambda * xl[1] + b1) + xl[2] * (lambda * xl[2] + b2);
        }
      }
      // pose / velocity / bias update into the trial buffers (ImuCamPose::Update, src/G2oTypes.cc:187-220): last block of the group
      double* poses_t = v.pose[trs] + (size_t)d.pose_off * 24;
      double* vba_t = v.vba[trs] + (size_t)d.vel_off * 9;
      if (m == G - 1) {
        for (int k = kLT - 1 - tid; k < N; k += kLT) {
          const double* pu = xg + off_pose(d, k);
          const double* P = poses + 24 * (size_t)k;
          double* Q = poses_t + 24 * (size_t)k;
          double tw[3], Ex[9], Rwb[9], Rbw[9], tbw[3], tc[3];
          imu::m3_vec(P + 12, pu + 3, tw);
          for (int i = 0; i < 3; ++i) Q[21 + i] = P[21 + i] + tw[i];
          imu::exp_so3(pu, Ex);
          imu::m3_mul(P + 12, Ex, Rwb);
          for (int i = 0; i < 9; ++i) Q[12 + i] = Rwb[i];
          for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbw[i * 3 + j] = Rwb[j * 3 + i];
          imu::m3_vec(Rbw, Q + 21, tbw);
          tbw[0] = -tbw[0]; tbw[1] = -tbw[1]; tbw[2] = -tbw[2];
          imu::m3_mul(d.Rcb, Rbw, Q);
          imu::m3_vec(d.Rcb, tbw, tc);
          for (int i = 0; i < 3; ++i) Q[9 + i] = tc[i] + d.tcb[i];
          for (int i = 0; i < 9; ++i) vba_t[9 * k + i] = vba[9 * k + i] + xg[off_vba(d, k) + i];
        }
        for (int k = tid; k < n; k += kLT) sc += xg[k] * (lambda * xg[k] + bfull[k]);
      }
  return sc;
}

// ---- per-edge chi2 and depth flags, final estimates into the result arena
__device__ __noinline__ void liba_outputs(const LibaCtx& c, int sel, int eval_sel) {
  OSH_LIBA_LOCALS
  // e->chi2() of the errors computeActiveErrors saw last (buffer eval_sel: stale after a rejected final trial) and
  // isDepthPositive() of the final estimates (src/Optimizer.cc:2861-2888; ImuCamPose::isDepthPositive, G2oTypes.cc:185-188)
  {
    const double* pe = v.pose[eval_sel] + (size_t)d.pose_off * 24;
    const double* xe = v.pts[eval_sel] + (size_t)d.pt_off * 3;
    const double* pf = v.pose[sel] + (size_t)d.pose_off * 24;
    const double* xf = v.pts[sel] + (size_t)d.pt_off * 3;
    for (int e = gt; e < d.E; e += GT) {
      const size_t ge = (size_t)d.edge_off + e;
      const int ip = v.e_pose[ge], il = v.e_point[ge];
      VisEval ev;
      vis_residual(d, v.e_kind[ge], pe + 24 * (size_t)ip, xe + 3 * (size_t)il, v.e_obs + ge * 3, v.e_info[ge], ev);
      const double* R = pf + 24 * (size_t)ip; const double* X = xf + 3 * (size_t)il;
      const size_t go = (size_t)d.edge_off + v.e_orig[ge];
      v.out_chi2[go] = ev.chi2;
      if (v.e_kind[ge] == OSH_EDGE_RIGHT) {   // isDepthPositive(Xw, 1): row 2 of Rcw[1] = Rrl Rcw[0], tcw[1] = Rrl tcw[0] + trl
        double r2[3], t2 = d.trl[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) r2[c] = d.Rrl[6] * R[c] + d.Rrl[7] * R[3 + c] + d.Rrl[8] * R[6 + c];
        t2 += d.Rrl[6] * R[9] + d.Rrl[7] * R[10] + d.Rrl[8] * R[11];
        v.out_depth[go] = (r2[0] * X[0] + r2[1] * X[1] + r2[2] * X[2] + t2) > 0.0 ? 1 : 0;
      } else {
        v.out_depth[go] = (R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + R[11]) > 0.0 ? 1 : 0;
      }
    }
  }
  // final estimates of the optimisable keyframes and of the landmarks into the result arena
  {
    const size_t kf0 = (size_t)(d.b_off / 15);
    const double* pf = v.pose[sel] + (size_t)d.pose_off * 24;
    const double* sf = v.vba[sel] + (size_t)d.vel_off * 9;
    const double* xf = v.pts[sel] + (size_t)d.pt_off * 3;
    for (int k = gt; k < N * 24; k += GT) v.res_pose[kf0 * 24 + k] = pf[k];
    for (int k = gt; k < N * 9; k += GT) v.res_vba[kf0 * 9 + k] = sf[k];
    for (int k = gt; k < L * 3; k += GT) v.res_pts[(size_t)d.pt_off * 3 + k] = xf[k];
  }
}

template <int NB>
__global__ __launch_bounds__(kLT) void k_liba(LibaView v, int W, int G) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  // consecutive workgroups go to the 8 XCDs in turn: the G blocks of a window are 8 apart, so they share one XCD and its L2
  const int bid = blockIdx.x;
  const int win = (bid / (8 * G)) * 8 + (bid & 7);
  if (win >= v.nw) return;
  const LibaDesc& d = v.desc[win];
  LibaOut& out = v.out[win];
  const int tid = threadIdx.x;
  Grp g;
  g.bar = v.bar + win; g.abort_flag = v.abort_flag; g.res_abort = v.res_abort; g.red = v.red + (size_t)win * 4 * kLG * 2; g.G = G; g.m = (bid >> 3) % G;
  g.gen = 0; g.nred = 0; g.light = 0; g.test_abort = v.test_abort;
  const int m = g.m, GT = G * kLT, gt = m * kLT + tid, GW = G * (kLT / 64);
  const int N = d.N, n = d.n, L = d.L, n6 = 6 * d.N;
  // LDS carve: [0, ldlt) the LDL^T scratch (reused as general scratch between solves), then control words
  double* shw = sh + liba_scratch_doubles(NB, W);             // [kLT/64] reductions
  int* lds_flag = reinterpret_cast<int*>(shw + kLT / 64 + 1);
  const double* H = v.H + d.H_off;
  const double* Hll = v.Hll + (size_t)d.pt_off * 6;
  const double* ppart = v.ppart + (size_t)(d.b_off / 15) * kPoseChunks * 27;
  double* ctrl = v.ctrl + (size_t)win * 4;
  const bool prof_on = (m == 0 && tid == 0);
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long prof_last = clock64();
  if (bid == 0 && tid == 0) *v.res_abort = 0;
  // the trial buffers start as copies of the estimates (the fixed keyframes and the fixed IMU state are only ever read)
  for (int k = gt; k < d.K * 24; k += GT) v.pose[1][(size_t)d.pose_off * 24 + k] = v.pose[0][(size_t)d.pose_off * 24 + k];
  for (int k = gt; k < d.NV * 9; k += GT) v.vba[1][(size_t)d.vel_off * 9 + k] = v.vba[0][(size_t)d.vel_off * 9 + k];

  int C = (GW - GW / 4) / (N > 0 ? N : 1);     // pose rows are summed in C chunks each so that a few keyframes still spread over the group
  C = C < 1 ? 1 : (C > kPoseChunks ? kPoseChunks : C);
  LibaCtx c;
  c.v = &v; c.d = &d; c.sh = sh; c.shw = shw; c.G = G; c.m = m; c.C = C; c.win = win; c.W = W;

  if (G > 1) {
    // which XCD is this block on?  (the first barrier is the full agent-scope one)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (tid == 0) g.red[3 * kLG * 2 + m] = (double)(xcc & 0xfu);
    OSH_GSYNC();
    bool same = true;
    for (int k = 1; k < G; ++k) same = same && g.red[3 * kLG * 2 + k] == g.red[3 * kLG * 2];
    g.light = (same && !v.force_heavy) ? 1 : 0;
    __syncthreads();
  }
  int sel = 0, eval_sel = 0;
  double lambda = -1.0, ni = 2.0;
  int nBad = 0, cj = 0, trials_total = 0, n_trace = 0;
  bool ok = true;
  double chi_init = eval_partial(v, d, 0, g), zero = 0.0;
  if (!grp_sum2(g, chi_init, zero, shw, lds_flag)) return;
  if (m == 0 && tid == 0) out.chi2_initial = chi_init;
  double last_chi = chi_init;      // activeRobustChi2() of the errors evaluated last (err_end)
  double currentChi = chi_init;    // of buffer `sel`: the accepted trial's value (the same sum over the same buffer the reference recomputes)
  OSH_PROF(6);

  for (int it = 0; it < d.max_iter && ok; ++it) {
    const double iniChi = currentChi;
    // ---------------------------------------------------------------- linearise (buildSystem)
    liba_linearise(c, sel);
    OSH_GSYNC();
    OSH_PROF(0);
    // ---- the links of the first colour into H; landmark and pose rows summed; (lambda known) Dinv, B Dinv
    const bool lambda_known = it > 0 || d.lambda_init > 0;
    if (it == 0 && lambda_known) lambda = d.lambda_init;
    liba_assemble_links(c, sel, 0);
    liba_landmark_pass(c, true, lambda_known, lambda, 0, G);
    liba_pose_pass(c, 0, G);
    OSH_GSYNC();
    for (int col = 1; col < d.n_colours; ++col) {
      liba_assemble_links(c, sel, col);
      OSH_GSYNC();
    }
    if (!lambda_known) {
      // setLambda's default: 1e-5 x the largest diagonal entry of the full Hessian (computeLambdaInit)
      if (m == 0) {
        double mx = 0.0;
        for (int k = tid; k < n; k += kLT) {
          double hd = H[(size_t)k * n + k];
          const bool is_pose = d.il ? (k % 15) < 6 : k < n6;
          if (is_pose) { const int i = d.il ? k / 15 : k / 6, r = d.il ? k % 15 : k - i * 6; for (int ch = 0; ch < C; ++ch) hd += ppart[((size_t)i * kPoseChunks + ch) * 27 + up21(r, r)]; }
          mx = fmax(mx, fabs(hd));
        }
        for (int j = tid; j < L; j += kLT) mx = fmax(mx, fmax(fabs(Hll[(size_t)j * 6]), fmax(fabs(Hll[(size_t)j * 6 + 3]), fabs(Hll[(size_t)j * 6 + 5]))));
        mx = blk_max(mx, shw);
        if (tid == 0) ctrl[0] = 1e-5 * mx;
      }
      OSH_GSYNC();
      lambda = ctrl[0];
      liba_landmark_pass(c, false, true, lambda, 0, G);
      OSH_GSYNC();
    }
    if (it == 0) { ni = 2.0; nBad = 0; }
    OSH_PROF(1);
    // ---------------------------------------------------------------- LM trials
    double rho = 0.0;
    int qmax = 0;
    do {
      const int trs = sel ^ 1;
      if (qmax > 0) {   // a rejected trial changed lambda: Dinv, B Dinv again
        liba_landmark_pass(c, false, true, lambda, 0, G);
        OSH_GSYNC();
      }
      OSH_PROF(2);
      liba_schur(c, lambda);
      OSH_GSYNC();
      OSH_PROF(3);
      if constexpr (NB == kLNB) { if (m == 0) liba_solve<NB>(c, sh); }
      else { if (!liba_solve_group(c, g, lds_flag)) return; }
      OSH_GSYNC();
      const bool ok2 = ctrl[1] != 0.0;
      OSH_PROF(4);
      double sc = liba_backsub(c, sel, lambda, ok2);
      OSH_GSYNC();
      OSH_PROF(5);
      double tempChi = eval_partial(v, d, trs, g);
      if (!grp_sum2(g, tempChi, sc, shw, lds_flag)) return;
      const double scale_sum = sc;
      last_chi = tempChi;
      eval_sel = trs;
      if (!ok2) tempChi = DBL_MAX;
      // controller: identical decisions in every thread of every block (all inputs are group-uniform)
      rho = (currentChi - tempChi);
      const double scale = scale_sum + 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        lambda *= fmax(1. / 3., alpha);
        ni = 2; currentChi = tempChi;
        sel = trs;   // discardTop
      } else {
        lambda *= ni; ni *= 2;   // pop
      }
      qmax++; trials_total++;
      OSH_PROF(6);
    } while (rho < 0 && qmax < 10);
    ++cj;
    if (m == 0 && tid == 0 && n_trace < OSH_LBA_MAX_TRACE) { out.chi2_trace[n_trace] = currentChi; out.lambda_trace[n_trace] = lambda; out.trials_trace[n_trace] = qmax; }
    if (n_trace < OSH_LBA_MAX_TRACE) ++n_trace;
    if (qmax == 10 || rho == 0) { ok = false; continue; }
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) { ok = false; continue; }
  }
  if (m == 0 && tid == 0) { out.iterations = cj; out.trials = trials_total; out.n_trace = n_trace; out.sel = sel; out.chi2_final = last_chi; }
  liba_outputs(c, sel, eval_sel);
  OSH_PROF(7);
  if (prof_on) for (int k = 0; k < 8; ++k) out.prof[k] = prof[k];
}
#undef OSH_GSYNC
#undef OSH_PROF

}  // namespace osh

// =============================================================================================
// Host driver: osh_liba_solve (upload + one launch + download)
// =============================================================================================
using namespace osh;

namespace {
struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  ~PinnedBuf() { if (p) (void)hipHostFree(p); }
  void* reserve(size_t bytes) {
    if (bytes <= cap) return p;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&p, want) != hipSuccess) { p = nullptr; return nullptr; }
    cap = want;
    return p;
  }
};
struct LibaBuffers {
  PinnedBuf h_in, h_out;
  DevBuf arena;
};

thread_local int g_liba_last_group = 0;
thread_local long long g_liba_last_prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
}  // namespace

#define OSH_TRY(expr) do { int _rc = (expr); if (_rc != OSH_OK) return _rc; } while (0)

extern "C" int osh_lba_stream(osh_lba_ctx* ctx, int* device, hipStream_t* stream);   // lba_device.hip
extern "C" void** osh_lba_attachment(osh_lba_ctx* ctx, int slot, void (*free_fn)(void*));   // lba_device.hip

extern "C" int osh_liba_solve(osh_lba_ctx* ctx, int32_t nw, const osh_liba_problem* pr, osh_liba_result* res) {
  if (!ctx || nw <= 0 || !pr || !res) { set_error("osh_liba_solve: bad arguments"); return OSH_ERR_INVALID; }
  int device = 0;
  hipStream_t s = nullptr;
  OSH_TRY(osh_lba_stream(ctx, &device, &s));
  OSH_HIP(hipSetDevice(device));
  std::vector<LibaDesc> h_desc(nw);
  size_t K = 0, NV = 0, L = 0, E = 0, NL = 0, Htot = 0, btot = 0, LO = 0, PO = 0, EF = 0, LP = 0;
  int n_max = 0;
  for (int w = 0; w < nw; ++w) {
    const osh_liba_problem& p = pr[w];
    if (p.n_opt <= 0 || p.n_fixed_imu < 0 || p.n_fixed_imu > 1 || p.n_fixed < 0 || p.n_points < 0 || p.n_edges < 0 || p.n_links < 0 ||
        p.max_iterations > OSH_LBA_MAX_TRACE) { set_error("window %d: bad sizes", w); return OSH_ERR_INVALID; }
    LibaDesc& d = h_desc[w];
    d.N = p.n_opt; d.NV = p.n_opt + p.n_fixed_imu; d.K = d.NV + p.n_fixed; d.L = p.n_points; d.E = p.n_edges; d.NL = p.n_links;
    d.n = 15 * d.N; d.max_iter = p.max_iterations; d.n_colours = 0; d.il = 0; d.bw = d.n; d.bw_kf = d.N;
    d.pose_off = (int)K; d.vel_off = (int)NV; d.pt_off = (int)L; d.edge_off = (int)E; d.link_off = (int)NL; d.lmoff_off = (int)LO;
    d.peloff_off = (int)PO; d.pel_off = (int)EF; d.lmpose_off = (int)LP; d.H_off = (long long)Htot; d.b_off = (int)btot;
    std::memcpy(d.Rcb, p.Rcb, 72); std::memcpy(d.tcb, p.tcb, 24); std::memcpy(d.tbc, p.tbc, 24); std::memcpy(d.cam, p.cam, 40);
    d.huber_mono = p.huber_mono; d.huber_stereo = p.huber_stereo; d.huber_inertial = p.huber_inertial; d.lambda_init = p.lambda_init;
    d.kb8_on = p.kb8 ? 1 : 0;
    for (int k = 0; k < 4; ++k) d.kb8[k] = p.kb8 ? p.kb8[k] : 0.0;
    d.rig_on = (p.kb8 && p.cam2 && p.trl) ? 1 : 0;
    if (d.rig_on) {
      // ImuCamPose(KeyFrame*) camera 1 (src/G2oTypes.cc:56-66): Rcb[1] = Rrl Rcb[0], tcb[1] = Rrl tcb[0] + trl, tbc[1] = -Rbc[1] tcb[1]
      double tcb1[3];
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) d.Rrl[i * 3 + j] = p.trl[i * 4 + j]; d.trl[i] = p.trl[i * 4 + 3]; }
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          double a = 0.0;
          for (int k = 0; k < 3; ++k) a += d.Rrl[i * 3 + k] * d.Rcb[k * 3 + j];
          d.Rcb1[i * 3 + j] = a;
        }
      for (int i = 0; i < 3; ++i) tcb1[i] = d.Rrl[i * 3] * d.tcb[0] + d.Rrl[i * 3 + 1] * d.tcb[1] + d.Rrl[i * 3 + 2] * d.tcb[2] + d.trl[i];
      for (int i = 0; i < 3; ++i) d.tbc1[i] = -(d.Rcb1[i] * tcb1[0] + d.Rcb1[3 + i] * tcb1[1] + d.Rcb1[6 + i] * tcb1[2]);
      std::memcpy(d.cam2, p.cam2, 64);
    }
    if (p.kb8)
      for (int e = 0; e < p.n_edges; ++e)
        if (p.edge_kind[e] == OSH_EDGE_STEREO) { set_error("window %d: a KannalaBrandt8 window takes monocular edges only (edge %d)", w, e); return OSH_ERR_UNSUPPORTED; }
    size_t ef = 0;
    for (int e = 0; e < p.n_edges; ++e) {
      if (p.edge_pose[e] < 0 || p.edge_pose[e] >= d.K || p.edge_point[e] < 0 || p.edge_point[e] >= d.L || p.edge_kind[e] > OSH_EDGE_RIGHT) {
        set_error("window %d edge %d: index or kind out of range", w, e); return OSH_ERR_INVALID;
      }
      if (p.edge_kind[e] == OSH_EDGE_RIGHT && !d.rig_on) { set_error("window %d edge %d: a right-camera edge (EdgeMono(1)) needs kb8, cam2 and trl", w, e); return OSH_ERR_INVALID; }
      if (p.edge_pose[e] < d.N) ++ef;
    }
    for (int l = 0; l < p.n_links; ++l)
      if (p.link_prev[l] < 0 || p.link_prev[l] >= d.NV || p.link_cur[l] < 0 || p.link_cur[l] >= d.N) {
        set_error("window %d link %d: keyframe index out of range", w, l); return OSH_ERR_INVALID;
      }
    if (p.link_bias)
      for (int l = 0; l < p.n_links; ++l) {
        if (p.link_bias[l] < 0 || p.link_bias[l] >= d.NV) { set_error("window %d link %d: bias keyframe out of range", w, l); return OSH_ERR_INVALID; }
        if (p.link_bias[l] == p.link_prev[l]) continue;
        // the random-walk terms of the later keyframe are added beside the edge's own terms, by other threads of the same phase
        if (p.link_bias[l] == p.link_cur[l]) { set_error("window %d link %d: the bias vertices of a link cannot be those of its later keyframe", w, l); return OSH_ERR_UNSUPPORTED; }
        // the random-walk terms of a link's earlier keyframe are summed into the blocks of the edge's own bias vertices
        for (int k = 0; k < 9; ++k)
          if (p.link_info_g[(size_t)l * 9 + k] != 0.0 || p.link_info_a[(size_t)l * 9 + k] != 0.0) {
            set_error("window %d link %d: a link whose bias vertices belong to another keyframe carries no random-walk edges", w, l); return OSH_ERR_UNSUPPORTED;
          }
      }
    K += d.K; NV += d.NV; L += d.L; E += d.E; NL += d.NL; Htot += (size_t)d.n * d.n; btot += d.n; LO += (size_t)d.L + 1; PO += (size_t)d.N + 1;
    EF += ef; LP += (size_t)d.L * d.N;
    n_max = std::max(n_max, d.n);
  }
  const int W = ldlt_row_stride(n_max);
  // 24-wide panels in the LDS of one block while they fit (51 keyframes); beyond, the whole group factorises in global memory
  // (liba_solve_group; k_liba<6> -- its LDS need, vectors of n doubles, stays below what 6-wide panels would take)
  int NB = kLNB;
  if ((liba_scratch_doubles(NB, W) + kLT / 64 + 8) * sizeof(double) > 160 * 1024 - 64) NB = 6;
  const size_t lds = (liba_scratch_doubles(NB, W) + kLT / 64 + 8) * sizeof(double);
  if (lds > 160 * 1024 - 64 || n_max > 15 * kLibaMaxKeyframes) {
    set_error("inertial window with %d optimisable keyframes: the device path handles up to %d (LocalInertialBA uses 10 or 25)", n_max / 15, kLibaMaxKeyframes);
    return OSH_ERR_UNSUPPORTED;
  }
  // Map-sized problems (the group factorisation in global memory): with the keyframes in temporal order a landmark is seen by nearby
  // keyframes and an IMU link joins neighbours, so with the unknowns interleaved per keyframe the reduced system is banded.  The band is
  // the largest keyframe distance any landmark or link spans; a loop closure or the one-bias-pair mode of FullInertialBA (every link on
  // one keyframe's bias vertices) makes it the whole map, and the problem stays in the dense layout.
  if (NB != kLNB && !getenv("OSH_LIBA_DENSE")) {
    std::vector<int> lo, hi;
    for (int w = 0; w < nw; ++w) {
      const osh_liba_problem& p = pr[w];
      LibaDesc& d = h_desc[w];
      if (d.N < 32) continue;
      int span = 1;
      lo.assign(d.L, d.N); hi.assign(d.L, -1);
      for (int e = 0; e < p.n_edges; ++e) {
        const int ip = p.edge_pose[e], j = p.edge_point[e];
        if (ip >= d.N) continue;
        lo[j] = std::min(lo[j], ip); hi[j] = std::max(hi[j], ip);
      }
      for (int j = 0; j < d.L; ++j) if (hi[j] >= 0) span = std::max(span, hi[j] - lo[j]);
      for (int l = 0; l < p.n_links; ++l) {
        const int a = p.link_prev[l], c2 = p.link_cur[l], ab = p.link_bias ? p.link_bias[l] : a;
        int mn = c2, mx = c2;
        if (a < d.N) { mn = std::min(mn, a); mx = std::max(mx, a); }
        if (ab < d.N) { mn = std::min(mn, ab); mx = std::max(mx, ab); }
        span = std::max(span, mx - mn);
      }
      if (15 * (span + 1) <= d.n / 2) { d.il = 1; d.bw_kf = span; d.bw = 15 * (span + 1) - 1; }
    }
  }
  // ---- pack: every input array goes into ONE pinned staging buffer and travels in ONE copy (an upload per array cost more than
  // the optimisation of a single window); the device pointers are offsets into the arena.
  // staging and work buffers live with the context (one solver at a time per context, as for the visual path)
  void** slot = osh_lba_attachment(ctx, 0, [](void* q) { delete static_cast<LibaBuffers*>(q); });
  if (!slot) { set_error("osh_liba_solve: no context"); return OSH_ERR_INVALID; }
  if (!*slot) *slot = new LibaBuffers();
  LibaBuffers& B = *static_cast<LibaBuffers*>(*slot);
  size_t in_bytes = 0;
  auto take = [&](size_t bytes) { const size_t o = in_bytes; in_bytes = (in_bytes + bytes + 255) & ~(size_t)255; return o; };
  const size_t o_desc = take(nw * sizeof(LibaDesc)), o_pose = take(K * 24 * 8), o_vba = take(NV * 9 * 8), o_pts = take(L * 3 * 8), o_obs = take(E * 3 * 8),
               o_info = take(E * 8), o_ep = take(E * 4), o_el = take(E * 4), o_eo = take(E * 4), o_lmo = take(LO * 4), o_po = take(PO * 4),
               o_pel = take(E * 4), o_lmpe = take(LP * 4), o_lp = take(NL * 4), o_lc = take(NL * 4), o_kind = take(E), o_rob = take(NL),
               o_pre = take(NL * OSH_PREINT_FLOATS * 4), o_li = take(NL * 81 * 8), o_lg = take(NL * 9 * 8), o_la = take(NL * 9 * 8),
               o_bar = take(nw * sizeof(unsigned)), o_abort = take(sizeof(int)), o_col = take(NL * 4), o_lb = take(NL * 4);
  char* hs = static_cast<char*>(B.h_in.reserve(in_bytes));
  if (!hs) { set_error("osh_liba_solve: pinned staging allocation of %zu bytes failed", in_bytes); return OSH_ERR_DEVICE; }
  LibaDesc* h_descp = reinterpret_cast<LibaDesc*>(hs + o_desc);
  double* h_pose = reinterpret_cast<double*>(hs + o_pose); double* h_vba = reinterpret_cast<double*>(hs + o_vba);
  double* h_pts = reinterpret_cast<double*>(hs + o_pts); double* h_obs = reinterpret_cast<double*>(hs + o_obs);
  double* h_info = reinterpret_cast<double*>(hs + o_info);
  int* h_ep = reinterpret_cast<int*>(hs + o_ep); int* h_el = reinterpret_cast<int*>(hs + o_el); int* h_eo = reinterpret_cast<int*>(hs + o_eo);
  int* h_lmo = reinterpret_cast<int*>(hs + o_lmo); int* h_po = reinterpret_cast<int*>(hs + o_po); int* h_pel = reinterpret_cast<int*>(hs + o_pel);
  int* h_lmpe = reinterpret_cast<int*>(hs + o_lmpe); int* h_lp = reinterpret_cast<int*>(hs + o_lp); int* h_lc = reinterpret_cast<int*>(hs + o_lc); int* h_lb = reinterpret_cast<int*>(hs + o_lb);
  unsigned char* h_kind = reinterpret_cast<unsigned char*>(hs + o_kind); unsigned char* h_rob = reinterpret_cast<unsigned char*>(hs + o_rob);
  float* h_pre = reinterpret_cast<float*>(hs + o_pre);
  int* h_col = reinterpret_cast<int*>(hs + o_col);
  double* h_li = reinterpret_cast<double*>(hs + o_li); double* h_lg = reinterpret_cast<double*>(hs + o_lg); double* h_la = reinterpret_cast<double*>(hs + o_la);
  std::memset(h_lmpe, 0xff, LP * 4);
  std::memset(hs + o_bar, 0, nw * sizeof(unsigned));
  std::memset(hs + o_abort, 0, sizeof(int));
  std::vector<int> cnt, fill, order;
  for (int w = 0; w < nw; ++w) {
    const osh_liba_problem& p = pr[w];
    const LibaDesc& d = h_desc[w];
    for (int k = 0; k < d.K; ++k) {
      double* o = &h_pose[((size_t)d.pose_off + k) * 24];
      std::memcpy(o, p.pose_Rcw + 9 * k, 72); std::memcpy(o + 9, p.pose_tcw + 3 * k, 24);
      std::memcpy(o + 12, p.pose_Rwb + 9 * k, 72); std::memcpy(o + 21, p.pose_twb + 3 * k, 24);
    }
    for (int k = 0; k < d.NV; ++k) {
      double* o = &h_vba[((size_t)d.vel_off + k) * 9];
      std::memcpy(o, p.vel + 3 * k, 24); std::memcpy(o + 3, p.bias_g + 3 * k, 24); std::memcpy(o + 6, p.bias_a + 3 * k, 24);
    }
    if (d.L) std::memcpy(&h_pts[(size_t)d.pt_off * 3], p.points, (size_t)d.L * 24);
    cnt.assign((size_t)d.L + 1, 0);
    for (int e = 0; e < d.E; ++e) cnt[p.edge_point[e] + 1]++;
    for (int j = 0; j < d.L; ++j) cnt[j + 1] += cnt[j];
    fill.assign(cnt.begin(), cnt.end() - 1);
    order.resize(d.E);
    for (int e = 0; e < d.E; ++e) order[fill[p.edge_point[e]]++] = e;
    for (int j = 0; j <= d.L; ++j) h_lmo[d.lmoff_off + j] = cnt[j];
    for (int j = 0; j < d.L; ++j) {
      std::stable_sort(order.begin() + cnt[j], order.begin() + cnt[j + 1], [&](int a, int b) {
        return p.edge_pose[a] != p.edge_pose[b] ? p.edge_pose[a] < p.edge_pose[b] : p.edge_kind[a] < p.edge_kind[b];
      });
      for (int x = cnt[j]; x < cnt[j + 1]; ++x) {
        if (x > cnt[j] && p.edge_pose[order[x]] == p.edge_pose[order[x - 1]]) {
          // one Hessian block, two edges: only the left EdgeMono(0) + right EdgeMono(1) of a fisheye rig (src/Optimizer.cc:2737-2835)
          const bool pair = p.edge_kind[order[x]] == OSH_EDGE_RIGHT && p.edge_kind[order[x - 1]] == OSH_EDGE_MONO &&
                            !(x - 1 > cnt[j] && p.edge_pose[order[x - 2]] == p.edge_pose[order[x]]);
          if (!pair) {
            set_error("window %d: landmark %d is observed twice by keyframe %d with edge kinds that do not form a left + right pair", w, j, p.edge_pose[order[x]]);
            return OSH_ERR_UNSUPPORTED;
          }
          continue;   // the pair's block is the first edge's
        }
        if (p.edge_pose[order[x]] < d.N) h_lmpe[(size_t)d.lmpose_off + (size_t)j * d.N + p.edge_pose[order[x]]] = x;
      }
    }
    int* po = &h_po[d.peloff_off];
    for (int i = 0; i <= d.N; ++i) po[i] = 0;
    for (int x = 0; x < d.E; ++x) {
      const int e = order[x];
      const size_t g = (size_t)d.edge_off + x;
      h_ep[g] = p.edge_pose[e]; h_el[g] = p.edge_point[e]; h_kind[g] = p.edge_kind[e]; h_eo[g] = e; h_info[g] = p.edge_info[e];
      for (int k = 0; k < 3; ++k) h_obs[g * 3 + k] = p.edge_obs[3 * e + k];
      if (p.edge_pose[e] < d.N) po[p.edge_pose[e] + 1]++;
    }
    for (int i = 0; i < d.N; ++i) po[i + 1] += po[i];
    fill.assign(po, po + d.N);
    int nfix = po[d.N];   // the fixed keyframes' edges follow the optimisable ones in the walk order of the linearisation
    cnt.assign((size_t)d.E, -1);   // place of each optimisable-pose edge in that order
    for (int x = 0; x < d.E; ++x) {
      const int ip = h_ep[(size_t)d.edge_off + x];
      if (ip < d.N) cnt[x] = fill[ip];
      h_pel[(size_t)d.edge_off + (ip < d.N ? fill[ip]++ : nfix++)] = x;
    }
    // (landmark, pose) -> place of the pair's block: Hpl and B Dinv are stored pose by pose, a landmark's neighbours next to it
    for (size_t k = 0; k < (size_t)d.L * d.N; ++k) { int& x = h_lmpe[(size_t)d.lmpose_off + k]; if (x >= 0) x = cnt[x]; }
    for (int l = 0; l < d.NL; ++l) {
      const size_t g = (size_t)d.link_off + l;
      h_lp[g] = p.link_prev[l]; h_lc[g] = p.link_cur[l]; h_rob[g] = p.link_robust[l];
      h_lb[g] = p.link_bias ? p.link_bias[l] : p.link_prev[l];
      std::memcpy(&h_pre[g * OSH_PREINT_FLOATS], p.link_preint + (size_t)l * OSH_PREINT_FLOATS, OSH_PREINT_FLOATS * 4);
      std::memcpy(&h_li[g * 81], p.link_info + (size_t)l * 81, 81 * 8);
      std::memcpy(&h_lg[g * 9], p.link_info_g + (size_t)l * 9, 72); std::memcpy(&h_la[g * 9], p.link_info_a + (size_t)l * 9, 72);
      // greedy colouring: the first colour none of the earlier links sharing a keyframe with this one has (a chain takes two)
      int col = 0;
      for (bool clash = true; clash; ) {
        clash = false;
        for (int l2 = 0; l2 < l && !clash; ++l2) {
          if (h_col[(size_t)d.link_off + l2] != col) continue;
          const int k1[3] = {p.link_prev[l], p.link_cur[l], h_lb[g]}, k2[3] = {p.link_prev[l2], p.link_cur[l2], h_lb[(size_t)d.link_off + l2]};
          for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) clash = clash || k1[x] == k2[y];
        }
        if (clash) ++col;
      }
      h_col[g] = col;
      h_desc[w].n_colours = std::max(h_desc[w].n_colours, col + 1);
    }
  }
  std::memcpy(h_descp, h_desc.data(), nw * sizeof(LibaDesc));
  // ---- result arena (one copy back): LibaOut per window, abort word, final poses / velocities+biases / points, edge chi2 and depth flags
  size_t out_bytes = 0;
  auto take_out = [&](size_t bytes) { const size_t o = out_bytes; out_bytes = (out_bytes + bytes + 255) & ~(size_t)255; return o; };
  const size_t r_out = take_out(nw * sizeof(LibaOut)), r_abort = take_out(sizeof(int)), r_pose = take_out((btot / 15) * 24 * 8), r_vba = take_out((btot / 15) * 9 * 8),
               r_pts = take_out(L * 3 * 8), r_chi2 = take_out(E * 8), r_depth = take_out(E);
  char* hr = static_cast<char*>(B.h_out.reserve(out_bytes));
  if (!hr) { set_error("osh_liba_solve: pinned result allocation of %zu bytes failed", out_bytes); return OSH_ERR_DEVICE; }
  auto R = [](DevBuf& b, size_t bytes) { return b.reserve(std::max<size_t>(bytes, (size_t)64 << 20)); };   // at least 64 MiB: large page fragments
  // ONE device arena: inputs (uploaded in one copy), results (downloaded in one copy), then the work buffers
  size_t dev_bytes = 0;
  auto take_dev = [&](size_t bytes) { const size_t o = dev_bytes; dev_bytes = (dev_bytes + std::max<size_t>(bytes, 8) + 255) & ~(size_t)255; return o; };
  const size_t a_in = take_dev(in_bytes), a_res = take_dev(out_bytes), a_pose1 = take_dev(K * 24 * 8), a_vba1 = take_dev(NV * 9 * 8), a_pts1 = take_dev(L * 3 * 8),
               a_eh = take_dev(E * 9 * 8), a_ep = take_dev(EF * 27 * 8), a_bfull = take_dev(btot * 8), a_Hpl = take_dev(EF * 18 * 8), a_BD = take_dev(EF * 18 * 8),
               a_Hll = take_dev(L * 6 * 8), a_bl = take_dev(L * 3 * 8), a_dinv = take_dev(L * 9 * 8), a_H = take_dev(Htot * 8), a_S = take_dev(Htot * 8),
               a_b = take_dev(btot * 8), a_bs = take_dev(btot * 8), a_x = take_dev(btot * 8), a_linkQ = take_dev(NL * kLinkQ * 8),
               a_ppart = take_dev((btot / 15) * kPoseChunks * 27 * 8), a_red = take_dev((size_t)nw * 4 * kLG * 2 * 8), a_ctrl = take_dev((size_t)nw * 4 * 8);
  OSH_TRY(R(B.arena, dev_bytes));
  char* const dbase = B.arena.as<char>();
  auto dptr = [&](size_t off) { return reinterpret_cast<double*>(dbase + off); };
  char* din = dbase + a_in;
  char* dres = dbase + a_res;
  OSH_HIP(hipMemcpyAsync(din, hs, in_bytes, hipMemcpyHostToDevice, s));
  LibaView v{};
  v.desc = reinterpret_cast<const LibaDesc*>(din + o_desc); v.out = reinterpret_cast<LibaOut*>(dres + r_out);
  v.pose[0] = reinterpret_cast<double*>(din + o_pose); v.vba[0] = reinterpret_cast<double*>(din + o_vba); v.pts[0] = reinterpret_cast<double*>(din + o_pts);
  v.pose[1] = dptr(a_pose1); v.vba[1] = dptr(a_vba1); v.pts[1] = dptr(a_pts1);
  v.e_pose = reinterpret_cast<const int*>(din + o_ep); v.e_point = reinterpret_cast<const int*>(din + o_el);
  v.e_kind = reinterpret_cast<const unsigned char*>(din + o_kind); v.e_obs = reinterpret_cast<const double*>(din + o_obs);
  v.e_info = reinterpret_cast<const double*>(din + o_info); v.e_orig = reinterpret_cast<const int*>(din + o_eo);
  v.lm_off = reinterpret_cast<const int*>(din + o_lmo); v.pel_off = reinterpret_cast<const int*>(din + o_po);
  v.pel_edge = reinterpret_cast<const int*>(din + o_pel); v.lm_pose_edge = reinterpret_cast<const int*>(din + o_lmpe);
  v.link_prev = reinterpret_cast<const int*>(din + o_lp); v.link_cur = reinterpret_cast<const int*>(din + o_lc); v.link_bias = reinterpret_cast<const int*>(din + o_lb);
  v.link_preint = reinterpret_cast<const float*>(din + o_pre); v.link_info = reinterpret_cast<const double*>(din + o_li);
  v.link_info_g = reinterpret_cast<const double*>(din + o_lg); v.link_info_a = reinterpret_cast<const double*>(din + o_la);
  v.link_robust = reinterpret_cast<const unsigned char*>(din + o_rob);
  v.Hpl = dptr(a_Hpl); v.BD = dptr(a_BD); v.Hll = dptr(a_Hll); v.bl = dptr(a_bl);
  v.dinv = dptr(a_dinv); v.H = dptr(a_H); v.b = dptr(a_b); v.S = dptr(a_S); v.bs = dptr(a_bs);
  v.x = dptr(a_x); v.linkQ = dptr(a_linkQ); v.ppart = dptr(a_ppart);
  v.bar = reinterpret_cast<unsigned*>(din + o_bar); v.abort_flag = reinterpret_cast<int*>(din + o_abort);
  v.red = dptr(a_red); v.ctrl = dptr(a_ctrl); v.nw = nw;
  v.force_heavy = getenv("OSH_LIBA_HEAVY_BARRIER") ? 1 : 0;
  v.eh = dptr(a_eh); v.ep = dptr(a_ep); v.bfull = dptr(a_bfull); v.E_total = E; v.EF_total = EF;
  v.link_colour = reinterpret_cast<const int*>(din + o_col);
  v.res_abort = reinterpret_cast<int*>(dres + r_abort); v.res_pose = reinterpret_cast<double*>(dres + r_pose); v.res_vba = reinterpret_cast<double*>(dres + r_vba);
  v.res_pts = reinterpret_cast<double*>(dres + r_pts); v.out_chi2 = reinterpret_cast<double*>(dres + r_chi2);
  v.out_depth = reinterpret_cast<unsigned char*>(dres + r_depth);
  {
    // opt in to large dynamic LDS: the attribute is per device, so once per device of the process
    static std::mutex attr_mu;
    static std::vector<int> attr_devices;
    std::lock_guard<std::mutex> attr_lock(attr_mu);
    if (std::find(attr_devices.begin(), attr_devices.end(), device) == attr_devices.end()) {
      OSH_HIP(hipFuncSetAttribute((const void*)k_liba<24>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
      OSH_HIP(hipFuncSetAttribute((const void*)k_liba<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
      attr_devices.push_back(device);
    }
  }
  // blocks per window: the tracker's single window (and small batches) get a group of 32 = one whole XCD; a large batch fills the chip
  // with one block per window.  A group needs all its blocks resident (they meet at barriers): the grid is kept within what the device
  // holds at once (occupancy query) and launched as an ordinary kernel.  (hipLaunchCooperativeKernel would check the same, but it
  // makes the runtime create a second, cooperative HSA queue, and under rocprofv3 the process then faults at exit inside
  // libhsa-runtime64's shutdown, called from libamdhip64's exit handler, on that queue's device mapping -- resolved from the fault
  // report and /proc/self/maps by profiles/exit_probe.py; k_liba was the only cooperative launch of the library.)
  int G = nw <= 8 ? kLG : (nw <= 16 ? 16 : (nw <= 32 ? 8 : (nw <= 64 ? 4 : (nw <= 128 ? 2 : 1))));
  if (const char* gs = getenv("OSH_LIBA_GROUP")) { const int gv = atoi(gs); if (gv == 1 || gv == 2 || gv == 4 || gv == 8 || gv == 16 || gv == 32) G = gv; }
  int W_arg = W;
  const void* kfn = NB == 24 ? (const void*)k_liba<24> : (const void*)k_liba<6>;
  {
    int per_cu = 0, n_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, kLT, lds) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) {
      (void)hipGetLastError();
      per_cu = 0;
    }
    while (G > 1 && (long long)((nw + 7) / 8 * 8) * G > (long long)per_cu * n_cu) G >>= 1;
  }
  auto run = [&](int Gx, int test_abort) -> int {
    v.test_abort = test_abort;
    int g_arg = Gx;
    void* args[] = {(void*)&v, (void*)&W_arg, (void*)&g_arg};
    const hipError_t le = hipLaunchKernel(kfn, dim3((unsigned)((nw + 7) / 8 * 8 * Gx)), dim3(kLT), args, lds, s);
    if (le != hipSuccess) { set_error("k_liba launch failed: %s", hipGetErrorString(le)); return OSH_ERR_DEVICE; }
    OSH_HIP(hipMemcpyAsync(hr, dres, out_bytes, hipMemcpyDeviceToHost, s));
    OSH_HIP(hipStreamSynchronize(s));
    return OSH_OK;
  };
  OSH_TRY(run(G, (G > 1 && getenv("OSH_LIBA_TEST_ABORT")) ? 1 : 0));
  if (*reinterpret_cast<const int*>(hr + r_abort) && G > 1) {
    // A barrier of a block group gave up (blocks of other streams kept part of a group off the device for seconds): the same
    // problem once more with one block per window, which has no barrier to wait at.  The estimates live in the input arena: upload again.
    OSH_HIP(hipMemcpyAsync(din, hs, in_bytes, hipMemcpyHostToDevice, s));
    G = 1;
    OSH_TRY(run(1, 0));
  }
  if (*reinterpret_cast<const int*>(hr + r_abort)) { set_error("k_liba: a barrier of a window's block group did not complete (group of %d blocks)", G); return OSH_ERR_DEVICE; }
  const LibaOut* h_out = reinterpret_cast<const LibaOut*>(hr + r_out);
  g_liba_last_group = G;
  std::memcpy(g_liba_last_prof, h_out[0].prof, sizeof(g_liba_last_prof));
  for (int w = 0; w < nw; ++w) {
    const LibaDesc& d = h_desc[w];
    const LibaOut& o = h_out[w];
    osh_liba_result& r = res[w];
    r.status = OSH_OK; r.iterations = o.iterations; r.trials = o.trials; r.n_trace = o.n_trace;
    r.chi2_initial = o.chi2_initial; r.chi2_final = o.chi2_final;
    for (int k = 0; k < o.n_trace; ++k) { r.chi2_trace[k] = o.chi2_trace[k]; r.lambda_trace[k] = o.lambda_trace[k]; r.trials_trace[k] = o.trials_trace[k]; }
    const double* q0 = reinterpret_cast<const double*>(hr + r_pose) + (size_t)(d.b_off / 15) * 24;
    const double* s0 = reinterpret_cast<const double*>(hr + r_vba) + (size_t)(d.b_off / 15) * 9;
    for (int k = 0; k < d.N; ++k) {
      const double* q = q0 + (size_t)k * 24;
      if (r.pose_Rcw) std::memcpy(r.pose_Rcw + 9 * k, q, 72);
      if (r.pose_tcw) std::memcpy(r.pose_tcw + 3 * k, q + 9, 24);
      if (r.pose_Rwb) std::memcpy(r.pose_Rwb + 9 * k, q + 12, 72);
      if (r.pose_twb) std::memcpy(r.pose_twb + 3 * k, q + 21, 24);
      if (r.vel) std::memcpy(r.vel + 3 * k, s0 + (size_t)k * 9, 24);
      if (r.bias_g) std::memcpy(r.bias_g + 3 * k, s0 + (size_t)k * 9 + 3, 24);
      if (r.bias_a) std::memcpy(r.bias_a + 3 * k, s0 + (size_t)k * 9 + 6, 24);
    }
    if (r.points && d.L) std::memcpy(r.points, reinterpret_cast<const double*>(hr + r_pts) + (size_t)d.pt_off * 3, (size_t)d.L * 24);
    if (r.edge_chi2 && d.E) std::memcpy(r.edge_chi2, reinterpret_cast<const double*>(hr + r_chi2) + d.edge_off, (size_t)d.E * 8);
    if (r.edge_depth_pos && d.E) std::memcpy(r.edge_depth_pos, reinterpret_cast<const unsigned char*>(hr + r_depth) + d.edge_off, (size_t)d.E);
  }
  return OSH_OK;
}

// phase cycle counters of window 0 of the last osh_liba_solve on this thread (block 0 of its group) and the group size used
extern "C" int osh_liba_get_profile(int32_t* group, int64_t cycles[8]) {
  if (group) *group = g_liba_last_group;
  if (cycles) for (int k = 0; k < 8; ++k) cycles[k] = g_liba_last_prof[k];
  return OSH_OK;
}
