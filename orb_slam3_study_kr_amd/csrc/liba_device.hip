// liba_device.hip -- Optimizer::LocalInertialBA's optimisation (src/Optimizer.cc:2843-2848) on MI355X.
//
// An inertial window is small (N <= 10/25 keyframes x 15 dof, O(10^3) landmarks, O(10^4) edges) and the
// reference runs exactly one at a time, so the whole Levenberg-Marquardt loop of a window is ONE persistent
// thread block: no host round trip, every reduction in a fixed order (bitwise reproducible), many windows =
// many blocks.  Phases inside the block (all separated by __syncthreads):
//   errors      computeActiveErrors + activeRobustChi2      thread per landmark over its edges + thread per link
//   linearise   buildSystem: visual blocks (Hll, b_l, Hpl per landmark; Hpp, b_p by one wavefront per pose),
//               EdgeInertial / EdgeGyroRW / EdgeAccRW blocks (BaseMultiEdge::constructQuadraticForm) link by link
//   trial       setLambda, Dinv, Schur rows (wavefront per pose row, register accumulators), blocked LDL^T
//               (ldlt_block.h), landmark back-substitution, ImuCamPose::Update, trial errors, gain ratio
// Vertex order of the reduced system = g2o's: the 6-dof poses of the N temporal keyframes, then (v, bg, ba).
#include "common.h"
#include "lba_math.h"
#include "ldlt_block.h"
#include "liba_math.h"
#include "liba_edges.h"
#include <algorithm>
#include <cfloat>
#include <cstring>
#include <vector>

namespace osh {

constexpr int kLT = 256;      // threads of the persistent block
constexpr int kLNB = 24;      // LDL^T panel width
constexpr int kLG = 16;       // blocks per window at most (one XCD's worth of a group)
constexpr int kPoseChunks = 8;   // a pose row's edges are summed in at most this many chunks
constexpr int kLinkQ = 832;   // per link: J^T W J (24x24), -J^T W r (24), then J (9x24), -W r (9), rho'
constexpr int kStageEdges = 32;                                   // edges of a pose row staged in LDS per pass of the Schur loop
constexpr int kStageMaxN = 32;                                    // partner-edge list entries per staged edge (N <= 25 optimisable keyframes)
constexpr size_t kStageDoublesPerWave = kStageEdges * 18 + (kStageEdges * kStageMaxN + 1) / 2;
__host__ __device__ constexpr size_t liba_scratch_doubles(int W) {
  return ldlt_lds_doubles(kLNB, W, kLT) > (kLT / 64) * kStageDoublesPerWave ? ldlt_lds_doubles(kLNB, W, kLT) : (kLT / 64) * kStageDoublesPerWave;
}
struct LibaOut {
  double chi2_initial, chi2_final;
  int iterations, trials, n_trace, sel;
  double chi2_trace[OSH_LBA_MAX_TRACE], lambda_trace[OSH_LBA_MAX_TRACE];
  int trials_trace[OSH_LBA_MAX_TRACE];
  long long prof2[8];
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
  const int* pel_off; const int* pel_edge;   // per optimisable pose: its edges in landmark order
  const int* lm_pose_edge;    // [L*N] sorted edge index of (landmark, optimisable pose) or -1
  const int* link_prev; const int* link_cur; const float* link_preint; const double* link_info; const double* link_info_g;
  const double* link_info_a; const unsigned char* link_robust;
  double* Hpl; double* Hll; double* bl; double* dinv;   // [E*18] [L*6] [L*3] [L*9]
  double* BD;                 // [E*18] B Dinv of every optimisable-pose edge (Schur step)
  double* H; double* b; double* S; double* bs; double* x;   // [n*n] [n] [n*n] [n] [n]
  double* linkQ;              // [NL*kLinkQ]
  double* ppart;              // [sum N][kPoseChunks][27] pose-row chunk sums: Hpp upper (21), b_p (6)
  unsigned* bar; int* abort_flag; double* red; double* ctrl;   // group barrier counters [nw], abort word, published sums [nw*4*kLG*2], [nw*4]
  int nw;
  double* out_chi2; unsigned char* out_depth;   // result arena: per edge, caller's order
  int* res_abort; double* res_pose; double* res_vba; double* res_pts;   // result arena: abort word, final [sum N][24], [sum N][9], [sum L][3]
};

// deterministic block reductions over kLT threads
__device__ __forceinline__ double blk_sum(double v, double* shw) {
  v = dev::wave_sum(v);
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

// reduced-state offset of vertex v (0..5) of link (a -> c), or -1 when the vertex is fixed
__device__ __forceinline__ int link_vertex_offset(int v, int a, int c, int N) {
  const int kf = (v < 4) ? a : c;
  if (kf >= N) return -1;
  if (v == 0 || v == 4) return 6 * kf;
  const int base = 6 * N + 9 * kf;
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
  unsigned gen, nred;
};
constexpr unsigned kSpinLimit = 1u << 22;   // polls (about a microsecond each) before a barrier gives up

__device__ __forceinline__ bool grp_sync(Grp& g, int* lds_flag) {
  __syncthreads();
  if (g.G == 1) return true;
  ++g.gen;
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(g.bar, 1u);
    const unsigned target = g.gen * (unsigned)g.G;
    int good = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(g.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0 && (spins > kSpinLimit || __hip_atomic_load(g.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { good = 0; break; }
    }
    if (!good) { __hip_atomic_store(g.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *g.res_abort = 1; }
    __threadfence();
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

// this thread's share of the robust chi2 of the state in buffer `sel` (computeActiveErrors + activeRobustChi2): one edge per
// thread and round, the inertial links on the last wavefront of the blocks
__device__ double eval_partial(const LibaView& v, const LibaDesc& d, int sel, const Grp& g) {
  const int tid = threadIdx.x;
  const double* poses = v.pose[sel] + (size_t)d.pose_off * 24;
  const double* vba = v.vba[sel] + (size_t)d.vel_off * 9;
  const double* pts = v.pts[sel] + (size_t)d.pt_off * 3;
  double acc = 0.0;
  if ((tid >> 6) == kLT / 64 - 1) {
    for (int l = (tid & 63) * g.G + g.m; l < d.NL; l += 64 * g.G) {
      const int gl = d.link_off + l;
      const int a = v.link_prev[gl], c = v.link_cur[gl];
      double r[9];
      inertial_residual(v.link_preint + (size_t)gl * OSH_PREINT_FLOATS, poses + 24 * a, vba + 9 * a, poses + 24 * c, vba + 9 * c, r);
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
  for (int e = g.m * kLT + tid; e < d.E; e += GT) {
    const size_t ge = (size_t)d.edge_off + e;
    const int kind = v.e_kind[ge];
    VisEval ev;
    vis_residual(d, kind, poses + 24 * (size_t)v.e_pose[ge], pts + 3 * (size_t)v.e_point[ge], v.e_obs + ge * 3, v.e_info[ge], ev);
    double r0, r1;
    dev::huber(ev.chi2, kind == OSH_EDGE_STEREO ? d.huber_stereo : d.huber_mono, r0, r1);
    acc += r0;
  }
  return acc;
}

#define OSH_GSYNC() do { if (!grp_sync(g, lds_flag)) return; } while (0)
#define OSH_PROF(i) do { if (prof_on) { const long long _n = clock64(); prof[i] += _n - prof_last; prof_last = _n; } } while (0)

__global__ __launch_bounds__(kLT) void k_liba(LibaView v, int W, int G) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  // consecutive workgroups go to the 8 XCDs in turn: the G blocks of a window are 8 apart, so they share one XCD and its L2
  const int bid = blockIdx.x;
  const int win = (bid / (8 * G)) * 8 + (bid & 7);
  if (win >= v.nw) return;
  const LibaDesc& d = v.desc[win];
  LibaOut& out = v.out[win];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Grp g;
  g.bar = v.bar + win; g.abort_flag = v.abort_flag; g.res_abort = v.res_abort; g.red = v.red + (size_t)win * 4 * kLG * 2; g.G = G; g.m = (bid >> 3) % G; g.gen = 0; g.nred = 0;
  const int m = g.m, GT = G * kLT, gt = m * kLT + tid, GW = G * (kLT / 64);
  const int N = d.N, n = d.n, L = d.L;
  // LDS carve: [0, ldlt) the LDL^T scratch (reused as general scratch between solves), then control words
  double* shw = sh + liba_scratch_doubles(W);             // [kLT/64] reductions
  int* lds_flag = reinterpret_cast<int*>(shw + kLT / 64 + 1);
  double* H = v.H + d.H_off; double* S = v.S + d.H_off;
  double* b = v.b + d.b_off; double* bs = v.bs + d.b_off; double* xg = v.x + d.b_off;
  double* Hpl = v.Hpl + (size_t)d.edge_off * 18;
  double* BD = v.BD + (size_t)d.edge_off * 18;
  double* Hll = v.Hll + (size_t)d.pt_off * 6; double* bl = v.bl + (size_t)d.pt_off * 3; double* dinv = v.dinv + (size_t)d.pt_off * 9;
  const int* lmo = v.lm_off + d.lmoff_off;
  const int* po = v.pel_off + d.peloff_off;
  const int* lmpe = v.lm_pose_edge + d.lmpose_off;
  double* linkQ = v.linkQ + (size_t)d.link_off * kLinkQ;
  double* ppart = v.ppart + (size_t)(d.b_off / 15) * kPoseChunks * 27;
  double* ctrl = v.ctrl + (size_t)win * 4;
  const bool prof_on = (m == 0 && tid == 0);
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long prof_last = clock64();
  // pose rows are summed in chunks so that a window's few keyframes still spread over the group's wavefronts
  int C = (GW - GW / 4) / (N > 0 ? N : 1);
  C = C < 1 ? 1 : (C > kPoseChunks ? kPoseChunks : C);

  if (m == 0 && tid < 8) out.prof2[tid] = 0;
  if (bid == 0 && tid == 0) *v.res_abort = 0;
  // the trial buffers start as copies of the estimates (the fixed keyframes and the fixed IMU state are only ever read)
  for (int k = gt; k < d.K * 24; k += GT) v.pose[1][(size_t)d.pose_off * 24 + k] = v.pose[0][(size_t)d.pose_off * 24 + k];
  for (int k = gt; k < d.NV * 9; k += GT) v.vba[1][(size_t)d.vel_off * 9 + k] = v.vba[0][(size_t)d.vel_off * 9 + k];
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
    const double* poses = v.pose[sel] + (size_t)d.pose_off * 24;
    const double* vba = v.vba[sel] + (size_t)d.vel_off * 9;
    const double* pts = v.pts[sel] + (size_t)d.pt_off * 3;
    const double iniChi = currentChi;
    // ---------------------------------------------------------------- linearise (buildSystem)
    for (int k = gt; k < n * n; k += GT) H[k] = 0.0;
    for (int k = gt; k < n; k += GT) b[k] = 0.0;
    // inertial links, Jacobians: link l belongs to block l mod G, one lane of that block's wavefront 3 each
    const long long tl0 = clock64();
    if (wave == 3 % (kLT / 64)) {
      for (int l = lane * G + m; l < d.NL; l += 64 * G) {
        const int gl = d.link_off + l;
        const int a = v.link_prev[gl], c = v.link_cur[gl];
        double* Q = linkQ + (size_t)l * kLinkQ;
        double* J = Q + 600;
        double r[9];
        const float* rec = v.link_preint + (size_t)gl * OSH_PREINT_FLOATS;
        inertial_residual(rec, poses + 24 * a, vba + 9 * a, poses + 24 * c, vba + 9 * c, r);
        inertial_jacobian(rec, poses + 24 * a, vba + 9 * a, poses + 24 * c, vba + 9 * c, J);
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
    if (m == 0 && tid == kLT - 64) out.prof2[0] += clock64() - tl0;
    const long long tl1 = clock64();
    // landmark side: thread per landmark (landmark j on block j mod G), its edges in order
    for (int j = tid * G + m; j < L; j += GT) {
      const double* X = pts + 3 * (size_t)j;
      double hl[6] = {0, 0, 0, 0, 0, 0}, bj[3] = {0, 0, 0};
      for (int e = lmo[j]; e < lmo[j + 1]; ++e) {
        const size_t ge = (size_t)d.edge_off + e;
        const int kind = v.e_kind[ge], ip = v.e_pose[ge];
        const double info = v.e_info[ge];
        VisEval ev;
        vis_residual(d, kind, poses + 24 * (size_t)ip, X, v.e_obs + ge * 3, info, ev);
        double r0, r1, JX[9], Jp[18];
        dev::huber(ev.chi2, kind == OSH_EDGE_STEREO ? d.huber_stereo : d.huber_mono, r0, r1);
        vis_jacobians(d, kind, poses + 24 * (size_t)ip, ev.Xc, JX, Jp);
        const double ww = r1 * info;
        const double wr[3] = {-(info * ev.r[0]) * r1, -(info * ev.r[1]) * r1, -(info * ev.r[2]) * r1};
        hl[0] += (JX[0] * ww) * JX[0] + (JX[3] * ww) * JX[3] + (JX[6] * ww) * JX[6];
        hl[1] += (JX[0] * ww) * JX[1] + (JX[3] * ww) * JX[4] + (JX[6] * ww) * JX[7];
        hl[2] += (JX[0] * ww) * JX[2] + (JX[3] * ww) * JX[5] + (JX[6] * ww) * JX[8];
        hl[3] += (JX[1] * ww) * JX[1] + (JX[4] * ww) * JX[4] + (JX[7] * ww) * JX[7];
        hl[4] += (JX[1] * ww) * JX[2] + (JX[4] * ww) * JX[5] + (JX[7] * ww) * JX[8];
        hl[5] += (JX[2] * ww) * JX[2] + (JX[5] * ww) * JX[5] + (JX[8] * ww) * JX[8];
#pragma unroll
        for (int i = 0; i < 3; ++i) bj[i] += JX[i] * wr[0] + JX[3 + i] * wr[1] + JX[6 + i] * wr[2];
        if (ip < N) {
          // the right-camera edge of a (keyframe, landmark) pair adds to the block of the left edge sorted just before it
          const bool second = e > lmo[j] && v.e_pose[ge - 1] == ip;
          double* Hb = Hpl + (size_t)(second ? e - 1 : e) * 18;
#pragma unroll
          for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
              const double hv = (Jp[i] * ww) * JX[jj] + (Jp[6 + i] * ww) * JX[3 + jj] + (Jp[12 + i] * ww) * JX[6 + jj];
              Hb[i * 3 + jj] = second ? Hb[i * 3 + jj] + hv : hv;
            }
        }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) Hll[(size_t)j * 6 + k] = hl[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) bl[(size_t)j * 3 + k] = bj[k];
    }
    if (m == 0 && tid == 0) out.prof2[1] += clock64() - tl1;
    const long long tl2 = clock64();
    // pose side: one wavefront per (optimisable pose, chunk of its edges), taken from the last wavefronts of the blocks
    for (int item = (kLT / 64 - 1 - wave) * G + m; item < N * C; item += GW) {
      const int i = item / C, ch = item - i * C;
      const int cnt = po[i + 1] - po[i];
      const int per = ((cnt + C - 1) / C + 63) / 64 * 64;
      const int lo = po[i] + ch * per, hi = min(po[i + 1], lo + per);
      double Hp[21], bp[6];
#pragma unroll
      for (int k = 0; k < 21; ++k) Hp[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) bp[k] = 0.0;
      const double* pose = poses + 24 * (size_t)i;
      for (int idx = lo + lane; idx < hi; idx += 64) {
        const int e = v.pel_edge[(size_t)d.pel_off + idx];
        const size_t ge = (size_t)d.edge_off + e;
        const int kind = v.e_kind[ge];
        const double info = v.e_info[ge];
        VisEval ev;
        vis_residual(d, kind, pose, pts + 3 * (size_t)v.e_point[ge], v.e_obs + ge * 3, info, ev);
        double r0, r1, JX[9], Jp[18];
        dev::huber(ev.chi2, kind == OSH_EDGE_STEREO ? d.huber_stereo : d.huber_mono, r0, r1);
        vis_jacobians(d, kind, pose, ev.Xc, JX, Jp);
        const double ww = r1 * info;
        const double wr[3] = {-(info * ev.r[0]) * r1, -(info * ev.r[1]) * r1, -(info * ev.r[2]) * r1};
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = a; c < 6; ++c) { Hp[q] += (Jp[a] * ww) * Jp[c] + (Jp[6 + a] * ww) * Jp[6 + c] + (Jp[12 + a] * ww) * Jp[12 + c]; ++q; }
          bp[a] += Jp[a] * wr[0] + Jp[6 + a] * wr[1] + Jp[12 + a] * wr[2];
        }
      }
#pragma unroll
      for (int k = 0; k < 21; ++k) Hp[k] = dev::wave_sum(Hp[k]);
#pragma unroll
      for (int k = 0; k < 6; ++k) bp[k] = dev::wave_sum(bp[k]);
      if (lane == 0) {
        double* o = ppart + ((size_t)i * kPoseChunks + ch) * 27;
#pragma unroll
        for (int k = 0; k < 21; ++k) o[k] = Hp[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) o[21 + k] = bp[k];
      }
    }
    if (m == 0 && tid == kLT - 64) out.prof2[2] += clock64() - tl2;
    __syncthreads();
    const long long tl3 = clock64();
    if (m == 0 && tid == 0) out.prof2[3] += tl3 - tl0;
    // quadratic forms of this block's links (BaseMultiEdge::constructQuadraticForm): W J first, then J^T (W J)
    for (int l = m; l < d.NL; l += G) {
      const int gl = d.link_off + l;
      double* Q = linkQ + (size_t)l * kLinkQ;
      const double* J = Q + 600;
      const double rho1 = J[216 + 9];
      const double* Om = v.link_info + (size_t)gl * 81;
      double* WJ = sh;            // [9][24]
      double* Js = sh + 216;      // [9][24]
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
    if (m == 0 && tid == 0) out.prof2[4] += clock64() - tl3;
    const long long tl4 = clock64();
    OSH_GSYNC();
    if (m == 0 && tid == 0) out.prof2[5] += clock64() - tl4;
    OSH_PROF(0);
    // ---- assembly by block 0: pose diagonal blocks from the chunk sums, then the links one after another (fixed order)
    if (m == 0) {
      for (int idx = tid; idx < N * 27; idx += kLT) {
        const int i = idx / 27, k = idx - i * 27;
        double s = 0.0;
        for (int ch = 0; ch < C; ++ch) s += ppart[((size_t)i * kPoseChunks + ch) * 27 + k];
        if (k < 21) {
          int a = 0, rem = k;
          while (rem >= 6 - a) { rem -= 6 - a; ++a; }
          const int c = a + rem;
          H[(size_t)(6 * i + a) * n + 6 * i + c] = s;
          H[(size_t)(6 * i + c) * n + 6 * i + a] = s;
        } else {
          b[6 * i + (k - 21)] = s;
        }
      }
      __syncthreads();
      for (int l = 0; l < d.NL; ++l) {
        const int gl = d.link_off + l;
        const int a = v.link_prev[gl], c = v.link_cur[gl];
        const double* Q = linkQ + (size_t)l * kLinkQ;
        auto vert_of = [](int col) { return col < 6 ? 0 : col < 9 ? 1 : col < 12 ? 2 : col < 15 ? 3 : col < 21 ? 4 : 5; };
        const int vbase[6] = {0, 6, 9, 12, 15, 21};
        for (int idx = tid; idx < 600; idx += kLT) {
          if (idx < 576) {
            const int ca = idx / 24, cb = idx - ca * 24;
            const int va = vert_of(ca), vb = vert_of(cb);
            if (vb < va) continue;   // upper blocks + mirrored below
            const int oa = link_vertex_offset(va, a, c, N), ob = link_vertex_offset(vb, a, c, N);
            if (oa < 0 || ob < 0) continue;
            const int ra = oa + (ca - vbase[va]), rb = ob + (cb - vbase[vb]);
            H[(size_t)ra * n + rb] += Q[idx];
            if (va != vb) H[(size_t)rb * n + ra] += Q[idx];
          } else {
            const int ca = idx - 576;
            const int va = vert_of(ca);
            const int oa = link_vertex_offset(va, a, c, N);
            if (oa >= 0) b[oa + (ca - vbase[va])] += Q[idx];
          }
        }
        __syncthreads();
        // EdgeGyroRW / EdgeAccRW: r = b2 - b1, J = [-I, I], plain information
        if (tid < 18) {
          const int which = tid / 9, i = (tid % 9) / 3, j = tid % 3;
          const double* Og = (which == 0 ? v.link_info_g : v.link_info_a) + (size_t)gl * 9;
          const int o1 = (a < N) ? 6 * N + 9 * a + 3 + 3 * which : -1;
          const int o2 = 6 * N + 9 * c + 3 + 3 * which;
          const double gg = Og[i * 3 + j];
          if (o1 >= 0) {
            H[(size_t)(o1 + i) * n + o1 + j] += gg;
            H[(size_t)(o1 + i) * n + o2 + j] += -gg;
            H[(size_t)(o2 + j) * n + o1 + i] += -gg;
          }
          H[(size_t)(o2 + i) * n + o2 + j] += gg;
          if (j == 0) {
            double rb[3];
            for (int k = 0; k < 3; ++k) rb[k] = vba[9 * c + 3 + 3 * which + k] - vba[9 * a + 3 + 3 * which + k];
            const double Or = -(Og[i * 3] * rb[0] + Og[i * 3 + 1] * rb[1] + Og[i * 3 + 2] * rb[2]);
            if (o1 >= 0) b[o1 + i] += -Or;
            b[o2 + i] += Or;
          }
        }
        __syncthreads();
      }
      if (it == 0) {
        double l0 = d.lambda_init;
        if (!(d.lambda_init > 0)) {
          double mx = 0.0;
          for (int k = tid; k < n; k += kLT) mx = fmax(mx, fabs(H[(size_t)k * n + k]));
          for (int j = tid; j < L; j += kLT) mx = fmax(mx, fmax(fabs(Hll[(size_t)j * 6]), fmax(fabs(Hll[(size_t)j * 6 + 3]), fabs(Hll[(size_t)j * 6 + 5]))));
          l0 = 1e-5 * blk_max(mx, shw);
        }
        if (tid == 0) ctrl[0] = l0;
      }
    }
    OSH_GSYNC();
    if (it == 0) { lambda = ctrl[0]; ni = 2.0; nBad = 0; }
    OSH_PROF(1);
    // ---------------------------------------------------------------- LM trials
    double rho = 0.0;
    int qmax = 0;
    do {
      const int trs = sel ^ 1;
      // Dinv and Dinv b_l per landmark (setLambda on Hll, block_solver.hpp:389,582-587), and B Dinv of its optimisable-pose edges
      for (int j = tid * G + m; j < L; j += GT) {
        const double* hl = Hll + (size_t)j * 6;
        double Di[9];
        dev::inv3_sym(hl[0] + lambda, hl[1], hl[2], hl[3] + lambda, hl[4], hl[5] + lambda, Di);
        const double b0 = bl[(size_t)j * 3], b1 = bl[(size_t)j * 3 + 1], b2 = bl[(size_t)j * 3 + 2];
        double* o = dinv + (size_t)j * 9;
        o[0] = Di[0]; o[1] = Di[1]; o[2] = Di[2]; o[3] = Di[4]; o[4] = Di[5]; o[5] = Di[8];
        o[6] = Di[0] * b0 + Di[1] * b1 + Di[2] * b2; o[7] = Di[3] * b0 + Di[4] * b1 + Di[5] * b2; o[8] = Di[6] * b0 + Di[7] * b1 + Di[8] * b2;
        for (int e = lmo[j]; e < lmo[j + 1]; ++e) {
          const int ip = v.e_pose[(size_t)d.edge_off + e];
          if (ip >= N) continue;
          if (e > lmo[j] && v.e_pose[(size_t)d.edge_off + e - 1] == ip) continue;   // the pair's block lives in the first edge's slot
          const double* Be = Hpl + (size_t)e * 18;
          double* od = BD + (size_t)e * 18;
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            const double x0 = Be[r * 3], x1 = Be[r * 3 + 1], x2 = Be[r * 3 + 2];
            od[r * 3 + 0] = x0 * Di[0] + x1 * Di[1] + x2 * Di[2];
            od[r * 3 + 1] = x0 * Di[1] + x1 * Di[4] + x2 * Di[5];
            od[r * 3 + 2] = x0 * Di[2] + x1 * Di[5] + x2 * Di[8];
          }
        }
      }
      // S = H + lambda I (upper), rhs = b
      for (int k = gt; k < n * n; k += GT) { const int r = k / n, c = k - r * n; S[k] = H[k] + ((r == c) ? lambda : 0.0); }
      for (int k = gt; k < n; k += GT) bs[k] = b[k];
      OSH_GSYNC();
      OSH_PROF(2);
      // Schur complement (block_solver.hpp:381-432), landmark-parallel: one wavefront per pose PAIR (i <= i2), the lanes stride
      // the landmarks, a landmark seen by both poses adds (B Dinv)_i B_i2^T to the lane's 6x6 partial, the 36 partials are summed
      // by a fixed butterfly and subtracted from S(i, i2); the diagonal pairs also take the rhs term B (Dinv b_l) of their pose.
      const int npairs = N * (N + 1) / 2;
      for (int pr = wave * G + m; pr < npairs; pr += GW) {
        int i = 0, rem = pr;
        while (rem >= N - i) { rem -= N - i; ++i; }
        const int i2 = i + rem;
        double acc[36], ci[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = 0.0;
        for (int j = lane; j < L; j += 64) {
          const int e1 = lmpe[(size_t)j * N + i], e2 = lmpe[(size_t)j * N + i2];
          if (e1 < 0 || e2 < 0) continue;
          const double* A1 = BD + (size_t)e1 * 18;
          const double* B2 = Hpl + (size_t)e2 * 18;
          double b2[18];
#pragma unroll
          for (int k = 0; k < 18; ++k) b2[k] = B2[k];
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            const double a0 = A1[r * 3], a1 = A1[r * 3 + 1], a2 = A1[r * 3 + 2];
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[r * 6 + c] += a0 * b2[c * 3] + a1 * b2[c * 3 + 1] + a2 * b2[c * 3 + 2];
          }
          if (i == i2) {
            const double* Dj = dinv + (size_t)j * 9;
#pragma unroll
            for (int r = 0; r < 6; ++r) ci[r] += b2[r * 3] * Dj[6] + b2[r * 3 + 1] * Dj[7] + b2[r * 3 + 2] * Dj[8];
          }
        }
#pragma unroll
        for (int k = 0; k < 36; ++k) acc[k] = dev::wave_sum(acc[k]);
        if (lane < 36) {
          double val = acc[0];
#pragma unroll
          for (int k = 1; k < 36; ++k) val = (lane == k) ? acc[k] : val;
          const int r = lane / 6, c = lane - r * 6;
          S[(size_t)(6 * i + r) * n + 6 * i2 + c] -= val;
        }
        if (i == i2) {
#pragma unroll
          for (int r = 0; r < 6; ++r) ci[r] = dev::wave_sum(ci[r]);
          if (lane < 6) {
            double c = ci[0];
            if (lane == 1) c = ci[1]; else if (lane == 2) c = ci[2]; else if (lane == 3) c = ci[3];
            else if (lane == 4) c = ci[4]; else if (lane == 5) c = ci[5];
            bs[6 * i + lane] -= c;
          }
        }
      }
      OSH_GSYNC();
      OSH_PROF(3);
      if (m == 0) {
        double *xs, *shw2;
        const bool okb = ldlt_solve_block<kLNB, kLT>(S, bs, n, W, sh, xs, shw2);
        for (int k = tid; k < n; k += kLT) xg[k] = xs[k];
        if (tid == 0) ctrl[1] = okb ? 1.0 : 0.0;
      }
      OSH_GSYNC();
      const bool ok2 = ctrl[1] != 0.0;
      OSH_PROF(4);
      // landmark back-substitution, point update, landmark part of computeScale
      double sc = 0.0;
      double* pts_t = v.pts[trs] + (size_t)d.pt_off * 3;
      for (int j = tid * G + m; j < L; j += GT) {
        const double* Dj = dinv + (size_t)j * 9;
        const double b0 = bl[(size_t)j * 3], b1 = bl[(size_t)j * 3 + 1], b2 = bl[(size_t)j * 3 + 2];
        double c0 = b0, c1 = b1, c2 = b2;
        for (int e = lmo[j]; e < lmo[j + 1]; ++e) {
          const int ip = v.e_pose[(size_t)d.edge_off + e];
          if (ip >= N) continue;
          if (e > lmo[j] && v.e_pose[(size_t)d.edge_off + e - 1] == ip) continue;   // block already taken with the pair's first edge
          const double* B = Hpl + (size_t)e * 18;
          const double* xp = xg + 6 * ip;
          double a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
          for (int r = 0; r < 6; ++r) { const double mx = -xp[r]; a0 += B[r * 3] * mx; a1 += B[r * 3 + 1] * mx; a2 += B[r * 3 + 2] * mx; }
          c0 += a0; c1 += a1; c2 += a2;
        }
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
      // pose / velocity / bias update into the trial buffers (ImuCamPose::Update, src/G2oTypes.cc:187-220): last block of the group
      double* poses_t = v.pose[trs] + (size_t)d.pose_off * 24;
      double* vba_t = v.vba[trs] + (size_t)d.vel_off * 9;
      if (m == G - 1) {
        for (int k = kLT - 1 - tid; k < N; k += kLT) {
          const double* pu = xg + 6 * k;
          const double* P = poses + 24 * (size_t)k;
          double* Q = poses_t + 24 * (size_t)k;
          double tw[3], E[9], Rwb[9], Rbw[9], tbw[3], tc[3];
          imu::m3_vec(P + 12, pu + 3, tw);
          for (int i = 0; i < 3; ++i) Q[21 + i] = P[21 + i] + tw[i];
          imu::exp_so3(pu, E);
          imu::m3_mul(P + 12, E, Rwb);
          for (int i = 0; i < 9; ++i) Q[12 + i] = Rwb[i];
          for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rbw[i * 3 + j] = Rwb[j * 3 + i];
          imu::m3_vec(Rbw, Q + 21, tbw);
          tbw[0] = -tbw[0]; tbw[1] = -tbw[1]; tbw[2] = -tbw[2];
          imu::m3_mul(d.Rcb, Rbw, Q);
          imu::m3_vec(d.Rcb, tbw, tc);
          for (int i = 0; i < 3; ++i) Q[9 + i] = tc[i] + d.tcb[i];
          for (int i = 0; i < 9; ++i) vba_t[9 * k + i] = vba[9 * k + i] + xg[6 * N + 9 * k + i];
        }
        for (int k = tid; k < n; k += kLT) sc += xg[k] * (lambda * xg[k] + b[k]);
      }
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
        poses = v.pose[sel] + (size_t)d.pose_off * 24; vba = v.vba[sel] + (size_t)d.vel_off * 9; pts = v.pts[sel] + (size_t)d.pt_off * 3;
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
  DevBuf in, res, pose1, vba1, pts1, Hpl, BD, Hll, bl, dinv, H, b, S, bs, x, linkQ, ppart, red, ctrl;
};
LibaBuffers& liba_buffers() { static thread_local LibaBuffers b; return b; }
thread_local int g_liba_last_group = 0;
thread_local long long g_liba_last_prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
}  // namespace

#define OSH_TRY(expr) do { int _rc = (expr); if (_rc != OSH_OK) return _rc; } while (0)

extern "C" int osh_lba_stream(osh_lba_ctx* ctx, int* device, hipStream_t* stream);   // lba_device.hip

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
    d.n = 15 * d.N; d.max_iter = p.max_iterations;
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
    K += d.K; NV += d.NV; L += d.L; E += d.E; NL += d.NL; Htot += (size_t)d.n * d.n; btot += d.n; LO += (size_t)d.L + 1; PO += (size_t)d.N + 1;
    EF += ef; LP += (size_t)d.L * d.N;
    n_max = std::max(n_max, d.n);
  }
  if (n_max / 15 > 30) { set_error("inertial window with %d optimisable keyframes: the device path handles up to 30 (the reference uses 10 or 25)", n_max / 15); return OSH_ERR_UNSUPPORTED; }
  const int W = ldlt_row_stride(n_max);
  const size_t lds = (liba_scratch_doubles(W) + kLT / 64 + 8) * sizeof(double);
  if (lds > 160 * 1024 - 64) { set_error("inertial window with %d keyframes exceeds the LDS budget", n_max / 15); return OSH_ERR_UNSUPPORTED; }
  // ---- pack: every input array goes into ONE pinned staging buffer and travels in ONE copy (an upload per array cost more than
  // the optimisation of a single window); the device pointers are offsets into the arena.
  LibaBuffers& B = liba_buffers();
  size_t in_bytes = 0;
  auto take = [&](size_t bytes) { const size_t o = in_bytes; in_bytes = (in_bytes + bytes + 255) & ~(size_t)255; return o; };
  const size_t o_desc = take(nw * sizeof(LibaDesc)), o_pose = take(K * 24 * 8), o_vba = take(NV * 9 * 8), o_pts = take(L * 3 * 8), o_obs = take(E * 3 * 8),
               o_info = take(E * 8), o_ep = take(E * 4), o_el = take(E * 4), o_eo = take(E * 4), o_lmo = take(LO * 4), o_po = take(PO * 4),
               o_pel = take(EF * 4), o_lmpe = take(LP * 4), o_lp = take(NL * 4), o_lc = take(NL * 4), o_kind = take(E), o_rob = take(NL),
               o_pre = take(NL * OSH_PREINT_FLOATS * 4), o_li = take(NL * 81 * 8), o_lg = take(NL * 9 * 8), o_la = take(NL * 9 * 8),
               o_bar = take(nw * sizeof(unsigned)), o_abort = take(sizeof(int));
  char* hs = static_cast<char*>(B.h_in.reserve(in_bytes));
  if (!hs) { set_error("osh_liba_solve: pinned staging allocation of %zu bytes failed", in_bytes); return OSH_ERR_DEVICE; }
  LibaDesc* h_descp = reinterpret_cast<LibaDesc*>(hs + o_desc);
  double* h_pose = reinterpret_cast<double*>(hs + o_pose); double* h_vba = reinterpret_cast<double*>(hs + o_vba);
  double* h_pts = reinterpret_cast<double*>(hs + o_pts); double* h_obs = reinterpret_cast<double*>(hs + o_obs);
  double* h_info = reinterpret_cast<double*>(hs + o_info);
  int* h_ep = reinterpret_cast<int*>(hs + o_ep); int* h_el = reinterpret_cast<int*>(hs + o_el); int* h_eo = reinterpret_cast<int*>(hs + o_eo);
  int* h_lmo = reinterpret_cast<int*>(hs + o_lmo); int* h_po = reinterpret_cast<int*>(hs + o_po); int* h_pel = reinterpret_cast<int*>(hs + o_pel);
  int* h_lmpe = reinterpret_cast<int*>(hs + o_lmpe); int* h_lp = reinterpret_cast<int*>(hs + o_lp); int* h_lc = reinterpret_cast<int*>(hs + o_lc);
  unsigned char* h_kind = reinterpret_cast<unsigned char*>(hs + o_kind); unsigned char* h_rob = reinterpret_cast<unsigned char*>(hs + o_rob);
  float* h_pre = reinterpret_cast<float*>(hs + o_pre);
  double* h_li = reinterpret_cast<double*>(hs + o_li); double* h_lg = reinterpret_cast<double*>(hs + o_lg); double* h_la = reinterpret_cast<double*>(hs + o_la);
  std::memcpy(h_descp, h_desc.data(), nw * sizeof(LibaDesc));
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
    for (int x = 0; x < d.E; ++x) { const int ip = h_ep[(size_t)d.edge_off + x]; if (ip < d.N) h_pel[(size_t)d.pel_off + fill[ip]++] = x; }
    for (int l = 0; l < d.NL; ++l) {
      const size_t g = (size_t)d.link_off + l;
      h_lp[g] = p.link_prev[l]; h_lc[g] = p.link_cur[l]; h_rob[g] = p.link_robust[l];
      std::memcpy(&h_pre[g * OSH_PREINT_FLOATS], p.link_preint + (size_t)l * OSH_PREINT_FLOATS, OSH_PREINT_FLOATS * 4);
      std::memcpy(&h_li[g * 81], p.link_info + (size_t)l * 81, 81 * 8);
      std::memcpy(&h_lg[g * 9], p.link_info_g + (size_t)l * 9, 72); std::memcpy(&h_la[g * 9], p.link_info_a + (size_t)l * 9, 72);
    }
  }
  // ---- result arena (one copy back): LibaOut per window, abort word, final poses / velocities+biases / points, edge chi2 and depth flags
  size_t out_bytes = 0;
  auto take_out = [&](size_t bytes) { const size_t o = out_bytes; out_bytes = (out_bytes + bytes + 255) & ~(size_t)255; return o; };
  const size_t r_out = take_out(nw * sizeof(LibaOut)), r_abort = take_out(sizeof(int)), r_pose = take_out((btot / 15) * 24 * 8), r_vba = take_out((btot / 15) * 9 * 8),
               r_pts = take_out(L * 3 * 8), r_chi2 = take_out(E * 8), r_depth = take_out(E);
  char* hr = static_cast<char*>(B.h_out.reserve(out_bytes));
  if (!hr) { set_error("osh_liba_solve: pinned result allocation of %zu bytes failed", out_bytes); return OSH_ERR_DEVICE; }
  auto R = [](DevBuf& b, size_t bytes) { return b.reserve(std::max<size_t>(bytes, 8)); };
  OSH_TRY(R(B.in, in_bytes)); OSH_TRY(R(B.res, out_bytes));
  OSH_TRY(R(B.pose1, K * 24 * 8)); OSH_TRY(R(B.vba1, NV * 9 * 8)); OSH_TRY(R(B.pts1, L * 3 * 8));
  OSH_TRY(R(B.Hpl, E * 18 * 8)); OSH_TRY(R(B.BD, E * 18 * 8)); OSH_TRY(R(B.Hll, L * 6 * 8)); OSH_TRY(R(B.bl, L * 3 * 8));
  OSH_TRY(R(B.dinv, L * 9 * 8)); OSH_TRY(R(B.H, Htot * 8)); OSH_TRY(R(B.S, Htot * 8)); OSH_TRY(R(B.b, btot * 8)); OSH_TRY(R(B.bs, btot * 8));
  OSH_TRY(R(B.x, btot * 8)); OSH_TRY(R(B.linkQ, NL * kLinkQ * 8)); OSH_TRY(R(B.ppart, (btot / 15) * kPoseChunks * 27 * 8));
  OSH_TRY(R(B.red, (size_t)nw * 4 * kLG * 2 * 8)); OSH_TRY(R(B.ctrl, (size_t)nw * 4 * 8));
  OSH_HIP(hipMemcpyAsync(B.in.p, hs, in_bytes, hipMemcpyHostToDevice, s));
  char* din = B.in.as<char>();
  char* dres = B.res.as<char>();
  LibaView v{};
  v.desc = reinterpret_cast<const LibaDesc*>(din + o_desc); v.out = reinterpret_cast<LibaOut*>(dres + r_out);
  v.pose[0] = reinterpret_cast<double*>(din + o_pose); v.vba[0] = reinterpret_cast<double*>(din + o_vba); v.pts[0] = reinterpret_cast<double*>(din + o_pts);
  v.pose[1] = B.pose1.as<double>(); v.vba[1] = B.vba1.as<double>(); v.pts[1] = B.pts1.as<double>();
  v.e_pose = reinterpret_cast<const int*>(din + o_ep); v.e_point = reinterpret_cast<const int*>(din + o_el);
  v.e_kind = reinterpret_cast<const unsigned char*>(din + o_kind); v.e_obs = reinterpret_cast<const double*>(din + o_obs);
  v.e_info = reinterpret_cast<const double*>(din + o_info); v.e_orig = reinterpret_cast<const int*>(din + o_eo);
  v.lm_off = reinterpret_cast<const int*>(din + o_lmo); v.pel_off = reinterpret_cast<const int*>(din + o_po);
  v.pel_edge = reinterpret_cast<const int*>(din + o_pel); v.lm_pose_edge = reinterpret_cast<const int*>(din + o_lmpe);
  v.link_prev = reinterpret_cast<const int*>(din + o_lp); v.link_cur = reinterpret_cast<const int*>(din + o_lc);
  v.link_preint = reinterpret_cast<const float*>(din + o_pre); v.link_info = reinterpret_cast<const double*>(din + o_li);
  v.link_info_g = reinterpret_cast<const double*>(din + o_lg); v.link_info_a = reinterpret_cast<const double*>(din + o_la);
  v.link_robust = reinterpret_cast<const unsigned char*>(din + o_rob);
  v.Hpl = B.Hpl.as<double>(); v.BD = B.BD.as<double>(); v.Hll = B.Hll.as<double>(); v.bl = B.bl.as<double>();
  v.dinv = B.dinv.as<double>(); v.H = B.H.as<double>(); v.b = B.b.as<double>(); v.S = B.S.as<double>(); v.bs = B.bs.as<double>();
  v.x = B.x.as<double>(); v.linkQ = B.linkQ.as<double>(); v.ppart = B.ppart.as<double>();
  v.bar = reinterpret_cast<unsigned*>(din + o_bar); v.abort_flag = reinterpret_cast<int*>(din + o_abort);
  v.red = B.red.as<double>(); v.ctrl = B.ctrl.as<double>(); v.nw = nw;
  v.res_abort = reinterpret_cast<int*>(dres + r_abort); v.res_pose = reinterpret_cast<double*>(dres + r_pose); v.res_vba = reinterpret_cast<double*>(dres + r_vba);
  v.res_pts = reinterpret_cast<double*>(dres + r_pts); v.out_chi2 = reinterpret_cast<double*>(dres + r_chi2);
  v.out_depth = reinterpret_cast<unsigned char*>(dres + r_depth);
  static bool attr_done = false;
  if (!attr_done) { OSH_HIP(hipFuncSetAttribute((const void*)k_liba, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); attr_done = true; }
  // blocks per window: the tracker's single window (and small batches) get a group of 16 on one XCD; a large batch fills the chip
  // with one block per window.  A group needs all its blocks resident (they meet at barriers): cooperative launch checks that.
  int G = nw <= 16 ? kLG : (nw <= 32 ? 8 : (nw <= 64 ? 4 : (nw <= 128 ? 2 : 1)));
  if (const char* gs = getenv("OSH_LIBA_GROUP")) { const int gv = atoi(gs); if (gv == 1 || gv == 2 || gv == 4 || gv == 8 || gv == 16) G = gv; }
  int W_arg = W;
  hipError_t le = hipSuccess;
  if (G > 1) {
    void* args[] = {(void*)&v, (void*)&W_arg, (void*)&G};
    le = hipLaunchCooperativeKernel((const void*)k_liba, dim3((unsigned)((nw + 7) / 8 * 8 * G)), dim3(kLT), args, (unsigned)lds, s);
    if (le == hipErrorCooperativeLaunchTooLarge) { (void)hipGetLastError(); G = 1; le = hipSuccess; }
  }
  if (G == 1) {
    hipLaunchKernelGGL(k_liba, dim3((unsigned)((nw + 7) / 8 * 8)), dim3(kLT), lds, s, v, W, 1);
    le = hipGetLastError();
  }
  if (le != hipSuccess) { set_error("k_liba launch failed: %s", hipGetErrorString(le)); return OSH_ERR_DEVICE; }
  OSH_HIP(hipMemcpyAsync(hr, dres, out_bytes, hipMemcpyDeviceToHost, s));
  OSH_HIP(hipStreamSynchronize(s));
  if (*reinterpret_cast<const int*>(hr + r_abort)) { set_error("k_liba: a barrier of a window's block group did not complete (group of %d blocks)", G); return OSH_ERR_DEVICE; }
  const LibaOut* h_out = reinterpret_cast<const LibaOut*>(hr + r_out);
  g_liba_last_group = G;
  std::memcpy(g_liba_last_prof, h_out[0].prof, sizeof(g_liba_last_prof));
  if (getenv("OSH_LIBA_PROF2")) for (int k = 0; k < 6; ++k) fprintf(stderr, "prof2[%d] = %lld\n", k, h_out[0].prof2[k]);
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
