// lba_device.hip -- local bundle adjustment on MI355X (gfx950): HIP kernels + C-ABI driver.
//
// Replaces, for a batch of independent windows, what
//   optimizer.initializeOptimization(); optimizer.optimize(10);     src/Optimizer.cc:1410-1411
// does on one CPU thread in the reference (g2o LM + Schur block solver).  See
// DESIGN.md for the data layout and the roofline of each kernel.
//
// Kernel map (reference loop -> kernel), "g2o/" = Thirdparty/g2o/g2o/:
//   k_lin_items   computeActiveErrors + activeRobustChi2 + buildSystem
//                 g2o/core/sparse_optimizer.cpp:61-114, block_solver.hpp:502-560,
//                 base_binary_edge.hpp:55-120  (Hll, b_l, Hpl, per-group Hpp / b_p partials)
//   k_pose_reduce the pose side of buildSystem (Hpp, b_p) from the group partials, fixed order
//   k_residual    computeActiveErrors + activeRobustChi2 at the trial estimates
//   k_schur_items BlockSolver::solve Schur part, block_solver.hpp:367-439: landmarks grouped by observer
//                 set (schur_plan.h), one wavefront per group, BD * W^T on the FP64 matrix cores
//   k_schur_reduce  S = Hpp + lambda I - sum of the group products, b_s = b_p - ..., fixed order
//   k_solve       LinearSolverEigen::solve -> dense blocked LDL^T, g2o/solvers/linear_solver_eigen.h:94-124,
//                 + pose update (VertexSE3Expmap::oplusImpl)
//   k_backsub     landmark back-substitution block_solver.hpp:461-483 + VertexSBAPointXYZ::oplusImpl
//                 + computeScale partials (optimization_algorithm_levenberg.cpp:187-194)
//   k_control     the Levenberg-Marquardt controller, optimization_algorithm_levenberg.cpp:61-169,
//                 and the optimize() loop conditions, sparse_optimizer.cpp:354-419
//   k_finalize    per-edge chi2 / isDepthPositive for the outlier test, src/Optimizer.cc:1413-1460
#include "common.h"
#include <type_traits>
#include <cstdlib>
#include <cstdio>
#include <atomic>
#include <mutex>
#include <thread>
#include "lba_math.h"
#include "ldlt_block.h"
#include "schur_plan.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace osh {

constexpr int kBlock = 256;        // threads per block of the edge/landmark kernels
constexpr int kChunkEdges = 256;   // edges handled per pass of a chunk (== kBlock)
constexpr int kResidualSplit = 4;    // blocks of k_residual per chunk (kChunkMaxEdges / kChunkEdges)
constexpr int kChunkMaxEdges = 1024;   // edges of one chunk (= one block of k_residual / k_backsub / k_finalize): four passes amortise
                                       // the per-block staging of the window's poses and x_p in k_backsub
constexpr double kTau = 1e-5;      // OptimizationAlgorithmLevenberg::_tau
constexpr int kMaxTrials = 10;     // maxTrialsAfterFailure
// k_solve runs one block per window.  The factorisation is a chain of dependent pivots: with many windows two 256-thread
// blocks per CU (LDS <= 75 KB each, 12-wide panels) overlap their latencies (measured 1.21 ms per 512 windows against 1.70 ms
// for one 512-thread block per CU; 128 / 192 / 384 threads: 1.79 / 1.39 / 1.65); with fewer windows than CUs one 512-thread
// block with 24-wide panels finishes a single system sooner.
constexpr int kSolveThreadsBatch = 256, kSolveThreadsLatency = 512;

struct WinDesc {
  int P, F, L, E;
  int pose_off;    // first pose of the window in the pose arrays (P+F poses per window)
  int fpose_off;   // first optimisable pose in the per-free-pose arrays
  int pt_off;      // first landmark
  int edge_off;    // first (sorted) edge
  int lmoff_off;   // start of this window's L+1 landmark->edge offsets
  int peloff_off;  // start of this window's P+1 pose->edge-list offsets
  int pel_off;     // start of this window's pose edge list
  int chunk_off, n_chunks;
  int sitem_off, n_sitems;  // this window's symmetric items of the Schur plan (they also carry the linearisation)
  int aux_off, n_aux;       // this window's chunks of k_lin_aux
  int n;           // 6P
  int max_iter;
  long long S_off; // doubles
  double huber_mono, huber_stereo, lambda_init;
  double kb8[4];   // KannalaBrandt8 k1..k4 of the window's camera (osh_lba_problem.kb8)
  int kb8_on;      // 1: the window's mono edges project through KannalaBrandt8
};

struct LmState {
  double lambda, ni, currentChi, iniChi, tempChi, rho, scale_pose, chi2_initial;
  int iter;           // index of the running solve() call
  int qmax;           // trials of the running iteration
  int nBad;
  int active;         // window still inside optimize()
  int need_lin;       // next round opens a new iteration (linearise at cur)
  int sel;            // state buffer holding the current estimates
  int last_eval_sel;  // buffer whose errors computeActiveErrors saw last
  int iterations;     // cjIterations
  int trials;
  int solve_ok;
  int stop;           // host stop flag snapshot
  int n_trace;
  double chi2_trace[OSH_LBA_MAX_TRACE];
  double lambda_trace[OSH_LBA_MAX_TRACE];
  int trials_trace[OSH_LBA_MAX_TRACE];
};

struct Chunk { int win, lm0, lm1; };

// Everything the kernels need, passed by value.
struct BatchView {
  int n_windows, n_chunks, n_fposes;
  const WinDesc* win;
  LmState* lm;
  const Chunk* chunks;
  const int* fpose_win;      // [n_fposes] window of each optimisable pose
  // state: [2] buffers
  double* pose_state[2];     // [NP*7]
  double* pt_state[2];       // [NL*3]
  const double* pose_cam;    // [NP*5]
  // sorted edges (landmark-major, poses ascending inside a landmark, free poses first)
  const int* e_pose;         // [NE] window-local pose index
  const int* e_point;        // [NE] window-local landmark index
  const unsigned char* e_kind;
  const double* e_rec;       // [NE*4] u v u_right invSigma2 of every sorted edge: one 32-byte record (the observation and its
                             // weight are always read together; as two arrays they cost two partial cache lines per landmark)
  const int* e_orig;         // [NE] index in the caller's edge order
  const int* lm_off;         // per window L+1 offsets (window-local edge index)
  const int* lm_nfree;       // [NL] free-pose edges of each landmark
  const int* pel_off;        // per window P+1
  const int* pel_edge;       // [NEfree] window-local sorted edge index, landmark order
  // Schur plan (schur_plan.h)
  const SItem* sitems;       // [n_items] symmetric items first
  const SRec* srecs;         // landmark records of the items
  const int* spair;          // [n_items*64] contribution index of pose pair (sa,sb) or -1
  const int* scslot;         // [n_items*8] rhs contribution index of row pose sa or -1
  const int* sposex;         // [n_items*8] window-local row pose of slot sa or -1
  const int2* pose_crange;   // [NFP] {first, count} of the pose's (item, pose) contributions
  double* hcontrib;          // [n_ccontrib*27] per linearisation: upper(Hpp) (21) + b_p (6) of one item and pose
  double* chi_item;          // [n_sym] robust chi2 partial of each symmetric item
  double* dmax_item;         // [n_sym] largest Hll diagonal entry seen by the item
  const int4* aux_chunks;    // [n_aux_chunks] {window, first entry, entries, 0} of k_lin_aux
  const int2* aux_entries;   // {landmark, sorted edge}, window-local
  double* chi_aux;           // [n_aux_chunks]
  double* dmax_aux;          // [n_aux_chunks]
  const RBlk* rblk;          // [n_rblk] blocks of S + rhs segments with their contribution ranges
  int n_rblk;
  double* contrib;           // [n_contrib*36] per trial: 6x6 products of one item and pose pair
  double* ccontrib;          // [n_ccontrib*6] per trial: rhs products of one item and pose
  double* dinv;              // [NL*9] unused since the landmark inverse is re-formed in k_backsub (kept for the debug exports)
  // system
  double* Hpl;               // [NE*18] 6x3 row-major per sorted edge
  double* Hll;               // [NL*6] upper: 00 01 02 11 12 22
  double* bl;                // [NL*3]
  double* Hpp;               // [NFP*36]
  double* bp;                // [NFP*6]
  double* S;                 // per window (6P)^2, upper triangle valid
  double* bs;                // [NFP*6]
  double* xp;                // [NFP*6]
  double* chi_part;          // [n_chunks * kResidualSplit]
  double* scale_part;        // [n_chunks]
  double* dmax_pose;         // [NFP]
  int* n_active;             // [1]
  // outputs
  double* out_chi2;          // [NE] caller order
  unsigned char* out_depth;  // [NE]
};

// Read-only snapshot of the controller fields a kernel needs, taken once at kernel entry (a reference into
// global memory would be re-read after every store: the compiler cannot prove the stores do not alias it).
struct LmView { double lambda; int active, need_lin, sel, solve_ok, last_eval_sel, iterations; };
__device__ __forceinline__ LmView lm_view(const LmState* lm, int w) {
  const LmState& s = lm[w];
  return LmView{s.lambda, s.active, s.need_lin, s.sel, s.solve_ok, s.last_eval_sel, s.iterations};
}

// --------------------------------------------------------------------------------------------
// block-wide deterministic reductions (4 wavefronts of 64)
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* sh4) {
  v = dev::wave_sum(v);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh4[wv] = v;
  __syncthreads();
  const double r = (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ double block_max(double v, double* sh4) {
  v = dev::wave_max(v);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh4[wv] = v;
  __syncthreads();
  const double r = fmax(fmax(sh4[0], sh4[1]), fmax(sh4[2], sh4[3]));
  __syncthreads();
  return r;
}

// Ordering point for the single-wavefront kernels (64-thread blocks): LDS operations of one wavefront execute in order, so
// only the compiler has to be told not to move LDS accesses across this point.  __syncthreads() would additionally wait for
// every outstanding GLOBAL access of the wavefront (it is also a workgroup-scope memory fence): a full store round trip per
// chunk in k_lin_items (the Hpl stores); k_schur_items has no global store in its chunk loop at all (exact s_waitcnt counts).
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Residual / Jacobians of one visual edge through the window's camera model.  KB8 is a compile-time switch: the pinhole
// instantiations of the kernels (every batch without a fisheye window) carry no KannalaBrandt8 code or registers.
template <bool KB8>
__device__ __forceinline__ double win_edge_residual(const WinDesc& wd, int kind, const double* qt, const double* cam, const double* X,
                                                    const double* obs, double info, double* r, double* Xc) {
  if (KB8 && wd.kb8_on && kind == OSH_EDGE_MONO) return dev::edge_residual_kb8(qt, cam, wd.kb8, X, obs, info, r, Xc);
  return dev::edge_residual(kind, qt, cam, X, obs, info, r, Xc);
}
template <bool KB8>
__device__ __forceinline__ void win_edge_jacobians(const WinDesc& wd, int kind, const double* R, const double* cam, const double* Xc,
                                                   double* JX, double* Jp) {
  if (KB8 && wd.kb8_on && kind == OSH_EDGE_MONO) { dev::edge_jacobians_kb8(R, cam, wd.kb8, Xc, JX, Jp); return; }
  dev::edge_jacobians(kind, R, cam, Xc, JX, Jp);
}

// --------------------------------------------------------------------------------------------
// k_residual: computeActiveErrors + activeRobustChi2 at the TRIAL estimates (every active window),
// one block per chunk of consecutive landmarks, lane per edge; chunk partial sums in fixed order.
// --------------------------------------------------------------------------------------------
template <bool KB8>
__global__ __launch_bounds__(kBlock) void k_residual(BatchView bv) {
  __shared__ double sh4[4];
  // kResidualSplit blocks per chunk (a chunk holds up to kChunkMaxEdges edges for k_backsub's sake): block q takes the passes
  // q, q + kResidualSplit, ... of the chunk, so every edge still has its own lane in the common case
  const Chunk ch = bv.chunks[blockIdx.x / kResidualSplit];
  const int q = blockIdx.x % kResidualSplit;
  const WinDesc wd = bv.win[ch.win];   // by value: the fields stay in SGPRs across the kernel's stores
  const LmView st = lm_view(bv.lm, ch.win);
  if (!st.active) return;
  const int sel = st.sel ^ 1;
  const double* poses = bv.pose_state[sel] + (size_t)wd.pose_off * 7;
  const double* pts = bv.pt_state[sel] + (size_t)wd.pt_off * 3;
  const double* cams = bv.pose_cam + (size_t)wd.pose_off * 5;
  const int* lmo = bv.lm_off + wd.lmoff_off;
  const int e0 = lmo[ch.lm0], e1 = lmo[ch.lm1];
  double chi_acc = 0.0;
  for (int e = e0 + q * kBlock + threadIdx.x; e < e1; e += kResidualSplit * kBlock) {
    const size_t ge = (size_t)wd.edge_off + e;
    const int ip = bv.e_pose[ge], il = bv.e_point[ge];
    const int kind = bv.e_kind[ge];
    double qt[7], cam[5], X[3], obs[3], r[3], Xc[3];
#pragma unroll
    for (int k = 0; k < 7; ++k) qt[k] = poses[(size_t)ip * 7 + k];
#pragma unroll
    for (int k = 0; k < 5; ++k) cam[k] = cams[(size_t)ip * 5 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { X[k] = pts[(size_t)il * 3 + k]; obs[k] = bv.e_rec[ge * 4 + k]; }
    const double chi2 = win_edge_residual<KB8>(wd, kind, qt, cam, X, obs, bv.e_rec[ge * 4 + 3], r, Xc);
    double rho0, rho1;
    dev::huber(chi2, kind == OSH_EDGE_MONO ? wd.huber_mono : wd.huber_stereo, rho0, rho1);
    chi_acc += rho0;
  }
  const double chi = block_sum(chi_acc, sh4);
  if (threadIdx.x == 0) bv.chi_part[blockIdx.x] = chi;
}

// --------------------------------------------------------------------------------------------
// k_schur_items: the landmark products of the Schur complement (block_solver.hpp:381-432) on the
// FP64 matrix cores, one wavefront per ITEM of schur_plan.h (landmarks that share their set of
// optimisable observers).  Per chunk of 8 landmarks lane (l, s) loads the Hpl block W of landmark
// l / row pose s, forms Dinv = (Hll + lambda I)^-1 (setLambda + Matrix3d::inverse, :389,582-587),
// BD = W Dinv (:403) and W (Dinv b_l) (the _coefficients term, :404-409) and stores BD and W as
// [row][k] images in LDS (row = 6 s + r, k = 3 l + m; row stride 25 doubles: conflict-free for the
// MFMA reads).  D = BD * W^T is then 6 k-steps of v_mfma_f64_16x16x4_f64 per 16x16 tile; the
// accumulators stay in registers across the chunks of the item.  At the end every live pose pair
// (sa, sb) is written as one 6x6 contribution; k_schur_reduce sums them in plan order.
//   SYM: X == Y, only the tiles on and above the diagonal; also owns the rhs term.
// --------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kSiLm = 8;               // landmarks per chunk
constexpr int kSiKS = 3 * kSiLm + 1;   // LDS row stride (doubles)
constexpr int kSiRows = 6 * kItemPoses;

template <bool SYM>
__global__ __launch_bounds__(64, 2) void k_schur_items(BatchView bv, int item_base) {
  __shared__ double shA[kSiRows * kSiKS];
  __shared__ double shB[kSiRows * kSiKS];
  __shared__ int shP[64 + 8];
  const int item_idx = item_base + blockIdx.x;
  const SItem it = bv.sitems[item_idx];
  const WinDesc wd = bv.win[it.win];   // by value: the fields stay in SGPRs across the kernel's stores
  const LmView st = lm_view(bv.lm, it.win);
  if (!st.active) return;
  const int lane = threadIdx.x;
  const int l = lane >> 3, s = lane & 7;
  const int nx = it.shape & 0xff, ny = (it.shape >> 8) & 0xff;
  const int TX = (6 * nx + 15) >> 4, TY = (6 * ny + 15) >> 4;
  const double lambda = st.lambda;
  shP[lane] = bv.spair[(size_t)item_idx * 64 + lane];
  if (lane < 8) shP[64 + lane] = bv.scslot[(size_t)item_idx * 8 + lane];
  const SRec* __restrict__ recs = bv.srecs + it.rec_off;
  const double* __restrict__ Hpl = bv.Hpl + (size_t)wd.edge_off * 18;
  f64x4 acc[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
  double csum[6] = {0, 0, 0, 0, 0, 0};
  const int mrow = lane & 15, mk = lane >> 4;

  // Software pipeline over the chunks: the blocks of chunk c+1 are requested right after chunk c's operands are parked in LDS,
  // so they are in flight during chunk c's MFMA phase (which needs few registers), and the record of chunk c+2 with them.
  // Every load is unconditional on a clamped index (a load inside a divergent branch is waited for at the end of the branch,
  // which used to expose three dependent global round trips per chunk); validity is applied when the values are used.
  struct Dat { double2 w[9]; double2 y[9]; double hl[6], bl[3]; };
  const int last_rec = it.n_lm - 1;
  const int last_edge = wd.E - 1;
  auto load_rec = [&](int c0, int4& ra, int4& rb) {
    const int4* src = reinterpret_cast<const int4*>(recs + min(c0 + l, last_rec));
    ra = src[0]; rb = src[1];          // {lm, e_first, x_lo, x_hi} {y_lo, y_hi, flags, pad}
  };
  auto slot_of = [&](unsigned lo, unsigned hi) { return (((s < 4) ? lo : hi) >> (8 * (s & 3))) & 0xffu; };
  auto load_dat = [&](const int4& ra, const int4& rb, Dat& d) {
    const unsigned xo = slot_of((unsigned)ra.z, (unsigned)ra.w);
    const double2* src = reinterpret_cast<const double2*>(Hpl + (size_t)min(ra.y + (xo != kAbsent ? (int)xo : 0), last_edge) * 18);
#pragma unroll
    for (int k = 0; k < 9; ++k) d.w[k] = src[k];
    if (!SYM) {
      const unsigned yo = slot_of((unsigned)rb.x, (unsigned)rb.y);
      const double2* sy = reinterpret_cast<const double2*>(Hpl + (size_t)min(ra.y + (yo != kAbsent ? (int)yo : 0), last_edge) * 18);
#pragma unroll
      for (int k = 0; k < 9; ++k) d.y[k] = sy[k];
    }
    const size_t gl = (size_t)wd.pt_off + ra.x;
#pragma unroll
    for (int k = 0; k < 6; ++k) d.hl[k] = bv.Hll[gl * 6 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) d.bl[k] = bv.bl[gl * 3 + k];
  };
  auto park = [&](int c0, const int4& ra, const int4& rb, const Dat& d) {
    const bool valid = (c0 + l) < it.n_lm;
    const unsigned xo = valid ? slot_of((unsigned)ra.z, (unsigned)ra.w) : kAbsent;
    double W[18];
#pragma unroll
    for (int k = 0; k < 9; ++k) { W[2 * k] = (xo != kAbsent) ? d.w[k].x : 0.0; W[2 * k + 1] = (xo != kAbsent) ? d.w[k].y : 0.0; }
    double D[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) D[k] = 0.0;
    if (valid) {
      double Dinv[9];
      dev::inv3_sym(d.hl[0] + lambda, d.hl[1], d.hl[2], d.hl[3] + lambda, d.hl[4], d.hl[5] + lambda, Dinv);
      const double b0 = d.bl[0], b1 = d.bl[1], b2 = d.bl[2];
      D[0] = Dinv[0]; D[1] = Dinv[1]; D[2] = Dinv[2]; D[3] = Dinv[4]; D[4] = Dinv[5]; D[5] = Dinv[8];
      D[6] = Dinv[0] * b0 + Dinv[1] * b1 + Dinv[2] * b2;
      D[7] = Dinv[3] * b0 + Dinv[4] * b1 + Dinv[5] * b2;
      D[8] = Dinv[6] * b0 + Dinv[7] * b1 + Dinv[8] * b2;
    }
    wave_sync();   // the previous chunk's MFMA reads are done
    {
      double* a_row = shA + (6 * s) * kSiKS + 3 * l;
      double* b_row = shB + (6 * s) * kSiKS + 3 * l;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double x0 = W[r * 3], x1 = W[r * 3 + 1], x2 = W[r * 3 + 2];
        a_row[r * kSiKS + 0] = x0 * D[0] + x1 * D[1] + x2 * D[2];
        a_row[r * kSiKS + 1] = x0 * D[1] + x1 * D[3] + x2 * D[4];
        a_row[r * kSiKS + 2] = x0 * D[2] + x1 * D[4] + x2 * D[5];
        if (SYM) {
          csum[r] += x0 * D[6] + x1 * D[7] + x2 * D[8];
          b_row[r * kSiKS + 0] = x0; b_row[r * kSiKS + 1] = x1; b_row[r * kSiKS + 2] = x2;
        }
      }
      if (!SYM) {
        const unsigned yo = valid ? slot_of((unsigned)rb.x, (unsigned)rb.y) : kAbsent;
        double Y[18];
#pragma unroll
        for (int k = 0; k < 9; ++k) { Y[2 * k] = (yo != kAbsent) ? d.y[k].x : 0.0; Y[2 * k + 1] = (yo != kAbsent) ? d.y[k].y : 0.0; }
#pragma unroll
        for (int r = 0; r < 6; ++r) { b_row[r * kSiKS + 0] = Y[r * 3]; b_row[r * kSiKS + 1] = Y[r * 3 + 1]; b_row[r * kSiKS + 2] = Y[r * 3 + 2]; }
      }
    }
  };
  // The tile counts are compile-time constants inside each instantiation (dispatched once per chunk): with run-time tile tests
  // every MFMA and every operand read sat behind its own scalar branch (127 branches per chunk against 36 MFMAs).
  auto multiply_t = [&](auto txc, auto tyc) {
    constexpr int TXc = decltype(txc)::value, TYc = decltype(tyc)::value;
#pragma unroll
    for (int ks = 0; ks < (3 * kSiLm) / 4; ++ks) {
      double a[3], b[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        a[t] = (t < TXc) ? shA[(16 * t + mrow) * kSiKS + 4 * ks + mk] : 0.0;
        b[t] = (t < TYc) ? shB[(16 * t + mrow) * kSiKS + 4 * ks + mk] : 0.0;
      }
#pragma unroll
      for (int ti = 0; ti < 3; ++ti)
#pragma unroll
        for (int tj = SYM ? ti : 0; tj < 3; ++tj)
          if (ti < TXc && tj < TYc) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
      if (ks & 1) __builtin_amdgcn_sched_barrier(0);   // operands of two k-steps in flight at most (the prefetched chunk needs the registers)
    }
  };
  using std::integral_constant;
  auto multiply = [&]() {
    if (SYM) {   // TX == TY
      if (TX == 3) multiply_t(integral_constant<int, 3>{}, integral_constant<int, 3>{});
      else if (TX == 2) multiply_t(integral_constant<int, 2>{}, integral_constant<int, 2>{});
      else multiply_t(integral_constant<int, 1>{}, integral_constant<int, 1>{});
    } else {
      // cross items keep run-time tile tests: nine instantiations cost them 50 more registers (spills) for no gain
#pragma unroll
      for (int ks = 0; ks < (3 * kSiLm) / 4; ++ks) {
        double a[3], b[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          a[t] = (t < TX) ? shA[(16 * t + mrow) * kSiKS + 4 * ks + mk] : 0.0;
          b[t] = (t < TY) ? shB[(16 * t + mrow) * kSiKS + 4 * ks + mk] : 0.0;
        }
#pragma unroll
        for (int ti = 0; ti < 3; ++ti)
#pragma unroll
          for (int tj = 0; tj < 3; ++tj)
            if (ti < TX && tj < TY) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
      }
    }
  };
  int4 rc0, rc1, rn0, rn1;
  Dat d;
  load_rec(0, rc0, rc1);
  load_rec(kSiLm, rn0, rn1);
  load_dat(rc0, rc1, d);
  for (int c0 = 0; c0 < it.n_lm; c0 += kSiLm) {
    park(c0, rc0, rc1, d);                 // Dinv, BD, W of chunk c0 -> LDS (consumes d)
    wave_sync();
    rc0 = rn0; rc1 = rn1;                  // loaded one iteration ago
    load_dat(rc0, rc1, d);                 // chunk c0 + 1: in flight during the MFMAs below
    load_rec(c0 + 2 * kSiLm, rn0, rn1);
    multiply();
  }
  wave_sync();
  // contributions: lane holds D[row = (lane >> 4) + 4 reg][col = lane & 15] of each tile
#pragma unroll
  for (int ti = 0; ti < 3; ++ti)
#pragma unroll
    for (int tj = SYM ? ti : 0; tj < 3; ++tj) {
      if (ti < TX && tj < TY) {
        const int C = 16 * tj + (lane & 15);
        const int sb = C / 6, cc = C - 6 * sb;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int R = 16 * ti + (lane >> 4) + 4 * reg;
          const int sa = R / 6, rr = R - 6 * sa;
          const int slot = shP[sa * 8 + sb];
          if (slot >= 0) bv.contrib[(size_t)slot * 36 + rr * 6 + cc] = acc[ti][tj][reg];
        }
      }
    }
  if (SYM) {
    // rhs term of row pose s: sum over the 8 landmark lanes in fixed order
#pragma unroll
    for (int r = 0; r < 6; ++r) shA[l * kSiRows + 6 * s + r] = csum[r];
    wave_sync();
    if (lane < kSiRows) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < kSiLm; ++k) v += shA[k * kSiRows + lane];
      const int sa = lane / 6;
      const int slot = shP[64 + sa];
      if (slot >= 0) bv.ccontrib[(size_t)slot * 6 + (lane - 6 * sa)] = v;
    }
  }
}

// --------------------------------------------------------------------------------------------
// k_schur_reduce: S(i,j) = [i == j] (Hpp_i + lambda I) - sum of the block's contributions,
// b_s(i) = b_p(i) - sum of the pose's rhs contributions; 7 blocks of S per 256-thread group,
// one thread per entry, contributions in plan order.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_schur_reduce(BatchView bv) {
  const int t = threadIdx.x / 36, el = threadIdx.x - 36 * t;
  const int idx = blockIdx.x * 7 + t;
  if (t >= 7 || idx >= bv.n_rblk) return;
  const RBlk rb = bv.rblk[idx];
  const WinDesc wd = bv.win[rb.win];   // by value: the fields stay in SGPRs across the kernel's stores
  const LmView st = lm_view(bv.lm, rb.win);
  if (!st.active) return;
  const int i = rb.ij & 0xffff, j = (rb.ij >> 16) & 0xffff;
  if (j == 0xffff) {
    if (el >= 6) return;
    const size_t gp = (size_t)wd.fpose_off + i;
    double v = bv.bp[gp * 6 + el];
    const double* c = bv.ccontrib + (size_t)rb.start * 6 + el;
    for (int k = 0; k < rb.count; ++k) v -= c[(size_t)k * 6];
    bv.bs[gp * 6 + el] = v;
    return;
  }
  const int r = el / 6, cc = el - 6 * r;
  double v = 0.0;
  if (i == j) v = bv.Hpp[((size_t)wd.fpose_off + i) * 36 + el] + ((r == cc) ? st.lambda : 0.0);
  const double* c = bv.contrib + (size_t)rb.start * 36 + el;
  for (int k = 0; k < rb.count; ++k) v -= c[(size_t)k * 36];
  bv.S[wd.S_off + (size_t)(6 * i + r) * wd.n + 6 * j + cc] = v;
}

// --------------------------------------------------------------------------------------------
// k_lin_items: linearisation at the current estimates (computeActiveErrors + activeRobustChi2 +
// buildSystem, g2o/core/sparse_optimizer.cpp:61-114, block_solver.hpp:502-560,
// base_binary_edge.hpp:55-120), one wavefront per SYMMETRIC item of the Schur plan.  Lane (l, s)
// owns the edge of landmark l (8 per chunk) and row pose X[s]: residual, Huber weight, both
// Jacobians, then
//   Hpl block  -> global (used by every trial of the iteration and by the back-substitution)
//   Hpp / b_p  -> summed over the item's landmarks in registers, one 27-double contribution per
//                 (item, pose), reduced in plan order by k_pose_reduce
//   Hll / b_l  -> summed over the 8 pose lanes of the landmark by a fixed butterfly and written by
//                 the record that owns the landmark (part 0).  The landmark's remaining edges
//                 (observers beyond the first 8, fixed keyframes) are added by k_lin_aux.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ double group8_sum(double v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  return v;
}

template <int SIDE, bool KB8>   // SIDE 0: landmark side (Hpl, Hll, b_l, chi2)   1: pose side (Hpp, b_p partials)
__global__ __launch_bounds__(64, 2) void k_lin_items(BatchView bv) {
  __shared__ double shStage[64 * 18];   // Hpl blocks of one chunk; also the scratch of the final Hpp reduction
  __shared__ double shPose[SIDE == 0 ? 21 * 64 : 1];   // landmark side, per lane: quaternion + translation (7), camera (5), rotation matrix (9)
  __shared__ int2 shRun[8];
  const int item_idx = blockIdx.x;
  const SItem it = bv.sitems[item_idx];
  const WinDesc wd = bv.win[it.win];   // by value: the fields stay in SGPRs across the kernel's stores
  const LmView st = lm_view(bv.lm, it.win);
  if (!st.active || !st.need_lin) return;
  const int lane = threadIdx.x;
  const int l = lane >> 3, s = lane & 7;
  const int nx = it.shape & 0xff;
  const double* poses = bv.pose_state[st.sel] + (size_t)wd.pose_off * 7;
  const double* pts = bv.pt_state[st.sel] + (size_t)wd.pt_off * 3;
  const double* cams = bv.pose_cam + (size_t)wd.pose_off * 5;
  const SRec* __restrict__ recs = bv.srecs + it.rec_off;
  double pose_reg[21];   // pose side: the lane's pose stays in registers
  {
    double qt[7], cam[5], Rm[9];
#pragma unroll
    for (int k = 0; k < 7; ++k) qt[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k) cam[k] = 0.0;
    if (s < nx) {
      const int ip = bv.sposex[(size_t)item_idx * 8 + s];
#pragma unroll
      for (int k = 0; k < 7; ++k) qt[k] = poses[(size_t)ip * 7 + k];
#pragma unroll
      for (int k = 0; k < 5; ++k) cam[k] = cams[(size_t)ip * 5 + k];
    }
    dev::quat_to_R(qt, Rm);
#pragma unroll
    for (int k = 0; k < 7; ++k) pose_reg[k] = qt[k];
#pragma unroll
    for (int k = 0; k < 5; ++k) pose_reg[7 + k] = cam[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) pose_reg[12 + k] = Rm[k];
    if constexpr (SIDE == 0) {
#pragma unroll
      for (int k = 0; k < 21; ++k) shPose[k * 64 + lane] = pose_reg[k];
    }
  }
  double H[21], b[6];
#pragma unroll
  for (int k = 0; k < 21; ++k) H[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) b[k] = 0.0;
  double chi_acc = 0.0, dmax = 0.0;

  // Two-stage software pipeline, unrolled by two so that no register is copied between iterations (a copy
  // would force the wait for the prefetch at the end of every iteration): while chunk c is computed the
  // record of chunk c+2 and the edge data of chunk c+1 are in flight.  All loads are unconditional on
  // clamped indices (a load inside a divergent branch is waited for at the end of the branch); validity is
  // applied when the values are used.
  struct Cur { int lm, e_first, flags, ne, r0, np; unsigned xo; };
  struct In { int kind; double info, X[3], obs[3]; };
  const int last_rec = it.n_lm - 1;
  const size_t last_edge = (size_t)wd.edge_off + (size_t)(wd.E - 1);
  auto load_rec = [&](int c0, int4& ra, int4& rb) {
    const int4* src = reinterpret_cast<const int4*>(recs + min(c0 + l, last_rec));
    ra = src[0]; rb = src[1];          // {lm, e_first, x_lo, x_hi} {y_lo, y_hi, flags, pad}
  };
  auto decode = [&](int c0, const int4& ra, const int4& rb) {
    Cur cu;
    const bool valid = (c0 + l) < it.n_lm;
    cu.lm = ra.x; cu.e_first = ra.y; cu.flags = valid ? rb.z : 0; cu.ne = rb.w;
    cu.r0 = (rb.z >> 16) & 0xff; cu.np = valid ? ((rb.z >> 24) & 0xff) : 0;
    cu.xo = valid ? ((((s < 4) ? (unsigned)ra.z : (unsigned)ra.w) >> (8 * (s & 3))) & 0xffu) : kAbsent;
    return cu;
  };
  auto load_in = [&](const Cur& cu, In& in) {
    const size_t ge = min((size_t)wd.edge_off + cu.e_first + (cu.xo != kAbsent ? (int)cu.xo : 0), last_edge);
    in.kind = bv.e_kind[ge];
    in.info = bv.e_rec[ge * 4 + 3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { in.X[k] = pts[(size_t)cu.lm * 3 + k]; in.obs[k] = bv.e_rec[ge * 4 + k]; }
  };
  auto process = [&](const Cur& cur, const In& inp) {
    const bool owner = (cur.flags & 1) != 0;
    const unsigned xo = cur.xo;
    const bool present = xo != kAbsent;
    double hl[9], JX[9], Jp[18], wr[3], ww = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) { hl[k] = 0.0; JX[k] = 0.0; }
#pragma unroll
    for (int k = 0; k < 18; ++k) Jp[k] = 0.0;
    wr[0] = wr[1] = wr[2] = 0.0;
    if (present) {
      // the lane's pose (quaternion + translation, camera, rotation matrix) lives in LDS between chunks
      double qt[7], cam[5], Rm[9];
#pragma unroll
      for (int k = 0; k < 7; ++k) qt[k] = (SIDE == 0) ? shPose[k * 64 + lane] : pose_reg[k];
#pragma unroll
      for (int k = 0; k < 5; ++k) cam[k] = (SIDE == 0) ? shPose[(7 + k) * 64 + lane] : pose_reg[7 + k];
#pragma unroll
      for (int k = 0; k < 9; ++k) Rm[k] = (SIDE == 0) ? shPose[(12 + k) * 64 + lane] : pose_reg[12 + k];
      const double X[3] = {inp.X[0], inp.X[1], inp.X[2]};
      const int kind = inp.kind;
      const double info = inp.info;
      double r[3], Xc[3];
      const double obs[3] = {inp.obs[0], inp.obs[1], inp.obs[2]};
      const double chi2 = win_edge_residual<KB8>(wd, kind, qt, cam, X, obs, info, r, Xc);
      double rho0, rho1;
      dev::huber(chi2, kind == OSH_EDGE_MONO ? wd.huber_mono : wd.huber_stereo, rho0, rho1);
      win_edge_jacobians<KB8>(wd, kind, Rm, cam, Xc, JX, Jp);
      ww = rho1 * info;                       // robustInformation (first order only)
      wr[0] = -(info * r[0]) * rho1; wr[1] = -(info * r[1]) * rho1; wr[2] = -(info * r[2]) * rho1;
      if (SIDE == 0 && owner) {
        chi_acc += rho0;
        double AtW[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int k = 0; k < 3; ++k) AtW[i * 3 + k] = JX[k * 3 + i] * ww;
        hl[0] = AtW[0] * JX[0] + AtW[1] * JX[3] + AtW[2] * JX[6];
        hl[1] = AtW[0] * JX[1] + AtW[1] * JX[4] + AtW[2] * JX[7];
        hl[2] = AtW[0] * JX[2] + AtW[1] * JX[5] + AtW[2] * JX[8];
        hl[3] = AtW[3] * JX[1] + AtW[4] * JX[4] + AtW[5] * JX[7];
        hl[4] = AtW[3] * JX[2] + AtW[4] * JX[5] + AtW[5] * JX[8];
        hl[5] = AtW[6] * JX[2] + AtW[7] * JX[5] + AtW[8] * JX[8];
        hl[6] = JX[0] * wr[0] + JX[3] * wr[1] + JX[6] * wr[2];
        hl[7] = JX[1] * wr[0] + JX[4] * wr[1] + JX[7] * wr[2];
        hl[8] = JX[2] * wr[0] + JX[5] * wr[1] + JX[8] * wr[2];
      }
    }
    if constexpr (SIDE == 1) {
      if (present) {
        // Hpp += Jp^T W Jp (upper), b_p += Jp^T (-rho' Omega r)
        int m = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const double b0 = Jp[i] * ww, b1 = Jp[6 + i] * ww, b2 = Jp[12 + i] * ww;
#pragma unroll
          for (int c = i; c < 6; ++c) { H[m] += b0 * Jp[c] + b1 * Jp[6 + c] + b2 * Jp[12 + c]; ++m; }
          b[i] += Jp[i] * wr[0] + Jp[6 + i] * wr[1] + Jp[12 + i] * wr[2];
        }
      }
      return;
    }
    if (s == 0) shRun[l] = make_int2(cur.e_first + cur.r0, cur.np * 9);   // the landmark's run of Hpl blocks, in 16-byte units
#pragma unroll
    for (int k = 0; k < 9; ++k) hl[k] = group8_sum(hl[k]);
    if (owner && s == 0) {
      const size_t gl = (size_t)wd.pt_off + cur.lm;
#pragma unroll
      for (int k = 0; k < 6; ++k) bv.Hll[gl * 6 + k] = hl[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) bv.bl[gl * 3 + k] = hl[6 + k];
      // a landmark with further edges (k_lin_aux adds them) reports its diagonal there
      if (cur.ne == ((cur.flags >> 8) & 0xff)) dmax = fmax(dmax, fmax(fabs(hl[0]), fmax(fabs(hl[3]), fabs(hl[5]))));
    }
    if (present) {
      // Hpl block = Jp^T W JX (6x3); the block is staged in LDS, compacted by rank: the blocks of one landmark form
      // one contiguous run in global memory
      double2* stg = reinterpret_cast<double2*>(shStage) + (l * 8 + (int)xo - cur.r0) * 9;
#pragma unroll
      for (int i = 0; i < 6; i += 2) {
        double hp[6];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const double b0 = Jp[i + ii] * ww, b1 = Jp[6 + i + ii] * ww, b2 = Jp[12 + i + ii] * ww;
#pragma unroll
          for (int j = 0; j < 3; ++j) hp[ii * 3 + j] = b0 * JX[j] + b1 * JX[3 + j] + b2 * JX[6 + j];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) stg[(i / 2) * 3 + k] = make_double2(hp[2 * k], hp[2 * k + 1]);
      }
    }
    // coalesced copy of the staged blocks (a strided 16-byte store per lane costs a partial-line write each)
    wave_sync();
    {
      const double2* stg = reinterpret_cast<const double2*>(shStage);
#pragma unroll
      for (int u0 = 0; u0 < 8 * 72; u0 += 64) {
        const int u = u0 + lane;
        const int lu = u / 72, ku = u - lu * 72;
        const int2 run = shRun[lu];
        if (ku < run.y) {
          // non-temporal: the blocks are next read by another kernel, after ~5 GB of other traffic (measured 2.31 -> 2.02 ms)
          typedef double f64x2 __attribute__((ext_vector_type(2)));
          const double2 val = stg[lu * 72 + ku];
          __builtin_nontemporal_store((f64x2){val.x, val.y}, reinterpret_cast<f64x2*>(bv.Hpl + ((size_t)wd.edge_off + run.x) * 18) + ku);
        }
      }
    }
    wave_sync();
  };
  int4 rA0, rA1, rB0, rB1;
  In inA, inB;
  load_rec(0, rA0, rA1);
  load_rec(kSiLm, rB0, rB1);
  load_in(decode(0, rA0, rA1), inA);
  for (int c0 = 0; c0 < it.n_lm; c0 += 2 * kSiLm) {
    {
      const Cur cur = decode(c0, rA0, rA1);
      load_rec(c0 + 2 * kSiLm, rA0, rA1);
      load_in(decode(c0 + kSiLm, rB0, rB1), inB);
      process(cur, inA);
    }
    if (c0 + kSiLm < it.n_lm) {
      const Cur cur = decode(c0 + kSiLm, rB0, rB1);
      load_rec(c0 + 3 * kSiLm, rB0, rB1);
      load_in(decode(c0 + 2 * kSiLm, rA0, rA1), inA);
      process(cur, inB);
    }
  }
  if constexpr (SIDE == 0) {
    chi_acc = dev::wave_sum(chi_acc);
    dmax = dev::wave_max(dmax);
    if (lane == 0) { bv.chi_item[item_idx] = chi_acc; bv.dmax_item[item_idx] = dmax; }
    return;
  }
  // Hpp / b_p of row pose s: sum over the 8 landmark lanes in fixed order (two halves through the staging buffer)
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int k0 = half * 14, nk = half ? 13 : 14;
    wave_sync();
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      const int kk = k0 + k;
      if (k < nk) shStage[k * 64 + lane] = (kk < 21) ? H[kk < 21 ? kk : 0] : b[(kk - 21) < 0 ? 0 : ((kk - 21) > 5 ? 5 : (kk - 21))];
    }
    wave_sync();
    for (int o = lane; o < 8 * nk; o += 64) {
      const int k = o >> 3, sa = o & 7;
      const int slot = bv.scslot[(size_t)item_idx * 8 + sa];
      if (slot >= 0) {
        double v = 0.0;
#pragma unroll
        for (int ll = 0; ll < 8; ++ll) v += shStage[k * 64 + ll * 8 + sa];
        bv.hcontrib[(size_t)slot * 27 + k0 + k] = v;
      }
    }
  }
}

// k_lin_aux: the edges k_lin_items does not cover for the landmark side -- fixed-keyframe edges and the
// optimisable observers beyond a landmark's first 8 (their Hpl / Hpp part is done by the item of
// their own part): residual, Huber weight, d err / d point; Hll += J^T W J, b_l += ..., chi2.
// One lane per edge, edges sorted by landmark, a landmark never straddles two wavefronts (a chunk
// longer than 64 edges holds a single landmark); the first lane of each landmark sums its run in
// lane order and is the only writer of that landmark.  Runs after k_lin_items.
template <bool KB8>
__global__ __launch_bounds__(64) void k_lin_aux(BatchView bv) {
  __shared__ double shv[9 * 64];
  __shared__ int shl[64];
  const int4 ch = bv.aux_chunks[blockIdx.x];   // {window, first entry, entries, 0}
  const WinDesc wd = bv.win[ch.x];   // by value: the fields stay in SGPRs across the kernel's stores
  const LmView st = lm_view(bv.lm, ch.x);
  if (!st.active || !st.need_lin) return;
  const int lane = threadIdx.x;
  const double* poses = bv.pose_state[st.sel] + (size_t)wd.pose_off * 7;
  const double* pts = bv.pt_state[st.sel] + (size_t)wd.pt_off * 3;
  const double* cams = bv.pose_cam + (size_t)wd.pose_off * 5;
  double hl[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) hl[k] = 0.0;
  double chi_acc = 0.0;
  int my_lm = -1;
  for (int x = lane; x < ch.z; x += 64) {
    const int2 en = bv.aux_entries[(size_t)ch.y + x];   // {landmark, sorted edge}
    my_lm = en.x;
    const size_t ge = (size_t)wd.edge_off + en.y;
    const int ip = bv.e_pose[ge];
    const int kind = bv.e_kind[ge];
    const double info = bv.e_rec[ge * 4 + 3];
    double qt[7], cam[5], X[3], obs[3], r[3], Xc[3];
#pragma unroll
    for (int k = 0; k < 7; ++k) qt[k] = poses[(size_t)ip * 7 + k];
#pragma unroll
    for (int k = 0; k < 5; ++k) cam[k] = cams[(size_t)ip * 5 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { X[k] = pts[(size_t)en.x * 3 + k]; obs[k] = bv.e_rec[ge * 4 + k]; }
    const double chi2 = win_edge_residual<KB8>(wd, kind, qt, cam, X, obs, info, r, Xc);
    double rho0, rho1;
    dev::huber(chi2, kind == OSH_EDGE_MONO ? wd.huber_mono : wd.huber_stereo, rho0, rho1);
    chi_acc += rho0;
    double JX[9], Jp[18], R[9];
    dev::quat_to_R(qt, R);
    win_edge_jacobians<KB8>(wd, kind, R, cam, Xc, JX, Jp);
    const double ww = rho1 * info;
    const double wr[3] = {-(info * r[0]) * rho1, -(info * r[1]) * rho1, -(info * r[2]) * rho1};
    double AtW[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) AtW[i * 3 + k] = JX[k * 3 + i] * ww;
    hl[0] += AtW[0] * JX[0] + AtW[1] * JX[3] + AtW[2] * JX[6];
    hl[1] += AtW[0] * JX[1] + AtW[1] * JX[4] + AtW[2] * JX[7];
    hl[2] += AtW[0] * JX[2] + AtW[1] * JX[5] + AtW[2] * JX[8];
    hl[3] += AtW[3] * JX[1] + AtW[4] * JX[4] + AtW[5] * JX[7];
    hl[4] += AtW[3] * JX[2] + AtW[4] * JX[5] + AtW[5] * JX[8];
    hl[5] += AtW[6] * JX[2] + AtW[7] * JX[5] + AtW[8] * JX[8];
    hl[6] += JX[0] * wr[0] + JX[3] * wr[1] + JX[6] * wr[2];
    hl[7] += JX[1] * wr[0] + JX[4] * wr[1] + JX[7] * wr[2];
    hl[8] += JX[2] * wr[0] + JX[5] * wr[1] + JX[8] * wr[2];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) shv[k * 64 + lane] = hl[k];
  shl[lane] = my_lm;
  wave_sync();
  double dmax = 0.0;
  if (my_lm >= 0 && (lane == 0 || shl[lane - 1] != my_lm)) {
    for (int y = lane + 1; y < 64 && shl[y] == my_lm; ++y) {
#pragma unroll
      for (int k = 0; k < 9; ++k) hl[k] += shv[k * 64 + y];
    }
    const size_t gl = (size_t)wd.pt_off + my_lm;
#pragma unroll
    for (int k = 0; k < 6; ++k) { hl[k] += bv.Hll[gl * 6 + k]; bv.Hll[gl * 6 + k] = hl[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) bv.bl[gl * 3 + k] += hl[6 + k];
    dmax = fmax(fabs(hl[0]), fmax(fabs(hl[3]), fabs(hl[5])));
  }
  chi_acc = dev::wave_sum(chi_acc);
  dmax = dev::wave_max(dmax);
  if (lane == 0) { bv.chi_aux[blockIdx.x] = chi_acc; bv.dmax_aux[blockIdx.x] = dmax; }
}

// k_pose_reduce: Hpp_i (full symmetric 6x6), b_p(i) and the largest diagonal entry of pose i from the item
// contributions, in plan order.  32 lanes per optimisable pose: lane k < 27 sums entry k of the contributions
// (one contribution = 27 contiguous doubles -> coalesced, the loads of successive contributions are independent).
__global__ __launch_bounds__(64) void k_pose_reduce(BatchView bv) {
  const int gp = blockIdx.x * 2 + (threadIdx.x >> 5);
  const int k = threadIdx.x & 31;
  if (gp >= bv.n_fposes) return;
  const int w = bv.fpose_win[gp];
  const LmView st = lm_view(bv.lm, w);
  if (!st.active || !st.need_lin) return;
  const int2 rg = bv.pose_crange[gp];
  double a = 0.0;
  if (k < 27) {
    const double* src = bv.hcontrib + (size_t)rg.x * 27 + k;
    int c = 0;
    for (; c + 4 <= rg.y; c += 4) {
      const double v0 = src[(size_t)c * 27], v1 = src[(size_t)(c + 1) * 27], v2 = src[(size_t)(c + 2) * 27], v3 = src[(size_t)(c + 3) * 27];
      a += v0; a += v1; a += v2; a += v3;
    }
    for (; c < rg.y; ++c) a += src[(size_t)c * 27];
    if (k < 21) {
      // entry k of the upper triangle -> (r, cc)
      int r = 0, rem = k;
      while (rem >= 6 - r) { rem -= 6 - r; ++r; }
      const int cc = r + rem;
      double* Ho = bv.Hpp + (size_t)gp * 36;
      Ho[r * 6 + cc] = a;
      Ho[cc * 6 + r] = a;
    } else {
      bv.bp[(size_t)gp * 6 + (k - 21)] = a;
    }
  }
  // largest diagonal entry: upper-triangle entries 0, 6, 11, 15, 18, 20 are the diagonal
  __shared__ double shd[64];
  shd[threadIdx.x] = fabs(a);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (k == 0) {
    const double* d = shd + (threadIdx.x & 32);
    bv.dmax_pose[gp] = fmax(fmax(fmax(d[0], d[6]), fmax(d[11], d[15])), fmax(d[18], d[20]));
  }
}

// --------------------------------------------------------------------------------------------
// k_solve: dense blocked LDL^T (no pivoting, upper storage) of the reduced camera system of
// one window per block, forward elimination fused (rhs carried as an extra column), blocked
// back-substitution, then the pose update T <- exp(x) T and the pose part of computeScale.
// LDS: U panel [nb][W] (unscaled rows), L panel [nb][W] (rows / pivot), x [n], d [nb].
// --------------------------------------------------------------------------------------------
template <int NB, int NT>
__global__ __launch_bounds__(NT) void k_solve(BatchView bv, int W) {
  constexpr int kSolveThreads = NT;
  extern __shared__ __attribute__((aligned(16))) double sh[];   // all LDS scratch is dynamic (16-B aligned base)
  const int w = blockIdx.x;
  const WinDesc& wd = bv.win[w];
  LmState& st = bv.lm[w];
  if (!st.active) return;
  const int n = wd.n;
  const int tid = threadIdx.x;
  double* A = bv.S + wd.S_off;
  double* rhs = bv.bs + (size_t)wd.fpose_off * 6;
  double *xs, *shw;
  const bool ok = ldlt_solve_block<NB, kSolveThreads>(A, rhs, n, W, sh, xs, shw);
  double* xp = bv.xp + (size_t)wd.fpose_off * 6;
  for (int k = tid; k < n; k += kSolveThreads) xp[k] = xs[k];   // zeros when the factorisation failed
  __syncthreads();
  // pose update into the trial buffer + pose part of computeScale
  const int cur = st.sel, tr_sel = st.sel ^ 1;
  const double lambda = st.lambda;
  double sc = 0.0;
  for (int i = tid; i < wd.P; i += kSolveThreads) {
    double u[6], qin[7], qout[7];
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = xs[6 * i + k];
#pragma unroll
    for (int k = 0; k < 7; ++k) qin[k] = bv.pose_state[cur][((size_t)wd.pose_off + i) * 7 + k];
    dev::pose_oplus(u, qin, qout);
#pragma unroll
    for (int k = 0; k < 7; ++k) bv.pose_state[tr_sel][((size_t)wd.pose_off + i) * 7 + k] = qout[k];
    const double* bpi = bv.bp + ((size_t)wd.fpose_off + i) * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) sc += u[k] * (lambda * u[k] + bpi[k]);
  }
  sc = dev::wave_sum(sc);
  if ((tid & 63) == 0) shw[tid >> 6] = sc;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int k = 0; k < kSolveThreads / 64; ++k) tot += shw[k];
    st.scale_pose = tot; st.solve_ok = ok ? 1 : 0;
  }
}

// --------------------------------------------------------------------------------------------
// k_backsub: x_l = Dinv (b_l - Hpl^T x_p), X_trial = X + x_l, landmark part of computeScale.
// Same chunking as k_residual: lane per edge for the Hpl^T x_p products, lane per landmark
// for the ordered sum.  The product of an edge is formed from its Jacobians, Hpl^T x = JX^T (rho' Omega) (Jp x), re-evaluated at
// the linearisation point from ~45 bytes of edge data instead of reading the 144-byte block the linearisation stored (the
// kernel is HBM-bound: 5.9 GB -> see DESIGN.md); the optimisable poses of the window sit in LDS with their rotation matrices.
// --------------------------------------------------------------------------------------------
constexpr int kBsPoseStride = 17;   // qt(7) + R(9), odd stride against LDS bank conflicts
template <bool KB8>
__global__ __launch_bounds__(kBlock) void k_backsub(BatchView bv) {
  extern __shared__ __attribute__((aligned(16))) double sh_bs[];  // [3*256] partials, [4] reduce, [n] x_p, [P*17] poses
  double* sh_c = sh_bs;
  double* sh4 = sh_bs + 3 * kChunkEdges;
  double* sh_dyn = sh4 + 4;
  const Chunk ch = bv.chunks[blockIdx.x];
  const WinDesc wd = bv.win[ch.win];   // by value: the fields stay in SGPRs across the kernel's stores
  const LmView st = lm_view(bv.lm, ch.win);
  if (!st.active) return;
  const int tid = threadIdx.x;
  const int n = wd.n;
  const double* xp = bv.xp + (size_t)wd.fpose_off * 6;
  for (int k = tid; k < n; k += kBlock) sh_dyn[k] = xp[k];
  double* sh_pose = sh_dyn + n;
  {
    const double* poses = bv.pose_state[st.sel] + (size_t)wd.pose_off * 7;
    for (int i = tid; i < wd.P; i += kBlock) {
      double qt[7], R[9];
#pragma unroll
      for (int k = 0; k < 7; ++k) qt[k] = poses[(size_t)i * 7 + k];
      dev::quat_to_R(qt, R);
#pragma unroll
      for (int k = 0; k < 7; ++k) sh_pose[i * kBsPoseStride + k] = qt[k];
#pragma unroll
      for (int k = 0; k < 9; ++k) sh_pose[i * kBsPoseStride + 7 + k] = R[k];
    }
  }
  __syncthreads();
  const int* lmo = bv.lm_off + wd.lmoff_off;
  const int e0 = lmo[ch.lm0], e1 = lmo[ch.lm1];
  const int nl = ch.lm1 - ch.lm0;
  const double lambda = st.lambda;
  const double* pts = bv.pt_state[st.sel] + (size_t)wd.pt_off * 3;
  const double* cams = bv.pose_cam + (size_t)wd.pose_off * 5;
  double acc[3] = {0, 0, 0};
  int my_lo = 0, my_hi = 0;
  if (tid < nl) { my_lo = lmo[ch.lm0 + tid]; my_hi = lmo[ch.lm0 + tid + 1]; }
  for (int base = e0; base < e1; base += kChunkEdges) {
    const int e = base + tid;
    double c0 = 0, c1 = 0, c2 = 0;
    if (e < e1) {
      const size_t ge = (size_t)wd.edge_off + e;
      const int ip = bv.e_pose[ge];
      if (ip < wd.P) {
        const int il = bv.e_point[ge];
        const int kind = bv.e_kind[ge];
        const double info = bv.e_rec[ge * 4 + 3];
        double qt[7], R[9], cam[5], X[3], obs[3], r[3], Xc[3], JX[9], Jp[18];
#pragma unroll
        for (int k = 0; k < 7; ++k) qt[k] = sh_pose[ip * kBsPoseStride + k];
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = sh_pose[ip * kBsPoseStride + 7 + k];
#pragma unroll
        for (int k = 0; k < 5; ++k) cam[k] = cams[(size_t)ip * 5 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { X[k] = pts[(size_t)il * 3 + k]; obs[k] = bv.e_rec[ge * 4 + k]; }
        const double chi2 = win_edge_residual<KB8>(wd, kind, qt, cam, X, obs, info, r, Xc);
        double rho0, rho1;
        dev::huber(chi2, kind == OSH_EDGE_MONO ? wd.huber_mono : wd.huber_stereo, rho0, rho1);
        win_edge_jacobians<KB8>(wd, kind, R, cam, Xc, JX, Jp);
        const double ww = rho1 * info;
        const double* x = sh_dyn + 6 * ip;
        double t[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          double a = 0.0;
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) a += Jp[k * 6 + rr] * (-x[rr]);   // rightMultiply with cp = -xp
          t[k] = ww * a;
        }
        c0 = JX[0] * t[0] + JX[3] * t[1] + JX[6] * t[2];
        c1 = JX[1] * t[0] + JX[4] * t[1] + JX[7] * t[2];
        c2 = JX[2] * t[0] + JX[5] * t[1] + JX[8] * t[2];
      }
    }
    sh_c[tid] = c0; sh_c[kChunkEdges + tid] = c1; sh_c[2 * kChunkEdges + tid] = c2;
    __syncthreads();
    if (tid < nl) {
      const int lo = max(my_lo, base), hi = min(my_hi, base + kChunkEdges);
      for (int x = lo; x < hi; ++x) {
        acc[0] += sh_c[x - base]; acc[1] += sh_c[kChunkEdges + x - base]; acc[2] += sh_c[2 * kChunkEdges + x - base];
      }
    }
    __syncthreads();
  }
  double sc = 0.0;
  if (tid < nl) {
    const size_t gl = (size_t)wd.pt_off + ch.lm0 + tid;
    // (Hll + lambda I)^-1 is formed again here (the same inv3_sym on the same inputs as in k_schur_items: identical bits) rather
    // than stored there and re-read: a global store inside the Schur kernel's chunk loop makes its s_waitcnt counts inexact
    double Dinv[9];
    dev::inv3_sym(bv.Hll[gl * 6] + lambda, bv.Hll[gl * 6 + 1], bv.Hll[gl * 6 + 2], bv.Hll[gl * 6 + 3] + lambda, bv.Hll[gl * 6 + 4],
                  bv.Hll[gl * 6 + 5] + lambda, Dinv);
    const double D[6] = {Dinv[0], Dinv[1], Dinv[2], Dinv[4], Dinv[5], Dinv[8]};   // sym 00 01 02 11 12 22
    const double b0 = bv.bl[gl * 3], b1 = bv.bl[gl * 3 + 1], b2 = bv.bl[gl * 3 + 2];
    const double c0 = b0 + acc[0], c1 = b1 + acc[1], c2 = b2 + acc[2];
    double xl[3];
    if (st.solve_ok) {
      xl[0] = D[0] * c0 + D[1] * c1 + D[2] * c2;
      xl[1] = D[1] * c0 + D[3] * c1 + D[4] * c2;
      xl[2] = D[2] * c0 + D[4] * c1 + D[5] * c2;
    } else {
      xl[0] = xl[1] = xl[2] = 0.0;
    }
    const double* Xc = bv.pt_state[st.sel] + gl * 3;
    double* Xt = bv.pt_state[st.sel ^ 1] + gl * 3;
    Xt[0] = Xc[0] + xl[0]; Xt[1] = Xc[1] + xl[1]; Xt[2] = Xc[2] + xl[2];
    sc = xl[0] * (lambda * xl[0] + b0) + xl[1] * (lambda * xl[1] + b1) + xl[2] * (lambda * xl[2] + b2);
  }
  sc = block_sum(sc, sh4);
  if (tid == 0) bv.scale_part[blockIdx.x] = sc;
}

// --------------------------------------------------------------------------------------------
// k_control: one wavefront per window.
//   phase 0 (after linearise): open the iteration (currentChi, lambda init).
//   phase 1 (after the trial residual): gain ratio, accept / reject, stop rules.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_control(BatchView bv, int phase) {
  const int w = blockIdx.x;
  const WinDesc& wd = bv.win[w];
  LmState& st = bv.lm[w];
  const int lane = threadIdx.x;
  if (!st.active) return;
  // robust chi2 of the state evaluated last = sum of the partials (fixed order): item partials after a
  // linearisation, chunk partials after a trial residual
  double chi = 0.0;
  if (phase == 0) {
    for (int c = lane; c < wd.n_sitems; c += 64) chi += bv.chi_item[wd.sitem_off + c];
    for (int c = lane; c < wd.n_aux; c += 64) chi += bv.chi_aux[wd.aux_off + c];
  }
  else { for (int c = lane; c < wd.n_chunks * kResidualSplit; c += 64) chi += bv.chi_part[(size_t)wd.chunk_off * kResidualSplit + c]; }
  chi = dev::wave_sum(chi);
  if (phase == 0) {
    if (!st.need_lin) return;
    double dm = 0.0;
    if (st.iter == 0) {
      for (int c = lane; c < wd.n_sitems; c += 64) dm = fmax(dm, bv.dmax_item[wd.sitem_off + c]);
      for (int c = lane; c < wd.n_aux; c += 64) dm = fmax(dm, bv.dmax_aux[wd.aux_off + c]);
      for (int p = lane; p < wd.P; p += 64) dm = fmax(dm, bv.dmax_pose[wd.fpose_off + p]);
      dm = dev::wave_max(dm);
    }
    if (lane == 0) {
      st.currentChi = chi; st.tempChi = chi; st.iniChi = chi;
      if (st.iter == 0) {
        st.chi2_initial = chi;
        st.lambda = (wd.lambda_init > 0) ? wd.lambda_init : kTau * dm;  // computeLambdaInit
        st.ni = 2.0; st.nBad = 0;
      }
      st.rho = 0.0; st.qmax = 0; st.need_lin = 0;
      st.last_eval_sel = st.sel;
    }
    return;
  }
  // ---- phase 1
  double sc = 0.0;
  for (int c = lane; c < wd.n_chunks; c += 64) sc += bv.scale_part[wd.chunk_off + c];
  sc = dev::wave_sum(sc);
  if (lane != 0) return;
  double tempChi = chi;
  if (!st.solve_ok) tempChi = DBL_MAX;
  st.tempChi = tempChi;
  st.last_eval_sel = st.sel ^ 1;  // computeActiveErrors just ran on the trial estimates
  double rho = st.currentChi - tempChi;
  double scale = st.scale_pose + sc;
  scale += 1e-3;
  rho /= scale;
  if (rho > 0 && isfinite(tempChi)) {
    double alpha = 1. - pow((2 * rho - 1), 3);
    alpha = fmin(alpha, 2. / 3.);
    const double scaleFactor = fmax(1. / 3., alpha);
    st.lambda *= scaleFactor;
    st.ni = 2;
    st.currentChi = tempChi;
    st.sel ^= 1;  // discardTop: the trial estimates become current
  } else {
    st.lambda *= st.ni;
    st.ni *= 2;  // pop: trial buffer is simply abandoned
  }
  st.rho = rho;
  st.qmax++;
  st.trials++;
  const bool again = (rho < 0) && (st.qmax < kMaxTrials) && !st.stop;
  if (!again) {
    // the iteration is over
    st.iterations++;
    if (st.n_trace < OSH_LBA_MAX_TRACE) {
      st.chi2_trace[st.n_trace] = st.currentChi;
      st.lambda_trace[st.n_trace] = st.lambda;
      st.trials_trace[st.n_trace] = st.qmax;
      st.n_trace++;
    }
    bool ok = true;
    if (st.qmax == kMaxTrials || rho == 0) ok = false;  // Terminate
    else {
      if ((st.iniChi - st.currentChi) * 1e3 < st.iniChi) st.nBad++; else st.nBad = 0;  // Raul's stop
      if (st.nBad >= 3) ok = false;
    }
    st.iter++;
    if (!ok || st.iter >= wd.max_iter || st.stop) st.active = 0;
    else st.need_lin = 1;
  }
  if (st.active) atomicAdd(bv.n_active, 1);
}

// reset the controller before optimize()
__global__ void k_reset(BatchView bv, const unsigned char* stop) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= bv.n_windows) return;
  const WinDesc& wd = bv.win[w];
  LmState& st = bv.lm[w];
  st.lambda = -1.0; st.ni = 2.0; st.currentChi = 0; st.iniChi = 0; st.tempChi = 0; st.rho = 0; st.scale_pose = 0;
  st.chi2_initial = 0;
  st.iter = 0; st.qmax = 0; st.nBad = 0;
  st.stop = stop ? stop[w] : 0;
  st.active = (wd.max_iter > 0 && !st.stop && wd.E > 0) ? 1 : 0;
  st.need_lin = st.active;
  st.sel = 0; st.last_eval_sel = 0; st.iterations = 0; st.trials = 0; st.solve_ok = 1; st.n_trace = 0;
  if (st.active) atomicAdd(bv.n_active, 1);
}

__global__ void k_set_stop(BatchView bv, const unsigned char* stop) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= bv.n_windows) return;
  bv.lm[w].stop = stop[w];
}

// --------------------------------------------------------------------------------------------
// k_finalize: per-edge chi2 of the LAST evaluated errors (stale after a rejected final trial,
// levenberg.cpp:123-147) and isDepthPositive from the FINAL estimates (Optimizer.cc:1425).
// --------------------------------------------------------------------------------------------
template <bool KB8>
__global__ __launch_bounds__(kBlock) void k_finalize(BatchView bv) {
  const Chunk ch = bv.chunks[blockIdx.x];
  const WinDesc wd = bv.win[ch.win];   // by value: the fields stay in SGPRs across the kernel's stores
  const LmView st = lm_view(bv.lm, ch.win);
  const int* lmo = bv.lm_off + wd.lmoff_off;
  const int e0 = lmo[ch.lm0], e1 = lmo[ch.lm1];
  const bool evaluated = st.iterations > 0;
  for (int e = e0 + threadIdx.x; e < e1; e += kBlock) {
    const size_t ge = (size_t)wd.edge_off + e;
    const int ip = bv.e_pose[ge], il = bv.e_point[ge];
    const int kind = bv.e_kind[ge];
    double qt[7], cam[5], X[3], obs[3], r[3], Xc[3];
#pragma unroll
    for (int k = 0; k < 5; ++k) cam[k] = bv.pose_cam[((size_t)wd.pose_off + ip) * 5 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) obs[k] = bv.e_rec[ge * 4 + k];
    double chi2 = 0.0;
    if (evaluated) {
      const int s = st.last_eval_sel;
#pragma unroll
      for (int k = 0; k < 7; ++k) qt[k] = bv.pose_state[s][((size_t)wd.pose_off + ip) * 7 + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) X[k] = bv.pt_state[s][((size_t)wd.pt_off + il) * 3 + k];
      chi2 = win_edge_residual<KB8>(wd, kind, qt, cam, X, obs, bv.e_rec[ge * 4 + 3], r, Xc);
    }
    const int f = st.sel;
#pragma unroll
    for (int k = 0; k < 7; ++k) qt[k] = bv.pose_state[f][((size_t)wd.pose_off + ip) * 7 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) X[k] = bv.pt_state[f][((size_t)wd.pt_off + il) * 3 + k];
    double rot[3];
    dev::quat_rotate(qt, X, rot);
    const size_t go = (size_t)wd.edge_off + bv.e_orig[ge];
    bv.out_chi2[go] = chi2;
    bv.out_depth[go] = (rot[2] + qt[6] > 0.0) ? 1 : 0;
  }
}

}  // namespace osh

// =============================================================================================
// Host driver
// =============================================================================================
using namespace osh;

struct osh_lba_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  KernelTimer timer;
  // host-side batch description
  int n_windows = 0;
  std::vector<WinDesc> h_win;
  std::vector<const volatile unsigned char*> stop_ptr;
  bool any_stop = false;
  size_t NP = 0, NFP = 0, NL = 0, NE = 0, NEf = 0, n_chunks = 0;
  size_t n_items = 0, n_sym = 0, n_rblk = 0, n_contrib = 0, n_ccontrib = 0, n_aux_chunks = 0;
  long long plan_tile_steps = 0, plan_pair_blocks = 0;
  size_t S_total = 0;
  int n_max = 0, solve_nb = 24, solve_W = 0, solve_threads = kSolveThreadsBatch;
  size_t solve_lds = 0, backsub_lds = 0;
  // edge kernels of the batch's camera models: the KannalaBrandt8 instantiations only when a window asks for them
  bool has_kb8 = false;
  void (*kp_lin_lm)(BatchView) = nullptr;
  void (*kp_lin_pose)(BatchView) = nullptr;
  void (*kp_lin_aux)(BatchView) = nullptr;
  void (*kp_residual)(BatchView) = nullptr;
  void (*kp_finalize)(BatchView) = nullptr;
  void (*kp_backsub)(BatchView) = nullptr;
  // device buffers
  DevBuf d_win, d_lm, d_chunks, d_fpose_win, d_pose_init, d_pose[2], d_pt_init, d_pt[2], d_cam;
  DevBuf d_e_pose, d_e_point, d_e_kind, d_e_obs, d_e_info, d_e_orig, d_lm_off, d_lm_nfree, d_pel_off, d_pel_edge, d_sitems, d_srecs, d_spair, d_scslot, d_sposex, d_pose_crange, d_hcontrib, d_chi_item, d_dmax_item, d_aux_chunks, d_aux_entries, d_chi_aux, d_dmax_aux, d_rblk, d_contrib, d_ccontrib, d_dinv;
  DevBuf d_Hpl, d_Hll, d_bl, d_Hpp, d_bp, d_S, d_bs, d_xp, d_chi, d_scale, d_dmaxp, d_nactive, d_out_chi2, d_out_depth, d_stop;
  int* h_nactive = nullptr;          // pinned
  unsigned char* h_stop = nullptr;   // pinned [n_windows]
  size_t h_stop_cap = 0;
  std::vector<int> h_e_orig;         // for debug export
  BatchView bv{};
  bool optimized = false;
};

static int launch_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("kernel launch %s failed: %s", what, hipGetErrorString(e)); return OSH_ERR_DEVICE; }
  return OSH_OK;
}

#define OSH_TRY(expr) do { int _rc = (expr); if (_rc != OSH_OK) return _rc; } while (0)

extern "C" int osh_lba_create(int device, osh_lba_ctx** out) {
  if (!out) { set_error("osh_lba_create: out is NULL"); return OSH_ERR_INVALID; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible"); return OSH_ERR_NO_DEVICE; }
  if (device < 0 || device >= n) { set_error("device %d out of range (have %d)", device, n); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(device));
  osh_lba_ctx* c = new osh_lba_ctx();
  c->device = device;
  OSH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  OSH_HIP(hipHostMalloc((void**)&c->h_nactive, sizeof(int)));
  *out = c;
  return OSH_OK;
}

// used by liba_device.hip: the inertial path shares the context's device and stream
extern "C" int osh_lba_stream(osh_lba_ctx* c, int* device, hipStream_t* stream) {
  if (!c) { set_error("null context"); return OSH_ERR_INVALID; }
  *device = c->device; *stream = c->stream;
  return OSH_OK;
}

extern "C" void osh_lba_destroy(osh_lba_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  DevBuf* bufs[] = {&c->d_win, &c->d_lm, &c->d_chunks, &c->d_fpose_win, &c->d_pose_init, &c->d_pose[0], &c->d_pose[1],
                    &c->d_pt_init, &c->d_pt[0], &c->d_pt[1], &c->d_cam, &c->d_e_pose, &c->d_e_point, &c->d_e_kind,
                    &c->d_e_obs, &c->d_e_info, &c->d_e_orig, &c->d_lm_off, &c->d_lm_nfree, &c->d_pel_off, &c->d_pel_edge, &c->d_sitems, &c->d_srecs, &c->d_spair, &c->d_scslot, &c->d_sposex, &c->d_pose_crange, &c->d_hcontrib, &c->d_chi_item, &c->d_dmax_item, &c->d_aux_chunks, &c->d_aux_entries, &c->d_chi_aux, &c->d_dmax_aux, &c->d_rblk, &c->d_contrib, &c->d_ccontrib, &c->d_dinv,
                    &c->d_Hpl, &c->d_Hll, &c->d_bl, &c->d_Hpp, &c->d_bp, &c->d_S, &c->d_bs, &c->d_xp, &c->d_chi,
                    &c->d_scale, &c->d_dmaxp, &c->d_nactive, &c->d_out_chi2, &c->d_out_depth, &c->d_stop};
  for (DevBuf* b : bufs) b->release();
  c->timer.destroy();
  if (c->h_nactive) (void)hipHostFree(c->h_nactive);
  if (c->h_stop) (void)hipHostFree(c->h_stop);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

template <class T>
static int upload_vec(DevBuf& b, const std::vector<T>& v, hipStream_t s) {
  OSH_TRY(b.reserve(std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) OSH_HIP(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return OSH_OK;
}

extern "C" int osh_lba_upload(osh_lba_ctx* c, int32_t nw, const osh_lba_problem* pr) {
  if (!c || nw <= 0 || !pr) { set_error("osh_lba_upload: bad arguments"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  OSH_HIP(hipStreamSynchronize(c->stream));
  c->optimized = false;
  c->n_windows = 0;                  // stays 0 (nothing uploaded) unless this upload completes
  c->h_win.assign(nw, WinDesc{});
  c->stop_ptr.assign(nw, nullptr);
  c->any_stop = false;
  c->has_kb8 = false;
  size_t NP = 0, NFP = 0, NL = 0, NE = 0, NEf = 0, NLO = 0, NPO = 0, S_total = 0;
  int n_max = 0;
  // ---- pass 1: validate + offsets
  for (int w = 0; w < nw; ++w) {
    const osh_lba_problem& p = pr[w];
    if (p.n_free < 0 || p.n_fixed < 0 || p.n_points < 0 || p.n_edges < 0 ||
        (p.n_edges > 0 && (!p.edge_pose || !p.edge_point || !p.edge_kind || !p.edge_obs || !p.edge_info)) ||
        ((p.n_free + p.n_fixed) > 0 && (!p.pose_qt || !p.pose_cam)) || (p.n_points > 0 && !p.points)) {
      set_error("window %d: negative size or NULL array", w);
      return OSH_ERR_INVALID;
    }
    if (p.max_iterations > OSH_LBA_MAX_TRACE) { set_error("window %d: max_iterations > %d", w, OSH_LBA_MAX_TRACE); return OSH_ERR_INVALID; }
    WinDesc& d = c->h_win[w];
    d.P = p.n_free; d.F = p.n_fixed; d.L = p.n_points; d.E = p.n_edges;
    d.pose_off = (int)NP; d.fpose_off = (int)NFP; d.pt_off = (int)NL; d.edge_off = (int)NE;
    d.lmoff_off = (int)NLO; d.peloff_off = (int)NPO; d.pel_off = (int)NEf;
    d.n = 6 * p.n_free; d.max_iter = p.max_iterations; d.S_off = (long long)S_total;
    d.huber_mono = p.huber_mono; d.huber_stereo = p.huber_stereo; d.lambda_init = p.lambda_init;
    d.kb8_on = p.kb8 ? 1 : 0;
    for (int k = 0; k < 4; ++k) d.kb8[k] = p.kb8 ? p.kb8[k] : 0.0;
    if (p.kb8) {
      c->has_kb8 = true;
      for (int e = 0; e < p.n_edges; ++e)
        if (p.edge_kind[e] != OSH_EDGE_MONO) {
          set_error("window %d: a KannalaBrandt8 window takes monocular edges only (edge %d); the right-camera edges of a fisheye "
                    "stereo rig (EdgeSE3ProjectXYZToBody) are not supported by the device path yet", w, e);
          return OSH_ERR_UNSUPPORTED;
        }
    }
    c->stop_ptr[w] = p.stop_flag;
    if (p.stop_flag) c->any_stop = true;
    NP += (size_t)p.n_free + p.n_fixed; NFP += p.n_free; NL += p.n_points; NE += p.n_edges;
    NLO += (size_t)p.n_points + 1; NPO += (size_t)p.n_free + 1;
    S_total += (size_t)d.n * d.n;
    n_max = std::max(n_max, d.n);
    size_t nfree_e = 0;
    for (int e = 0; e < p.n_edges; ++e) {
      const int ip = p.edge_pose[e], il = p.edge_point[e];
      if (ip < 0 || ip >= p.n_free + p.n_fixed || il < 0 || il >= p.n_points || p.edge_kind[e] > OSH_EDGE_STEREO) {
        set_error("window %d edge %d: index or kind out of range", w, e);
        return OSH_ERR_INVALID;
      }
      if (ip < p.n_free) ++nfree_e;
    }
    NEf += nfree_e;
  }
  if (NP > 0x7fffff00u || NL > 0x7fffff00u || NE > 0x7fffff00u) { set_error("batch too large for 32-bit offsets"); return OSH_ERR_UNSUPPORTED; }
  c->NP = NP; c->NFP = NFP; c->NL = NL; c->NE = NE; c->NEf = NEf; c->S_total = S_total; c->n_max = n_max;

  // ---- pass 2: build sorted structure
  std::vector<double> h_pose(NP * 7), h_cam(NP * 5), h_pt(NL * 3), h_rec(NE * 4);
  std::vector<int> h_epose(NE), h_epoint(NE), h_eorig(NE), h_lmoff(NLO), h_lmnfree(NL), h_peloff(NPO), h_pel(NEf), h_fpw(NFP);
  std::vector<unsigned char> h_kind(NE);
  std::vector<Chunk> h_chunks;
  SchurPlan plan;
  std::vector<int4> h_aux_chunks;
  std::vector<int2> h_aux_entries;
  std::vector<plan_detail::Build> builds;
  std::vector<int> build_win;
  // The windows are independent: every window is packed by one host thread into its own slices of the flat arrays (offsets
  // from pass 1) and into window-local lists, which are merged in window order afterwards (contribution slots rebased).
  struct WinLocal {
    std::vector<Chunk> chunks;
    std::vector<int4> aux_chunks;
    std::vector<int2> aux_entries;
    std::vector<plan_detail::Build> builds;
    SchurPlan plan;
    int err = OSH_OK;
    char msg[320];
  };
  std::vector<WinLocal> locals(nw);
  auto pack_window = [&](int w, std::vector<int>& cnt, std::vector<int>& fill, std::vector<int>& order) {
    WinLocal& L = locals[w];
    const osh_lba_problem& p = pr[w];
    WinDesc& d = c->h_win[w];
    const int NPw = p.n_free + p.n_fixed;
    for (int i = 0; i < NPw; ++i) {
      double q[4] = {p.pose_qt[7 * i], p.pose_qt[7 * i + 1], p.pose_qt[7 * i + 2], p.pose_qt[7 * i + 3]};
      // g2o::SE3Quat(q,t) constructor: normalizeRotation (se3quat.h:61-63,280-285)
      if (q[3] < 0) { q[0] *= -1; q[1] *= -1; q[2] *= -1; q[3] *= -1; }
      const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      double* o = &h_pose[((size_t)d.pose_off + i) * 7];
      for (int k = 0; k < 4; ++k) o[k] = q[k] / nrm;
      for (int k = 0; k < 3; ++k) o[4 + k] = p.pose_qt[7 * i + 4 + k];
      for (int k = 0; k < 5; ++k) h_cam[((size_t)d.pose_off + i) * 5 + k] = p.pose_cam[5 * i + k];
    }
    if (p.n_points) std::memcpy(&h_pt[(size_t)d.pt_off * 3], p.points, sizeof(double) * 3 * p.n_points);
    for (int i = 0; i < p.n_free; ++i) h_fpw[(size_t)d.fpose_off + i] = w;
    // counting sort by landmark (stable), then order poses inside each landmark
    cnt.assign((size_t)p.n_points + 1, 0);
    for (int e = 0; e < p.n_edges; ++e) cnt[p.edge_point[e] + 1]++;
    for (int j = 0; j < p.n_points; ++j) cnt[j + 1] += cnt[j];
    fill.assign(cnt.begin(), cnt.end() - 1);
    order.resize(p.n_edges);
    for (int e = 0; e < p.n_edges; ++e) order[fill[p.edge_point[e]]++] = e;
    int* lmo = &h_lmoff[d.lmoff_off];
    for (int j = 0; j <= p.n_points; ++j) lmo[j] = cnt[j];
    for (int j = 0; j < p.n_points; ++j) {
      int lo = cnt[j], hi = cnt[j + 1];
      // stable insertion sort by pose: tracks are short (~8 edges) and std::stable_sort allocates a buffer per call
      for (int x = lo + 1; x < hi; ++x) {
        const int e = order[x], pe = p.edge_pose[e];
        int y = x;
        for (; y > lo && p.edge_pose[order[y - 1]] > pe; --y) order[y] = order[y - 1];
        order[y] = e;
      }
      int nf = 0;
      for (int x = lo; x < hi; ++x) {
        if (p.edge_pose[order[x]] < p.n_free) ++nf;
        if (x > lo && p.edge_pose[order[x]] == p.edge_pose[order[x - 1]]) {
          std::snprintf(L.msg, sizeof(L.msg), "window %d: landmark %d is observed twice by pose %d (two edges on one Hessian block); "
                        "not supported by the device path yet", w, j, p.edge_pose[order[x]]);
          L.err = OSH_ERR_UNSUPPORTED;
          return;
        }
      }
      h_lmnfree[(size_t)d.pt_off + j] = nf;
    }
    for (int x = 0; x < p.n_edges; ++x) {
      const int e = order[x];
      const size_t g = (size_t)d.edge_off + x;
      h_epose[g] = p.edge_pose[e]; h_epoint[g] = p.edge_point[e]; h_kind[g] = p.edge_kind[e]; h_eorig[g] = e;
      h_rec[g * 4 + 3] = p.edge_info[e];
      for (int k = 0; k < 3; ++k) h_rec[g * 4 + k] = p.edge_obs[3 * e + k];
    }
    // per-pose edge lists (landmark order)
    int* po = &h_peloff[d.peloff_off];
    for (int i = 0; i <= p.n_free; ++i) po[i] = 0;
    for (int x = 0; x < p.n_edges; ++x) { const int ip = h_epose[(size_t)d.edge_off + x]; if (ip < p.n_free) po[ip + 1]++; }
    for (int i = 0; i < p.n_free; ++i) po[i + 1] += po[i];
    fill.assign(po, po + p.n_free);
    for (int x = 0; x < p.n_edges; ++x) {
      const int ip = h_epose[(size_t)d.edge_off + x];
      if (ip < p.n_free) h_pel[(size_t)d.pel_off + fill[ip]++] = x;
    }
    // Schur work plan: landmarks grouped by observer set (schur_plan.h)
    {
      if (!plan_window(w, p.n_free, p.n_points, lmo, &h_lmnfree[(size_t)d.pt_off], &h_epose[(size_t)d.edge_off], L.builds, L.plan)) {
        std::snprintf(L.msg, sizeof(L.msg), "window %d: a landmark has more than 254 optimisable observers", w);
        L.err = OSH_ERR_UNSUPPORTED;
        return;
      }
    }
    // k_lin_aux work list: per landmark the edges beyond its first 8 optimisable observers (fixed keyframes included)
    {
      int first = 0, n = 0;   // entry indices are window-local here and rebased when the windows are merged
      auto close = [&]() { if (n > 0) L.aux_chunks.push_back(make_int4(w, first, n, 0)); first = (int)L.aux_entries.size(); n = 0; };
      for (int j = 0; j < p.n_points; ++j) {
        const int n0 = std::min(h_lmnfree[(size_t)d.pt_off + j], kItemPoses);
        const int cnt_j = (lmo[j + 1] - lmo[j]) - n0;
        if (cnt_j <= 0) continue;
        if (n + cnt_j > 64) close();
        for (int x = lmo[j] + n0; x < lmo[j + 1]; ++x) L.aux_entries.push_back(make_int2(j, x));
        n += cnt_j;
        if (n >= 64) close();
      }
      close();
    }
    // chunks: consecutive landmarks, <= kChunkMaxEdges edges and <= kBlock landmarks (a single
    // landmark with more edges gets its own multi-pass chunk)
    int j = 0;
    while (j < p.n_points) {
      int j1 = j + 1;
      while (j1 < p.n_points && (j1 - j) < kBlock && (lmo[j1 + 1] - lmo[j]) <= kChunkMaxEdges) ++j1;
      L.chunks.push_back(Chunk{w, j, j1});
      j = j1;
    }
    };
  {
    int n_threads = 1;
    if (nw > 1) {
      const char* env = std::getenv("ORBSLAM3_HIP_UPLOAD_THREADS");
      const unsigned hw = std::thread::hardware_concurrency();
      n_threads = env ? std::atoi(env) : (int)std::min<unsigned>(hw ? hw : 1u, 16u);
      n_threads = std::max(1, std::min(n_threads, nw));
    }
    std::atomic<int> next{0};
    auto worker = [&]() {
      std::vector<int> cnt, fill, order;
      for (int w = next.fetch_add(1); w < nw; w = next.fetch_add(1)) pack_window(w, cnt, fill, order);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (std::thread& t : pool) t.join();
  }
  for (int w = 0; w < nw; ++w) {
    WinLocal& L = locals[w];
    if (L.err != OSH_OK) { set_error("%s", L.msg); return L.err; }
    WinDesc& d = c->h_win[w];
    d.chunk_off = (int)h_chunks.size();
    h_chunks.insert(h_chunks.end(), L.chunks.begin(), L.chunks.end());
    d.n_chunks = (int)L.chunks.size();
    d.aux_off = (int)h_aux_chunks.size();
    const int base_e = (int)h_aux_entries.size();
    for (int4 ch : L.aux_chunks) { ch.y += base_e; h_aux_chunks.push_back(ch); }
    h_aux_entries.insert(h_aux_entries.end(), L.aux_entries.begin(), L.aux_entries.end());
    d.n_aux = (int)L.aux_chunks.size();
    const int base_c = (int)plan.n_contrib, base_cc = (int)plan.n_ccontrib;
    for (RBlk rb : L.plan.rblk) { rb.start += (((rb.ij >> 16) & 0xffff) == 0xffff) ? base_cc : base_c; plan.rblk.push_back(rb); }
    for (plan_detail::Build& bd : L.builds) {
      for (int k = 0; k < 64; ++k) if (bd.pair_slot[k] >= 0) bd.pair_slot[k] += base_c;
      for (int k = 0; k < 8; ++k) if (bd.c_slot[k] >= 0) bd.c_slot[k] += base_cc;
      builds.push_back(std::move(bd));
      build_win.push_back(w);
    }
    plan.n_contrib += L.plan.n_contrib; plan.n_ccontrib += L.plan.n_ccontrib;
    plan.tile_steps += L.plan.tile_steps; plan.pair_blocks += L.plan.pair_blocks;
    L = WinLocal();
  }
  c->n_chunks = h_chunks.size();
  finish_plan(build_win, builds, plan);
  if (plan.n_contrib > 0x7fffff00u / 36 * 16 || plan.recs.size() > 0x7fffff00u) { set_error("batch too large for 32-bit contribution offsets"); return OSH_ERR_UNSUPPORTED; }
  for (WinDesc& d : c->h_win) { d.sitem_off = 0; d.n_sitems = 0; }
  for (int x = 0; x < plan.n_sym; ++x) {
    WinDesc& d = c->h_win[plan.items[x].win];
    if (d.n_sitems == 0) d.sitem_off = x;
    d.n_sitems++;
  }
  std::vector<int2> h_crange;
  h_crange.reserve(NFP);
  for (const RBlk& rb : plan.rblk) if (((rb.ij >> 16) & 0xffff) == 0xffff) h_crange.push_back(make_int2(rb.start, rb.count));
  c->n_items = plan.items.size(); c->n_sym = (size_t)plan.n_sym; c->n_rblk = plan.rblk.size();
  c->n_contrib = plan.n_contrib; c->n_ccontrib = plan.n_ccontrib;
  c->plan_tile_steps = plan.tile_steps; c->plan_pair_blocks = plan.pair_blocks;
  c->h_e_orig = h_eorig;

  // ---- LDS budgets
  c->backsub_lds = (size_t)(3 * kChunkEdges + 4 + std::max(n_max, 1) + (std::max(n_max, 6) / 6) * kBsPoseStride) * sizeof(double);
  {
    // One 512-thread block per window with the widest panel that fits (one block per CU).  Two 256-thread blocks per CU with
    // 12-wide panels used to win for batches (their pivot-chain latencies overlapped); since the factorisation was rewritten the
    // wide panel is faster there too (halves the trailing-update traffic: 0.80 -> 0.66 ms per 512 windows), so it is used always.
    const bool latency = true;
    c->solve_threads = latency ? kSolveThreadsLatency : kSolveThreadsBatch;
    auto need = [&](int b) { return ldlt_lds_doubles(b, ldlt_row_stride(n_max), c->solve_threads) * sizeof(double); };
    int nb = 0;
    for (size_t budget : {(size_t)75 * 1024, (size_t)150 * 1024}) {
      if (latency && budget < 150 * 1024) continue;
      nb = 24;
      while (nb > 6 && need(nb) > budget) nb /= 2;  // 24 -> 12 -> 6 (template instantiations of k_solve)
      if (need(nb) <= budget && (nb >= 12 || budget > 75 * 1024)) break;
      nb = 0;
    }
    if (nb == 0) {
      set_error("window with %d optimisable poses exceeds the LDS budget of the reduced-system kernels", n_max / 6);
      return OSH_ERR_UNSUPPORTED;
    }
    c->solve_nb = nb;
    c->solve_W = ldlt_row_stride(n_max);
    c->solve_lds = need(nb);
  }

  // ---- device copies
  hipStream_t s = c->stream;
  OSH_TRY(upload_vec(c->d_win, c->h_win, s));
  OSH_TRY(upload_vec(c->d_chunks, h_chunks, s));
  OSH_TRY(upload_vec(c->d_fpose_win, h_fpw, s));
  OSH_TRY(upload_vec(c->d_pose_init, h_pose, s));
  OSH_TRY(upload_vec(c->d_pt_init, h_pt, s));
  OSH_TRY(upload_vec(c->d_cam, h_cam, s));
  OSH_TRY(upload_vec(c->d_e_pose, h_epose, s));
  OSH_TRY(upload_vec(c->d_e_point, h_epoint, s));
  OSH_TRY(upload_vec(c->d_e_kind, h_kind, s));
  OSH_TRY(upload_vec(c->d_e_obs, h_rec, s));
  OSH_TRY(upload_vec(c->d_e_orig, h_eorig, s));
  OSH_TRY(upload_vec(c->d_lm_off, h_lmoff, s));
  OSH_TRY(upload_vec(c->d_lm_nfree, h_lmnfree, s));
  OSH_TRY(upload_vec(c->d_pel_off, h_peloff, s));
  OSH_TRY(upload_vec(c->d_pel_edge, h_pel, s));
  OSH_TRY(upload_vec(c->d_sitems, plan.items, s));
  OSH_TRY(upload_vec(c->d_srecs, plan.recs, s));
  OSH_TRY(upload_vec(c->d_spair, plan.pair_slot, s));
  OSH_TRY(upload_vec(c->d_scslot, plan.c_slot, s));
  OSH_TRY(upload_vec(c->d_rblk, plan.rblk, s));
  OSH_TRY(upload_vec(c->d_sposex, plan.pose_x, s));
  OSH_TRY(upload_vec(c->d_aux_chunks, h_aux_chunks, s));
  OSH_TRY(upload_vec(c->d_aux_entries, h_aux_entries, s));
  c->n_aux_chunks = h_aux_chunks.size();
  OSH_TRY(upload_vec(c->d_pose_crange, h_crange, s));
  auto R = [&](DevBuf& b, size_t bytes) { return b.reserve(std::max<size_t>(bytes, 8)); };
  OSH_TRY(R(c->d_lm, nw * sizeof(LmState)));
  for (int k = 0; k < 2; ++k) { OSH_TRY(R(c->d_pose[k], NP * 7 * 8)); OSH_TRY(R(c->d_pt[k], NL * 3 * 8)); }
  OSH_TRY(R(c->d_dinv, NL * 9 * 8)); OSH_TRY(R(c->d_contrib, plan.n_contrib * 36 * 8)); OSH_TRY(R(c->d_ccontrib, plan.n_ccontrib * 6 * 8));
  OSH_TRY(R(c->d_hcontrib, plan.n_ccontrib * 27 * 8)); OSH_TRY(R(c->d_chi_item, (size_t)plan.n_sym * 8)); OSH_TRY(R(c->d_dmax_item, (size_t)plan.n_sym * 8));
  OSH_TRY(R(c->d_chi_aux, h_aux_chunks.size() * 8)); OSH_TRY(R(c->d_dmax_aux, h_aux_chunks.size() * 8));
  OSH_TRY(R(c->d_Hpl, NE * 18 * 8)); OSH_TRY(R(c->d_Hll, NL * 6 * 8)); OSH_TRY(R(c->d_bl, NL * 3 * 8));
  OSH_TRY(R(c->d_Hpp, NFP * 36 * 8)); OSH_TRY(R(c->d_bp, NFP * 6 * 8)); OSH_TRY(R(c->d_S, S_total * 8));
  OSH_TRY(R(c->d_bs, NFP * 6 * 8)); OSH_TRY(R(c->d_xp, NFP * 6 * 8));
  OSH_TRY(R(c->d_chi, c->n_chunks * kResidualSplit * 8)); OSH_TRY(R(c->d_scale, c->n_chunks * 8));
  OSH_TRY(R(c->d_dmaxp, NFP * 8)); OSH_TRY(R(c->d_nactive, sizeof(int)));
  OSH_TRY(R(c->d_out_chi2, NE * 8)); OSH_TRY(R(c->d_out_depth, NE)); OSH_TRY(R(c->d_stop, nw));
  if (c->h_stop_cap < (size_t)nw) {
    if (c->h_stop) (void)hipHostFree(c->h_stop);
    OSH_HIP(hipHostMalloc((void**)&c->h_stop, nw));
    c->h_stop_cap = nw;
  }
  OSH_HIP(hipMemsetAsync(c->d_xp.p, 0, std::max<size_t>(NFP * 6 * 8, 8), s));
  OSH_HIP(hipMemsetAsync(c->d_Hpl.p, 0, std::max<size_t>(NE * 18 * 8, 8), s));

  BatchView& bv = c->bv;
  bv.n_windows = nw; bv.n_chunks = (int)c->n_chunks; bv.n_fposes = (int)NFP;
  bv.win = c->d_win.as<WinDesc>(); bv.lm = c->d_lm.as<LmState>(); bv.chunks = c->d_chunks.as<Chunk>();
  bv.fpose_win = c->d_fpose_win.as<int>();
  for (int k = 0; k < 2; ++k) { bv.pose_state[k] = c->d_pose[k].as<double>(); bv.pt_state[k] = c->d_pt[k].as<double>(); }
  bv.pose_cam = c->d_cam.as<double>();
  bv.e_pose = c->d_e_pose.as<int>(); bv.e_point = c->d_e_point.as<int>(); bv.e_kind = c->d_e_kind.as<unsigned char>();
  bv.e_rec = c->d_e_obs.as<double>(); bv.e_orig = c->d_e_orig.as<int>();
  bv.lm_off = c->d_lm_off.as<int>(); bv.lm_nfree = c->d_lm_nfree.as<int>();
  bv.pel_off = c->d_pel_off.as<int>(); bv.pel_edge = c->d_pel_edge.as<int>();
  bv.sitems = c->d_sitems.as<SItem>(); bv.srecs = c->d_srecs.as<SRec>(); bv.spair = c->d_spair.as<int>(); bv.scslot = c->d_scslot.as<int>();
  bv.rblk = c->d_rblk.as<RBlk>(); bv.n_rblk = (int)c->n_rblk;
  bv.contrib = c->d_contrib.as<double>(); bv.ccontrib = c->d_ccontrib.as<double>();
  bv.sposex = c->d_sposex.as<int>(); bv.pose_crange = c->d_pose_crange.as<int2>(); bv.hcontrib = c->d_hcontrib.as<double>();
  bv.chi_item = c->d_chi_item.as<double>(); bv.dmax_item = c->d_dmax_item.as<double>();
  bv.aux_chunks = c->d_aux_chunks.as<int4>(); bv.aux_entries = c->d_aux_entries.as<int2>();
  bv.chi_aux = c->d_chi_aux.as<double>(); bv.dmax_aux = c->d_dmax_aux.as<double>();
  bv.dinv = c->d_dinv.as<double>();
  bv.Hpl = c->d_Hpl.as<double>(); bv.Hll = c->d_Hll.as<double>(); bv.bl = c->d_bl.as<double>();
  bv.Hpp = c->d_Hpp.as<double>(); bv.bp = c->d_bp.as<double>(); bv.S = c->d_S.as<double>();
  bv.bs = c->d_bs.as<double>(); bv.xp = c->d_xp.as<double>();
  bv.chi_part = c->d_chi.as<double>(); bv.scale_part = c->d_scale.as<double>();
  bv.dmax_pose = c->d_dmaxp.as<double>();
  bv.n_active = c->d_nactive.as<int>();
  bv.out_chi2 = c->d_out_chi2.as<double>(); bv.out_depth = c->d_out_depth.as<unsigned char>();
  OSH_HIP(hipStreamSynchronize(s));

  if (c->has_kb8) {
    c->kp_lin_lm = k_lin_items<0, true>; c->kp_lin_pose = k_lin_items<1, true>; c->kp_lin_aux = k_lin_aux<true>;
    c->kp_residual = k_residual<true>; c->kp_finalize = k_finalize<true>; c->kp_backsub = k_backsub<true>;
  } else {
    c->kp_lin_lm = k_lin_items<0, false>; c->kp_lin_pose = k_lin_items<1, false>; c->kp_lin_aux = k_lin_aux<false>;
    c->kp_residual = k_residual<false>; c->kp_finalize = k_finalize<false>; c->kp_backsub = k_backsub<false>;
  }

  // opt in to large dynamic LDS: the attribute is per device, so once per device of the process
  static std::mutex attr_mu;
  static std::vector<int> attr_devices;
  std::lock_guard<std::mutex> attr_lock(attr_mu);
  if (std::find(attr_devices.begin(), attr_devices.end(), c->device) == attr_devices.end()) {
#define OSH_SOLVE_ATTR(NB, NT) OSH_HIP(hipFuncSetAttribute((const void*)k_solve<NB, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64))
    OSH_SOLVE_ATTR(24, kSolveThreadsBatch); OSH_SOLVE_ATTR(12, kSolveThreadsBatch); OSH_SOLVE_ATTR(6, kSolveThreadsBatch);
    OSH_SOLVE_ATTR(24, kSolveThreadsLatency); OSH_SOLVE_ATTR(12, kSolveThreadsLatency); OSH_SOLVE_ATTR(6, kSolveThreadsLatency);
#undef OSH_SOLVE_ATTR
    OSH_HIP(hipFuncSetAttribute((const void*)k_backsub<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    OSH_HIP(hipFuncSetAttribute((const void*)k_backsub<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    attr_devices.push_back(c->device);
  }
  c->n_windows = nw;
  return OSH_OK;
}

static bool snapshot_stop(osh_lba_ctx* c) {
  bool any = false;
  for (int w = 0; w < c->n_windows; ++w) {
    const volatile unsigned char* f = c->stop_ptr[w];
    c->h_stop[w] = (f && *f) ? 1 : 0;
    any |= c->h_stop[w] != 0;
  }
  return any;
}

#define LAUNCH(kid, name, grid, block, lds, ...)                                              \
  do {                                                                                        \
    if ((grid) > 0) {                                                                         \
      const bool _t = c->timer.begin(kid, s);                                                 \
      hipLaunchKernelGGL(name, dim3((unsigned)(grid)), dim3(block), (lds), s, __VA_ARGS__);   \
      if (_t) c->timer.end(s);                                                                \
      OSH_TRY(launch_check(#name));                                                           \
    }                                                                                         \
  } while (0)

// the instantiation of k_solve chosen at upload time (panel width x threads per block)
static int launch_solve(osh_lba_ctx* c, hipStream_t s) {
  const dim3 grid((unsigned)c->n_windows);
#define OSH_SOLVE_CASE(NB, NT) \
  if (c->solve_nb == NB && c->solve_threads == NT) hipLaunchKernelGGL((k_solve<NB, NT>), grid, dim3(NT), c->solve_lds, s, c->bv, c->solve_W);
  OSH_SOLVE_CASE(24, kSolveThreadsBatch) OSH_SOLVE_CASE(12, kSolveThreadsBatch) OSH_SOLVE_CASE(6, kSolveThreadsBatch)
  OSH_SOLVE_CASE(24, kSolveThreadsLatency) OSH_SOLVE_CASE(12, kSolveThreadsLatency) OSH_SOLVE_CASE(6, kSolveThreadsLatency)
#undef OSH_SOLVE_CASE
  return launch_check("k_solve");
}

static int reset_state(osh_lba_ctx* c) {
  hipStream_t s = c->stream;
  for (int k = 0; k < 2; ++k) {
    if (c->NP) OSH_HIP(hipMemcpyAsync(c->d_pose[k].p, c->d_pose_init.p, c->NP * 7 * 8, hipMemcpyDeviceToDevice, s));
    if (c->NL) OSH_HIP(hipMemcpyAsync(c->d_pt[k].p, c->d_pt_init.p, c->NL * 3 * 8, hipMemcpyDeviceToDevice, s));
  }
  OSH_HIP(hipMemsetAsync(c->d_nactive.p, 0, sizeof(int), s));
  const unsigned char* dstop = nullptr;
  if (c->any_stop) {
    snapshot_stop(c);
    OSH_HIP(hipMemcpyAsync(c->d_stop.p, c->h_stop, c->n_windows, hipMemcpyHostToDevice, s));
    dstop = c->d_stop.as<unsigned char>();
  }
  hipLaunchKernelGGL(k_reset, dim3((c->n_windows + 63) / 64), dim3(64), 0, s, c->bv, dstop);
  OSH_TRY(launch_check("k_reset"));
  return OSH_OK;
}

static int read_nactive(osh_lba_ctx* c, int* out) {
  OSH_HIP(hipMemcpyAsync(c->h_nactive, c->d_nactive.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  OSH_HIP(hipStreamSynchronize(c->stream));
  *out = *c->h_nactive;
  return OSH_OK;
}

extern "C" int osh_lba_optimize(osh_lba_ctx* c) {
  if (!c || c->n_windows <= 0) { set_error("osh_lba_optimize: nothing uploaded"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  if (c->timer.enabled) OSH_TRY(c->timer.init());
  OSH_TRY(reset_state(c));
  int n_active = 0;
  OSH_TRY(read_nactive(c, &n_active));
  int max_iter = 0;
  for (const WinDesc& d : c->h_win) max_iter = std::max(max_iter, d.max_iter);
  const long max_rounds = (long)max_iter * kMaxTrials + 1;
  for (long round = 0; round < max_rounds && n_active > 0; ++round) {
    LAUNCH(OSH_K_LINEARIZE, c->kp_lin_lm, c->n_sym, 64, 0, c->bv);
    LAUNCH(OSH_K_LIN_POSE, c->kp_lin_pose, c->n_sym, 64, 0, c->bv);
    LAUNCH(OSH_K_LIN_AUX, c->kp_lin_aux, c->n_aux_chunks, 64, 0, c->bv);
    LAUNCH(OSH_K_POSE_HESS, k_pose_reduce, (c->NFP + 1) / 2, 64, 0, c->bv);
    LAUNCH(OSH_K_CONTROL, k_control, c->n_windows, 64, 0, c->bv, 0);
    LAUNCH(OSH_K_SCHUR, k_schur_items<true>, c->n_sym, 64, 0, c->bv, 0);
    LAUNCH(OSH_K_SCHUR_CROSS, k_schur_items<false>, c->n_items - c->n_sym, 64, 0, c->bv, (int)c->n_sym);
    LAUNCH(OSH_K_SCHUR_REDUCE, k_schur_reduce, (c->n_rblk + 6) / 7, 256, 0, c->bv);
    {
      const bool _t = c->timer.begin(OSH_K_SOLVE, s);
      OSH_TRY(launch_solve(c, s));
      if (_t) c->timer.end(s);
    }
    LAUNCH(OSH_K_BACKSUB, c->kp_backsub, c->n_chunks, kBlock, c->backsub_lds, c->bv);
    LAUNCH(OSH_K_RESIDUAL, c->kp_residual, c->n_chunks * kResidualSplit, kBlock, 0, c->bv);
    if (c->any_stop) {
      // terminate() is polled after every trial (levenberg.cpp:149) and before every iteration
      snapshot_stop(c);
      OSH_HIP(hipMemcpyAsync(c->d_stop.p, c->h_stop, c->n_windows, hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(k_set_stop, dim3((c->n_windows + 63) / 64), dim3(64), 0, s, c->bv, c->d_stop.as<unsigned char>());
      OSH_TRY(launch_check("k_set_stop"));
    }
    OSH_HIP(hipMemsetAsync(c->d_nactive.p, 0, sizeof(int), s));
    LAUNCH(OSH_K_CONTROL, k_control, c->n_windows, 64, 0, c->bv, 1);
    OSH_TRY(read_nactive(c, &n_active));
    if (c->timer.enabled) c->timer.collect();
  }
  if (c->n_chunks) {
    hipLaunchKernelGGL(c->kp_finalize, dim3((unsigned)c->n_chunks), dim3(kBlock), 0, s, c->bv);
    OSH_TRY(launch_check("k_finalize"));
  }
  OSH_HIP(hipStreamSynchronize(s));
  if (c->timer.enabled) c->timer.collect();
  c->optimized = true;
  return OSH_OK;
}

extern "C" int osh_lba_download(osh_lba_ctx* c, int32_t nw, osh_lba_result* res) {
  if (!c || !res || nw != c->n_windows) { set_error("osh_lba_download: bad arguments"); return OSH_ERR_INVALID; }
  if (!c->optimized) { set_error("osh_lba_download: call osh_lba_optimize first"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  std::vector<LmState> h_lm(nw);
  OSH_HIP(hipMemcpyAsync(h_lm.data(), c->d_lm.p, nw * sizeof(LmState), hipMemcpyDeviceToHost, s));
  OSH_HIP(hipStreamSynchronize(s));
  for (int w = 0; w < nw; ++w) {
    const WinDesc& d = c->h_win[w];
    const LmState& st = h_lm[w];
    osh_lba_result& r = res[w];
    const int sel = st.sel;
    if (r.pose_qt && d.P) OSH_HIP(hipMemcpyAsync(r.pose_qt, c->d_pose[sel].as<double>() + (size_t)d.pose_off * 7, (size_t)d.P * 7 * 8, hipMemcpyDeviceToHost, s));
    if (r.points && d.L) OSH_HIP(hipMemcpyAsync(r.points, c->d_pt[sel].as<double>() + (size_t)d.pt_off * 3, (size_t)d.L * 3 * 8, hipMemcpyDeviceToHost, s));
    if (r.edge_chi2 && d.E) OSH_HIP(hipMemcpyAsync(r.edge_chi2, c->d_out_chi2.as<double>() + d.edge_off, (size_t)d.E * 8, hipMemcpyDeviceToHost, s));
    if (r.edge_depth_pos && d.E) OSH_HIP(hipMemcpyAsync(r.edge_depth_pos, c->d_out_depth.as<unsigned char>() + d.edge_off, (size_t)d.E, hipMemcpyDeviceToHost, s));
    r.status = OSH_OK; r.iterations = st.iterations; r.trials = st.trials; r.n_trace = st.n_trace;
    r.chi2_initial = st.chi2_initial;
    for (int k = 0; k < st.n_trace; ++k) { r.chi2_trace[k] = st.chi2_trace[k]; r.lambda_trace[k] = st.lambda_trace[k]; r.trials_trace[k] = st.trials_trace[k]; }
  }
  OSH_HIP(hipStreamSynchronize(s));
  return OSH_OK;
}

extern "C" int osh_lba_solve(osh_lba_ctx* c, int32_t nw, const osh_lba_problem* pr, osh_lba_result* res) {
  OSH_TRY(osh_lba_upload(c, nw, pr));
  OSH_TRY(osh_lba_optimize(c));
  return osh_lba_download(c, nw, res);
}

// Debug / parity aid: one linearisation of `window` at the uploaded estimates.
extern "C" int osh_lba_linearize(osh_lba_ctx* c, int32_t window, double* Hpp, double* bp, double* Hll, double* bl,
                                 double* Hpl, double* chi2, double* robust_chi2) {
  if (!c || window < 0 || window >= c->n_windows) { set_error("osh_lba_linearize: bad window"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  OSH_TRY(reset_state(c));
  if (c->n_sym) {
    hipLaunchKernelGGL(c->kp_lin_lm, dim3((unsigned)c->n_sym), dim3(64), 0, s, c->bv); OSH_TRY(launch_check("k_lin_items<0>"));
    hipLaunchKernelGGL(c->kp_lin_pose, dim3((unsigned)c->n_sym), dim3(64), 0, s, c->bv); OSH_TRY(launch_check("k_lin_items<1>"));
  }
  if (c->n_aux_chunks) { hipLaunchKernelGGL(c->kp_lin_aux, dim3((unsigned)c->n_aux_chunks), dim3(64), 0, s, c->bv); OSH_TRY(launch_check("k_lin_aux")); }
  if (c->NFP) { hipLaunchKernelGGL(k_pose_reduce, dim3((unsigned)((c->NFP + 1) / 2)), dim3(64), 0, s, c->bv); OSH_TRY(launch_check("k_pose_reduce")); }
  hipLaunchKernelGGL(k_control, dim3((unsigned)c->n_windows), dim3(64), 0, s, c->bv, 0);
  OSH_TRY(launch_check("k_control"));
  // per-edge chi2 through k_finalize needs "evaluated" semantics: emulate by a residual pass bookkeeping
  OSH_HIP(hipStreamSynchronize(s));
  const WinDesc& d = c->h_win[window];
  std::vector<LmState> h_lm(c->n_windows);
  OSH_HIP(hipMemcpy(h_lm.data(), c->d_lm.p, c->n_windows * sizeof(LmState), hipMemcpyDeviceToHost));
  if (robust_chi2) *robust_chi2 = h_lm[window].chi2_initial;
  if (Hpp && d.P) OSH_HIP(hipMemcpy(Hpp, c->d_Hpp.as<double>() + (size_t)d.fpose_off * 36, (size_t)d.P * 36 * 8, hipMemcpyDeviceToHost));
  if (bp && d.P) OSH_HIP(hipMemcpy(bp, c->d_bp.as<double>() + (size_t)d.fpose_off * 6, (size_t)d.P * 6 * 8, hipMemcpyDeviceToHost));
  if (bl && d.L) OSH_HIP(hipMemcpy(bl, c->d_bl.as<double>() + (size_t)d.pt_off * 3, (size_t)d.L * 3 * 8, hipMemcpyDeviceToHost));
  if (Hll && d.L) {
    std::vector<double> up((size_t)d.L * 6);
    OSH_HIP(hipMemcpy(up.data(), c->d_Hll.as<double>() + (size_t)d.pt_off * 6, up.size() * 8, hipMemcpyDeviceToHost));
    for (int j = 0; j < d.L; ++j) {
      const double* u = &up[(size_t)j * 6];
      double* o = Hll + (size_t)j * 9;
      o[0] = u[0]; o[1] = u[1]; o[2] = u[2]; o[3] = u[1]; o[4] = u[3]; o[5] = u[4]; o[6] = u[2]; o[7] = u[4]; o[8] = u[5];
    }
  }
  if (Hpl && d.E) {
    std::vector<double> hs((size_t)d.E * 18);
    OSH_HIP(hipMemcpy(hs.data(), c->d_Hpl.as<double>() + (size_t)d.edge_off * 18, hs.size() * 8, hipMemcpyDeviceToHost));
    for (int x = 0; x < d.E; ++x) std::memcpy(Hpl + (size_t)c->h_e_orig[(size_t)d.edge_off + x] * 18, &hs[(size_t)x * 18], 18 * 8);
  }
  if (chi2 && d.E) {
    // mark every window evaluated so k_finalize emits chi2 of the current state
    for (auto& st : h_lm) { st.iterations = 1; st.last_eval_sel = st.sel; }
    OSH_HIP(hipMemcpy(c->d_lm.p, h_lm.data(), c->n_windows * sizeof(LmState), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(c->kp_finalize, dim3((unsigned)c->n_chunks), dim3(kBlock), 0, s, c->bv);
    OSH_TRY(launch_check("k_finalize"));
    OSH_HIP(hipStreamSynchronize(s));
    OSH_HIP(hipMemcpy(chi2, c->d_out_chi2.as<double>() + d.edge_off, (size_t)d.E * 8, hipMemcpyDeviceToHost));
  }
  c->optimized = false;
  return OSH_OK;
}

// Debug / parity aid: one LM trial of every window at the uploaded estimates with lambda
// forced to `lambda`; exports S (dense, upper valid), the reduced rhs and x = (x_p, x_l).
extern "C" int osh_lba_debug_trial(osh_lba_ctx* c, int32_t window, double lambda, double* S, double* bs, double* x) {
  if (!c || window < 0 || window >= c->n_windows) { set_error("osh_lba_debug_trial: bad window"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  OSH_TRY(reset_state(c));
  if (c->n_sym) {
    hipLaunchKernelGGL(c->kp_lin_lm, dim3((unsigned)c->n_sym), dim3(64), 0, s, c->bv); OSH_TRY(launch_check("k_lin_items<0>"));
    hipLaunchKernelGGL(c->kp_lin_pose, dim3((unsigned)c->n_sym), dim3(64), 0, s, c->bv); OSH_TRY(launch_check("k_lin_items<1>"));
  }
  if (c->n_aux_chunks) { hipLaunchKernelGGL(c->kp_lin_aux, dim3((unsigned)c->n_aux_chunks), dim3(64), 0, s, c->bv); OSH_TRY(launch_check("k_lin_aux")); }
  if (c->NFP) { hipLaunchKernelGGL(k_pose_reduce, dim3((unsigned)((c->NFP + 1) / 2)), dim3(64), 0, s, c->bv); OSH_TRY(launch_check("k_pose_reduce")); }
  hipLaunchKernelGGL(k_control, dim3((unsigned)c->n_windows), dim3(64), 0, s, c->bv, 0);
  OSH_TRY(launch_check("k_control"));
  OSH_HIP(hipStreamSynchronize(s));
  std::vector<LmState> h_lm(c->n_windows);
  OSH_HIP(hipMemcpy(h_lm.data(), c->d_lm.p, c->n_windows * sizeof(LmState), hipMemcpyDeviceToHost));
  for (auto& st : h_lm) st.lambda = lambda;
  OSH_HIP(hipMemcpy(c->d_lm.p, h_lm.data(), c->n_windows * sizeof(LmState), hipMemcpyHostToDevice));
  const WinDesc& d = c->h_win[window];
  if (c->n_sym) { hipLaunchKernelGGL(k_schur_items<true>, dim3((unsigned)c->n_sym), dim3(64), 0, s, c->bv, 0); OSH_TRY(launch_check("k_schur_items")); }
  if (c->n_items > c->n_sym) { hipLaunchKernelGGL(k_schur_items<false>, dim3((unsigned)(c->n_items - c->n_sym)), dim3(64), 0, s, c->bv, (int)c->n_sym); OSH_TRY(launch_check("k_schur_items")); }
  if (c->n_rblk) { hipLaunchKernelGGL(k_schur_reduce, dim3((unsigned)((c->n_rblk + 6) / 7)), dim3(256), 0, s, c->bv); OSH_TRY(launch_check("k_schur_reduce")); }
  OSH_HIP(hipStreamSynchronize(s));
  if (S && d.n) OSH_HIP(hipMemcpy(S, c->d_S.as<double>() + d.S_off, (size_t)d.n * d.n * 8, hipMemcpyDeviceToHost));
  if (bs && d.n) OSH_HIP(hipMemcpy(bs, c->d_bs.as<double>() + (size_t)d.fpose_off * 6, (size_t)d.n * 8, hipMemcpyDeviceToHost));
  OSH_TRY(launch_solve(c, s));
  hipLaunchKernelGGL(c->kp_backsub, dim3((unsigned)c->n_chunks), dim3(kBlock), c->backsub_lds, s, c->bv);
  OSH_TRY(launch_check("k_backsub"));
  OSH_HIP(hipStreamSynchronize(s));
  if (x) {
    if (d.n) OSH_HIP(hipMemcpy(x, c->d_xp.as<double>() + (size_t)d.fpose_off * 6, (size_t)d.n * 8, hipMemcpyDeviceToHost));
    // x_l = X_trial - X_cur
    std::vector<double> a((size_t)d.L * 3), b((size_t)d.L * 3);
    if (d.L) {
      OSH_HIP(hipMemcpy(a.data(), c->d_pt[1].as<double>() + (size_t)d.pt_off * 3, a.size() * 8, hipMemcpyDeviceToHost));
      OSH_HIP(hipMemcpy(b.data(), c->d_pt[0].as<double>() + (size_t)d.pt_off * 3, b.size() * 8, hipMemcpyDeviceToHost));
      for (size_t k = 0; k < a.size(); ++k) x[d.n + k] = a[k] - b[k];
    }
  }
  c->optimized = false;
  return OSH_OK;
}

extern "C" int osh_lba_set_profiling(osh_lba_ctx* c, int enable) {
  if (!c) return OSH_ERR_INVALID;
  OSH_HIP(hipSetDevice(c->device));
  if (enable) OSH_TRY(c->timer.init());
  c->timer.enabled = enable != 0;
  c->timer.reset();
  return OSH_OK;
}

extern "C" int osh_lba_get_profile(osh_lba_ctx* c, int64_t launches[OSH_K_COUNT], double total_ms[OSH_K_COUNT]) {
  if (!c || !launches || !total_ms) return OSH_ERR_INVALID;
  for (int k = 0; k < OSH_K_COUNT; ++k) { launches[k] = c->timer.launches[k]; total_ms[k] = c->timer.total_ms[k]; }
  return OSH_OK;
}

extern "C" int osh_lba_get_plan_stats(osh_lba_ctx* c, int64_t stats[6]) {
  if (!c || !stats || c->n_windows <= 0) { set_error("osh_lba_get_plan_stats: nothing uploaded"); return OSH_ERR_INVALID; }
  stats[0] = (int64_t)c->n_items; stats[1] = (int64_t)c->n_sym; stats[2] = c->plan_tile_steps; stats[3] = c->plan_pair_blocks;
  stats[4] = (int64_t)c->n_contrib; stats[5] = (int64_t)c->n_rblk;
  return OSH_OK;
}

extern "C" const char* osh_lba_kernel_name(int k) {
  static const char* names[OSH_K_COUNT] = {"k_lin_items<0, false>", "k_pose_reduce", "k_schur_items<true>", "k_solve", "k_backsub<false>", "k_residual<false>", "k_control", "k_schur_reduce", "k_schur_items<false>", "k_lin_aux<false>", "k_lin_items<1, false>"};   // pinhole instantiations (<.., true> for a fisheye batch)
  return (k >= 0 && k < OSH_K_COUNT) ? names[k] : "?";
}
