// lba_device.hip -- local bundle adjustment on MI355X (gfx950): HIP kernels + C-ABI driver.
//
// Replaces, for a batch of independent windows, what
//   optimizer.initializeOptimization(); optimizer.optimize(10);     src/Optimizer.cc:1410-1411
// does on one CPU thread in the reference (g2o LM + Schur block solver).  See
// DESIGN.md for the data layout and the roofline of each kernel.
//
// Kernel map (reference loop -> kernel), "g2o/" = Thirdparty/g2o/g2o/:
//   k_lin_lm      computeActiveErrors + activeRobustChi2 + the landmark side of buildSystem
//                 g2o/core/sparse_optimizer.cpp:61-114, block_solver.hpp:502-560,
//                 base_binary_edge.hpp:55-120  (Hll, b_l) + setLambda / D->inverse() of every landmark
//                 (block_solver.hpp:389,582-587) as the factor F, F F^T = (Hll + lambda I)^-1
//   k_schur_fused the pose side of buildSystem (per-item Hpp / b_p partials) and BlockSolver::solve's Schur
//                 part, block_solver.hpp:367-439: landmarks grouped by observer set (schur_plan.h), one
//                 wavefront per group, (B F)(B F)^T on the FP64 matrix cores.  The 6x3 blocks Hpl are formed
//                 in registers from the edge description (lba_math.h) and never stored
//   k_pose_reduce Hpp, b_p from the group partials, fixed order
//   k_schur_reduce  S = Hpp + lambda I - sum of the group products, b_s = b_p - ..., fixed order
//   k_solve       LinearSolverEigen::solve -> dense blocked LDL^T, g2o/solvers/linear_solver_eigen.h:94-124,
//                 + pose update (VertexSE3Expmap::oplusImpl)
//   k_backsub     landmark back-substitution block_solver.hpp:461-483 + VertexSBAPointXYZ::oplusImpl
//                 + computeScale partials (optimization_algorithm_levenberg.cpp:187-194)
//   k_residual    computeActiveErrors + activeRobustChi2 at the trial estimates
//   k_control     the Levenberg-Marquardt controller, optimization_algorithm_levenberg.cpp:61-169,
//                 and the optimize() loop conditions, sparse_optimizer.cpp:354-419
//   k_finalize    per-edge chi2 / isDepthPositive for the outlier test, src/Optimizer.cc:1413-1460
#include "common.h"
#include <type_traits>
#include <cstdlib>
#include <cstdio>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include "lba_math.h"
#include "ldlt_block.h"
#include "big_solve.h"
#include "schur_plan.h"
#include "lba_pack.h"
#include "lba_pack_device.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace osh {

constexpr int kResidualSplit = 1;    // blocks of k_residual per chunk
constexpr double kTau = 1e-5;      // OptimizationAlgorithmLevenberg::_tau
constexpr int kMaxTrials = 10;     // maxTrialsAfterFailure
constexpr int kSolveThreadsBatch = 256, kSolveThreadsLatency = 512;
constexpr int kPoseRec = 21;       // staged pose: quaternion + translation (7), camera (5), rotation matrix (9)
constexpr int kDevicePackMinWindows = 24;   // smaller batches are packed by host threads (osh_lba_upload)
constexpr int kLdsPoses = 512;     // windows with more poses read them from global memory in the landmark-major kernels

struct LmState {
  double lambda, ni, currentChi, iniChi, tempChi, rho, scale_pose, chi2_initial;
  int iter;           // index of the running solve() call
  int qmax;           // trials of the running iteration
  int nBad;
  int active;         // window still inside optimize()
  int need_lin;       // next round opens a new iteration (linearise at cur)
  int lin_now;        // this round opened an iteration: the pose side of the linearisation belongs to it
  int need_dl;        // lambda changed since the landmark factors were formed (first lambda, rejected trial)
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
  const unsigned char* e_kind;   // [NE] sorted-edge kind (kKind*); the pinhole kernels read the sign of e_rec[.][3] instead
  const double* e_rec;       // [NE*4] u v u_right +-invSigma2 of every sorted edge: one 32-byte record (sign bit set = mono)
  const float4* e_rec32;     // the same records as the float32 values they were uploaded as (null unless every record is exact in
                             // float32): the landmark-major kernels are bandwidth bound and read these, 16 bytes an edge
  const double* e_rec2;      // [NE*4] right-camera observation of a fisheye-rig edge (u v - invSigma2), rig batches only
  const int* e_orig;         // [NE] index in the caller's edge order
  const int* e_orig2;        // [NE] caller index of the merged right-camera edge or -1, rig batches only
  const int* lm_off;         // per window L+1 offsets (window-local edge index)
  // Schur plan (schur_plan.h)
  const SItem* sitems;       // [n_items] symmetric items first
  const SRec* srecs;         // landmark records of the items
  const int* spair;          // [n_items*64] contribution index of pose pair (sa,sb) or -1
  const int* scslot;         // [n_items*8] rhs contribution index of row pose sa or -1
  const int* sposex;         // [n_items*8] window-local row pose of slot sa or -1
  const int* sposey;         // [n_items*8] window-local column pose of slot sb or -1
  const I2* pose_crange;     // [NFP] {first, count} of the pose's (item, pose) contributions
  double* hcontrib;          // [n_ccontrib*27] per linearisation: upper(Hpp) (21) + b_p (6) of one item and pose
  double* chi_lin;           // [n_chunks] robust chi2 partial of each chunk at the linearisation point
  double* dmax_lin;          // [n_chunks] largest Hll diagonal entry of the chunk
  const RBlk* rblk;          // [n_rblk] blocks of S + rhs segments with their contribution ranges
  int n_rblk;
  int* pose_lo;              // [NFP] first block row with a non-zero in block column p of the window's S (its column envelope), k_schur_env
  double* contrib;           // [n_contrib*36] per trial: 6x6 products of one item and pose pair
  double* ccontrib;          // [n_ccontrib*6] per trial: rhs products of one item and pose
  // system
  double* Hll;               // [NL*6] upper: 00 01 02 11 12 22
  double* bl;                // [NL*3]
  double* DL;                // [NL*9] landmark factor F (6) + F^T b_l (3), see landmark_factor()
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
  double* out_chi2;          // [NOUT] caller order
  unsigned char* out_depth;  // [NOUT]
};

// Read-only snapshot of the controller fields a kernel needs, taken once at kernel entry (a reference into
// global memory would be re-read after every store: the compiler cannot prove the stores do not alias it).
struct LmView { double lambda; int active, need_lin, lin_now, need_dl, iter, sel, solve_ok, last_eval_sel, iterations; };
__device__ __forceinline__ LmView lm_view(const LmState* lm, int w) {
  const LmState& s = lm[w];
  return LmView{s.lambda, s.active, s.need_lin, s.lin_now, s.need_dl, s.iter, s.sel, s.solve_ok, s.last_eval_sel, s.iterations};
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
// every outstanding GLOBAL access of the wavefront (it is also a workgroup-scope memory fence).
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Edge description (lba_math.h) of sorted edge `ge` through the window's camera model.  KB8 is a compile-time switch: the
// pinhole instantiations of the kernels (every batch without a fisheye window) carry no KannalaBrandt8 code or registers.
// chi_l / chi_r are only meaningful for KB8 windows (per-edge chi2 of a merged rig edge).
template <bool KB8>
__device__ __forceinline__ void win_edge_core(const WinDesc& wd, const BatchView& bv, size_t ge, const double* rec, const double* qt,
                                              const double* cam, const double* R, const double* X, double* Xc, double* Q, double* g, double& rho0) {
  if (KB8 && wd.kb8_on) {
    double cl, cr;
    double rec2[4] = {0.0, 0.0, 0.0, 0.0};
    const int kind = bv.e_kind[ge];
    if (kind == kKindBoth) {
#pragma unroll
      for (int k = 0; k < 4; ++k) rec2[k] = bv.e_rec2[ge * 4 + k];
    }
    dev::edge_core_kb8(kind, qt, cam, wd.kb8, wd.cam2, wd.trl, X, rec, rec2, wd.huber_mono, Xc, Q, g, rho0, cl, cr);
    return;
  }
  dev::edge_core_pinhole(R, qt + 4, cam, X, rec, wd.huber_mono, wd.huber_stereo, Xc, Q, g, rho0);
}

// Robustified chi2 of sorted edge `ge` (the trial residual needs nothing else); chi_l / chi_r as above.
template <bool KB8>
__device__ __forceinline__ double win_edge_rho(const WinDesc& wd, const BatchView& bv, size_t ge, const double* rec, const double* qt,
                                               const double* cam, const double* R, const double* X, double& chi_l, double& chi_r) {
  if (KB8 && wd.kb8_on) {
    double Xc[3], Q[6], g[3], rho0;
    double rec2[4] = {0.0, 0.0, 0.0, 0.0};
    const int kind = bv.e_kind[ge];
    if (kind == kKindBoth) {
#pragma unroll
      for (int k = 0; k < 4; ++k) rec2[k] = bv.e_rec2[ge * 4 + k];
    }
    dev::edge_core_kb8(kind, qt, cam, wd.kb8, wd.cam2, wd.trl, X, rec, rec2, wd.huber_mono, Xc, Q, g, rho0, chi_l, chi_r);
    return rho0;
  }
  const bool stereo = rec[3] > 0.0;
  double r[3], Xc[3], rho0, rho1, iz;
  chi_l = dev::edge_residual_pinhole(R, qt + 4, cam, X, rec, r, Xc, iz);   // the same expressions as win_edge_core: chi2 of a state is one number
  chi_r = 0.0;
  dev::huber_fast(chi_l, stereo ? wd.huber_stereo : wd.huber_mono, rho0, rho1);
  return rho0;
}

// Stages the window's poses (quaternion + translation, camera, rotation matrix) in LDS for a landmark-major block.
__device__ __forceinline__ void stage_poses(double* sh_pose, const double* poses, const double* cams, int n_poses, int tid, int nthreads) {
  for (int i = tid; i < n_poses; i += nthreads) {
    double qt[7], R[9];
#pragma unroll
    for (int k = 0; k < 7; ++k) qt[k] = poses[(size_t)i * 7 + k];
    dev::quat_to_R(qt, R);
#pragma unroll
    for (int k = 0; k < 7; ++k) sh_pose[i * kPoseRec + k] = qt[k];
#pragma unroll
    for (int k = 0; k < 5; ++k) sh_pose[i * kPoseRec + 7 + k] = cams[(size_t)i * 5 + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) sh_pose[i * kPoseRec + 12 + k] = R[k];
  }
}
__device__ __forceinline__ void fetch_pose(bool staged, const double* sh_pose, const double* poses, const double* cams, int ip,
                                           double* qt, double* cam, double* R) {
  if (staged) {
#pragma unroll
    for (int k = 0; k < 7; ++k) qt[k] = sh_pose[ip * kPoseRec + k];
#pragma unroll
    for (int k = 0; k < 5; ++k) cam[k] = sh_pose[ip * kPoseRec + 7 + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = sh_pose[ip * kPoseRec + 12 + k];
  } else {
#pragma unroll
    for (int k = 0; k < 7; ++k) qt[k] = poses[(size_t)ip * 7 + k];
#pragma unroll
    for (int k = 0; k < 5; ++k) cam[k] = cams[(size_t)ip * 5 + k];
    dev::quat_to_R(qt, R);
  }
}

// --------------------------------------------------------------------------------------------
// Landmark-major kernels (k_lin_lm, k_backsub, k_residual): one block per chunk of consecutive landmarks (<= 1024 edges,
// <= 256 landmarks), lane per edge in kPasses passes of 256.  The block first requests everything it will touch -- the
// lane's edges of all passes (pose index, landmark index, observation record), the chunk's landmarks (coalesced) and the
// window's poses -- parks the shared part in LDS behind ONE barrier and then only computes: one memory round trip per block
// instead of two per pass.  (A wavefront-granular variant -- chunks of 64 edges, no block barriers, per-landmark sums as
// parallel (landmark, component) tasks -- was measured 2x SLOWER: the per-chunk reduction and the cold start of every
// short-lived wavefront cost more than the four barriers of a 1024-edge block.)
// --------------------------------------------------------------------------------------------
constexpr int kPasses = kChunkMaxEdges / kChunkEdges;

struct LaneEdges { int ip[kPasses], il[kPasses]; double2 ra[kPasses], rb[kPasses]; };

// edges base + p * 256 + tid, p = 0..kPasses-1, clamped to the last edge of the chunk (validity is applied at use)
// F32: the records are read as the float32 values they were uploaded as (exact: the packer checked) -- 16 bytes an edge instead of
// 32 in the three bandwidth-bound landmark-major kernels.  A compile-time choice: a run-time test in this loop splits the block of
// loads the kernels issue up front (measured: k_lin_lm 0.68 -> 1.09 ms with the branch, 0.56 ms without).
template <bool F32>
__device__ __forceinline__ void load_lane_edges(const BatchView& bv, const WinDesc& wd, int base, int e1, int tid, LaneEdges& le) {
  const int last = max(e1 - 1, 0);
#pragma unroll
  for (int p = 0; p < kPasses; ++p) {
    const size_t ge = (size_t)wd.edge_off + min(base + p * kChunkEdges + tid, last);
    le.ip[p] = bv.e_pose[ge];
    le.il[p] = bv.e_point[ge];
    if (F32) {
      const float4 r = bv.e_rec32[ge];
      le.ra[p] = make_double2((double)r.x, (double)r.y); le.rb[p] = make_double2((double)r.z, (double)r.w);
    } else {
      const double2* r = reinterpret_cast<const double2*>(bv.e_rec + ge * 4);
      le.ra[p] = r[0]; le.rb[p] = r[1];
    }
  }
}

// the chunk's landmarks -> LDS [nl][3] (coalesced: consecutive landmarks are consecutive in memory)
__device__ __forceinline__ void stage_points(double* sh_X, const double* pts, int lm0, int nl, int tid) {
  for (int k = tid; k < 3 * nl; k += kBlock) sh_X[k] = pts[(size_t)lm0 * 3 + k];
}

// --------------------------------------------------------------------------------------------
// k_residual: computeActiveErrors + activeRobustChi2 at the TRIAL estimates (every active window); chunk partial sums in
// fixed order.
// --------------------------------------------------------------------------------------------
template <bool KB8, bool F32>
__global__ __launch_bounds__(kBlock) void k_residual(BatchView bv) {
  extern __shared__ __attribute__((aligned(16))) double sh_rs[];   // [4] reduce, [256*3] landmarks, staged poses
  const Chunk ch = bv.chunks[blockIdx.x];
  const WinDesc wd = bv.win[ch.win];   // by value: the fields stay in SGPRs across the kernel's stores
  const LmView st = lm_view(bv.lm, ch.win);
  if (!st.active) return;
  const int tid = threadIdx.x;
  const int sel = st.sel ^ 1;
  double* sh4 = sh_rs;
  double* sh_X = sh4 + 4;
  double* sh_pose = sh_X + 3 * kBlock;
  const double* poses = bv.pose_state[sel] + (size_t)wd.pose_off * 7;
  const double* pts = bv.pt_state[sel] + (size_t)wd.pt_off * 3;
  const double* cams = bv.pose_cam + (size_t)wd.pose_off * 5;
  const int* lmo = bv.lm_off + wd.lmoff_off;
  const int e0 = lmo[ch.lm0], e1 = lmo[ch.lm1];
  const int nl = ch.lm1 - ch.lm0;
  const bool staged = (wd.P + wd.F) <= kLdsPoses;
  double chi_acc = 0.0;
  for (int base = e0; base < e1; base += kChunkMaxEdges) {
    LaneEdges le;
    load_lane_edges<F32>(bv, wd, base, e1, tid, le);
    if (base == e0) {
      stage_points(sh_X, pts, ch.lm0, nl, tid);
      if (staged) stage_poses(sh_pose, poses, cams, wd.P + wd.F, tid, kBlock);
      __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
      const int e = base + p * kChunkEdges + tid;
      if (e < e1) {
        double qt[7], cam[5], R[9], X[3];
        fetch_pose(staged, sh_pose, poses, cams, le.ip[p], qt, cam, R);
#pragma unroll
        for (int k = 0; k < 3; ++k) X[k] = sh_X[(le.il[p] - ch.lm0) * 3 + k];
        const double rec[4] = {le.ra[p].x, le.ra[p].y, le.rb[p].x, le.rb[p].y};
        double cl, cr;
        chi_acc += win_edge_rho<KB8>(wd, bv, (size_t)wd.edge_off + e, rec, qt, cam, R, X, cl, cr);
      }
    }
  }
  const double chi = block_sum(chi_acc, sh4);
  if (tid == 0) bv.chi_part[blockIdx.x] = chi;
}

// --------------------------------------------------------------------------------------------
// k_lin_lm: the landmark side of the linearisation: residual, Huber weight, d err / d point of EVERY edge of the landmark
// (optimisable and fixed keyframes) ->
//   Hll_j = sum R^T Q R,  b_l(j) = sum R^T g   (summed per landmark in edge order by the landmark's own lane: fixed order)
//   chi2 partial and the largest Hll diagonal entry of the chunk (computeLambdaInit),
// then the landmark factor for the current lambda (landmark_factor(): setLambda + D->inverse(), block_solver.hpp:389,582-587).
// A window that is not linearising this round but whose lambda changed (first lambda of optimize(), rejected trial) only
// re-forms its factors from the stored Hll / b_l.
// --------------------------------------------------------------------------------------------
template <bool KB8, bool F32>
__global__ __launch_bounds__(kBlock) void k_lin_lm(BatchView bv) {
  extern __shared__ __attribute__((aligned(16))) double sh_lm[];   // [9*256] partials, [4] reduce, [256*3] landmarks, staged poses
  const Chunk ch = bv.chunks[blockIdx.x];
  const WinDesc wd = bv.win[ch.win];   // by value: the fields stay in SGPRs across the kernel's stores
  const LmView st = lm_view(bv.lm, ch.win);
  if (!st.active || (!st.need_lin && !st.need_dl)) return;
  const int tid = threadIdx.x;
  const int nl = ch.lm1 - ch.lm0;
  const bool lambda_known = st.lambda >= 0.0;   // k_reset leaves -1 until k_control opened the first iteration
  if (!st.need_lin) {
    if (tid < nl) {
      const size_t gl = (size_t)wd.pt_off + ch.lm0 + tid;
      double hl[6], bl[3], dl[9];
#pragma unroll
      for (int k = 0; k < 6; ++k) hl[k] = bv.Hll[gl * 6 + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) bl[k] = bv.bl[gl * 3 + k];
      dev::landmark_factor(hl, bl, st.lambda, dl);
#pragma unroll
      for (int k = 0; k < 9; ++k) bv.DL[gl * 9 + k] = dl[k];
    }
    return;
  }
  double* sh_c = sh_lm;
  double* sh4 = sh_lm + 9 * kChunkEdges;
  double* sh_X = sh4 + 4;
  double* sh_pose = sh_X + 3 * kBlock;
  const double* poses = bv.pose_state[st.sel] + (size_t)wd.pose_off * 7;
  const double* pts = bv.pt_state[st.sel] + (size_t)wd.pt_off * 3;
  const double* cams = bv.pose_cam + (size_t)wd.pose_off * 5;
  const bool staged = (wd.P + wd.F) <= kLdsPoses;
  const int* lmo = bv.lm_off + wd.lmoff_off;
  const int e0 = lmo[ch.lm0], e1 = lmo[ch.lm1];
  int my_lo = 0, my_hi = 0;
  if (tid < nl) { my_lo = lmo[ch.lm0 + tid]; my_hi = lmo[ch.lm0 + tid + 1]; }
  double acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.0;
  double chi_acc = 0.0;
  for (int base = e0; base < e1; base += kChunkMaxEdges) {
    LaneEdges le;
    load_lane_edges<F32>(bv, wd, base, e1, tid, le);
    if (base == e0) {
      stage_points(sh_X, pts, ch.lm0, nl, tid);
      if (staged) stage_poses(sh_pose, poses, cams, wd.P + wd.F, tid, kBlock);
      __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
      const int pbase = base + p * kChunkEdges;
      if (pbase >= e1) break;     // uniform over the block
      const int e = pbase + tid;
      double hl[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) hl[k] = 0.0;
      if (e < e1) {
        double qt[7], cam[5], R[9], X[3], Xc[3], Q[6], g[3], rho0;
        fetch_pose(staged, sh_pose, poses, cams, le.ip[p], qt, cam, R);
#pragma unroll
        for (int k = 0; k < 3; ++k) X[k] = sh_X[(le.il[p] - ch.lm0) * 3 + k];
        const double rec[4] = {le.ra[p].x, le.ra[p].y, le.rb[p].x, le.rb[p].y};
        win_edge_core<KB8>(wd, bv, (size_t)wd.edge_off + e, rec, qt, cam, R, X, Xc, Q, g, rho0);
        chi_acc += rho0;
        dev::core_landmark_side<KB8>(Q, g, R, hl);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) sh_c[k * kChunkEdges + tid] = hl[k];
      __syncthreads();
      if (tid < nl) {
        const int lo = max(my_lo, pbase), hi = min(my_hi, pbase + kChunkEdges);
        for (int x = lo; x < hi; ++x) {
#pragma unroll
          for (int k = 0; k < 9; ++k) acc[k] += sh_c[k * kChunkEdges + x - pbase];
        }
      }
      __syncthreads();
    }
  }
  double dmax = 0.0;
  if (tid < nl) {
    const size_t gl = (size_t)wd.pt_off + ch.lm0 + tid;
#pragma unroll
    for (int k = 0; k < 6; ++k) bv.Hll[gl * 6 + k] = acc[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) bv.bl[gl * 3 + k] = acc[6 + k];
    dmax = fmax(fabs(acc[0]), fmax(fabs(acc[3]), fabs(acc[5])));
    if (lambda_known) {
      double dl[9];
      dev::landmark_factor(acc, acc + 6, st.lambda, dl);
#pragma unroll
      for (int k = 0; k < 9; ++k) bv.DL[gl * 9 + k] = dl[k];
    }
  }
  const double chi = block_sum(chi_acc, sh4);
  dmax = block_max(dmax, sh4);
  if (tid == 0) { bv.chi_lin[blockIdx.x] = chi; bv.dmax_lin[blockIdx.x] = dmax; }
}

// --------------------------------------------------------------------------------------------
// k_schur_fused: the pose side of the linearisation and the landmark products of the Schur complement
// (block_solver.hpp:381-432), one wavefront per ITEM of schur_plan.h (landmarks that share their set of optimisable
// observers).  Per chunk of 8 landmarks lane (l, s) owns the edge of landmark l / row pose X[s]:
//   edge description (Xc, Q, g) at the current estimates from the 32-byte observation record, the landmark and the
//   lane's pose (kept in LDS)                                           -> nothing 6x3 is read from memory
//   pose side (once per iteration): Hpp_s += D^T Q D, b_p(s) += D^T g in registers, one 27-double contribution
//   per (item, pose) at the end, reduced in plan order by k_pose_reduce          (base_binary_edge.hpp:55-120)
//   Schur side (every trial): the rows of  W F,  W = Hpl block = D^T Q R,  F F^T = (Hll + lambda I)^-1 from k_lin_lm,
//   stored as a [row][k] image in LDS (row = 6 s + r, k = 3 l + m; row stride 25 doubles: conflict-free MFMA reads),
//   and the rhs term (W F)(F^T b_l) (the _coefficients term, block_solver.hpp:404-409).
// S_ab -= (W_a F)(W_b F)^T is then 6 k-steps of v_mfma_f64_16x16x4_f64 per 16x16 tile, BOTH operands read from the one
// image for symmetric items; the accumulators stay in registers across the chunks of the item.  At the end every live
// pose pair (sa, sb) is written as one 6x6 contribution; k_schur_reduce sums them in plan order.
//   SYM: X == Y, only the tiles on and above the diagonal; owns the pose side and the rhs term.
//   cross items (parts a < b of a landmark with more than 8 optimisable observers): a second image for the Y poses.
//   mode 0: pose side only (the first round of optimize(): computeLambdaInit needs Hpp before any Schur product)
//   mode 1: Schur products, and the pose side when this round opened an iteration other than the first
// --------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kSiLm = 8;               // landmarks per chunk, items of 6 .. 8 row poses (and every cross item)
constexpr int kSiRows = 6 * kItemPoses;
// Lane mappings of a chunk, LM landmarks x PS pose lanes (lane = l * PS + s): 8 x 8, and for symmetric items of at most 5 / at most 4
// row poses 12 x 5 / 16 x 4 -- on the headline window 26 % / 24 % of the symmetric chunks belong to such items and left 3 / 4..7 of their
// 8 pose lanes empty.  The image is [6 PS rows, padded to whole 16-row tiles][3 LM + 1]: 48 x 25, 32 x 37, 32 x 49 doubles.  The k
// steps run over the same sequence of (landmark, component) columns whatever the chunk size (24, 36 and 48 are multiples of 4), so
// the Schur products are the same bits as with 8 x 8 chunks.  (32 x 2 for items of one or two poses -- 18 % of the chunks -- was
// measured too and gained nothing over 16 x 4: with one tile the 24 k steps of such a chunk are one dependent MFMA chain.)
constexpr int kSiImage = 32 * 49;

template <bool SYM, bool KB8>
__global__ __launch_bounds__(64, KB8 ? 1 : 2) void k_schur_fused(BatchView bv, int item_base, int mode) {
  __shared__ double shA[(SYM && !KB8) ? kSiImage : kSiRows * (3 * kSiLm + 1)];   // (the wider mappings: symmetric pinhole items only)
  __shared__ double shB[SYM ? 1 : kSiRows * (3 * kSiLm + 1)];
  __shared__ double shPose[(SYM ? 1 : 2) * 8 * kPoseRec];
  __shared__ int shP[64 + 8];
  const int item_idx = item_base + blockIdx.x;
  const SItem it = bv.sitems[item_idx];
  const WinDesc wd = bv.win[it.win];   // by value: the fields stay in SGPRs across the kernel's stores
  const LmView st = lm_view(bv.lm, it.win);
  if (!st.active) return;
  const bool do_hpp = SYM && (mode == 0 ? st.need_lin != 0 : (st.lin_now != 0 && st.iter > 0));
  const bool do_schur = mode != 0;
  if (!do_hpp && !do_schur) return;
  const int lane = threadIdx.x;
  const int nx = it.shape & 0xff, ny = (it.shape >> 8) & 0xff;
  const int TX = (6 * nx + 15) >> 4, TY = (6 * ny + 15) >> 4;
  shP[lane] = bv.spair[(size_t)item_idx * 64 + lane];
  if (lane < 8) shP[64 + lane] = bv.scslot[(size_t)item_idx * 8 + lane];
  {
    // poses of the item's slots -> LDS (lanes 0..7: row poses, lanes 8..15: column poses of a cross item)
    const double* poses = bv.pose_state[st.sel] + (size_t)wd.pose_off * 7;
    const double* cams = bv.pose_cam + (size_t)wd.pose_off * 5;
    const int side = lane >> 3, s8 = lane & 7;
    if (side < (SYM ? 1 : 2)) {
      const int ip = side == 0 ? bv.sposex[(size_t)item_idx * 8 + s8] : bv.sposey[(size_t)item_idx * 8 + s8];
      double qt[7], cam[5], Rm[9];
#pragma unroll
      for (int k = 0; k < 7; ++k) qt[k] = ip >= 0 ? poses[(size_t)ip * 7 + k] : 0.0;
#pragma unroll
      for (int k = 0; k < 5; ++k) cam[k] = ip >= 0 ? cams[(size_t)ip * 5 + k] : 0.0;
      dev::quat_to_R(qt, Rm);
      double* dst = shPose + (side * 8 + s8) * kPoseRec;
#pragma unroll
      for (int k = 0; k < 7; ++k) dst[k] = qt[k];
#pragma unroll
      for (int k = 0; k < 5; ++k) dst[7 + k] = cam[k];
#pragma unroll
      for (int k = 0; k < 9; ++k) dst[12 + k] = Rm[k];
    }
  }
  const SRec* __restrict__ recs = bv.srecs + it.rec_off;
  const double* pts = bv.pt_state[st.sel] + (size_t)wd.pt_off * 3;
  const double* DL = bv.DL + (size_t)wd.pt_off * 9;
  const int mrow = lane & 15, mk = lane >> 4;
  using std::integral_constant;

  auto body = [&](auto lmc, auto psc) __attribute__((always_inline)) {
    constexpr int LM = decltype(lmc)::value, PS = decltype(psc)::value;   // landmarks per chunk, pose lanes
    constexpr int KS = 3 * LM + 1;                                        // LDS row stride (doubles): odd, conflict-free MFMA reads
    constexpr int KST = (3 * LM) / 4;                                     // k steps per chunk
    static_assert(LM * PS <= 64 && (3 * LM) % 4 == 0 && ((6 * PS + 15) / 16) * 16 * KS <= ((SYM && !KB8) ? kSiImage : kSiRows * (3 * kSiLm + 1)), "chunk mapping");
    const int l = lane / PS, s = lane - l * PS;
    const bool lane_on = LM * PS == 64 || l < LM;
    f64x4 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
    double csum[6] = {0, 0, 0, 0, 0, 0};
    double H[21], hb[6];
#pragma unroll
    for (int k = 0; k < 21; ++k) H[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) hb[k] = 0.0;
    if (PS < 8) {
      // image rows past the pose lanes (up to the end of the last tile): nobody writes them
      for (int idx = lane; idx < (((6 * PS + 15) / 16) * 16 - 6 * PS) * KS; idx += 64) shA[6 * PS * KS + idx] = 0.0;
    }

    // Software pipeline over the chunks: the data of chunk c+1 are requested right after chunk c's operands are parked in LDS,
    // so they are in flight during chunk c's MFMA phase, and the record of chunk c+2 with them.  Every load is unconditional
    // on a clamped index (a load inside a divergent branch is waited for at the end of the branch); validity is applied
    // when the values are used.
    struct Dat { double2 ex[2]; double2 ey[2]; double X[3]; double dl[9]; };
    const int last_rec = it.n_lm - 1;
    const size_t last_edge = (size_t)wd.edge_off + (size_t)max(wd.E - 1, 0);
    auto load_rec = [&](int c0, int4& ra, int4& rb) {
      const int4* src = reinterpret_cast<const int4*>(recs + min(c0 + l, last_rec));
      ra = src[0]; rb = src[1];          // {lm, e_first, x_lo, x_hi} {y_lo, y_hi, flags, pad}
    };
    auto slot_of = [&](unsigned lo, unsigned hi) { return (((s < 4) ? lo : hi) >> (8 * (s & 3))) & 0xffu; };
    auto edge_of = [&](const int4& ra, unsigned o) { return min((size_t)wd.edge_off + (size_t)(ra.y + (o != kAbsent ? (int)o : 0)), last_edge); };
    auto load_dat = [&](const int4& ra, const int4& rb, Dat& d) {
      const double2* src = reinterpret_cast<const double2*>(bv.e_rec + edge_of(ra, slot_of((unsigned)ra.z, (unsigned)ra.w)) * 4);
      d.ex[0] = src[0]; d.ex[1] = src[1];
      if (!SYM) {
        const double2* sy = reinterpret_cast<const double2*>(bv.e_rec + edge_of(ra, slot_of((unsigned)rb.x, (unsigned)rb.y)) * 4);
        d.ey[0] = sy[0]; d.ey[1] = sy[1];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) d.X[k] = pts[(size_t)ra.x * 3 + k];
      if (do_schur) {
#pragma unroll
        for (int k = 0; k < 9; ++k) d.dl[k] = DL[(size_t)ra.x * 9 + k];
      }
    };
    // rows of W F of the lane's edge on one side of the item (side 0: row poses, 1: column poses) -> the side's LDS image
    auto side_rows = [&](int side, bool present, const int4& ra, unsigned o, const double2* er, const Dat& d, double* img) __attribute__((always_inline)) {
      const double* ps = shPose + (side * 8 + s) * kPoseRec;
      double qt[7], cam[5], Rm[9];
#pragma unroll
      for (int k = 0; k < 7; ++k) qt[k] = ps[k];
#pragma unroll
      for (int k = 0; k < 5; ++k) cam[k] = ps[7 + k];
#pragma unroll
      for (int k = 0; k < 9; ++k) Rm[k] = ps[12 + k];
      const double rec[4] = {er[0].x, er[0].y, er[1].x, er[1].y};
      double Xc[3], Q[6], g[3], rho0;
      win_edge_core<KB8>(wd, bv, edge_of(ra, o), rec, qt, cam, Rm, d.X, Xc, Q, g, rho0);
      // an empty slot computes on another edge's data: its result is discarded here (select, not multiply: it may be NaN)
#pragma unroll
      for (int k = 0; k < 6; ++k) Q[k] = present ? Q[k] : 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) g[k] = present ? g[k] : 0.0;
      if (SYM && do_hpp && side == 0) dev::core_pose_side<KB8>(Xc, Q, g, H, hb);
      if (do_schur) {
        double WF[18];
        dev::core_WF<KB8>(Xc, Q, Rm, d.dl, WF);
        double* row = img + (6 * s) * KS + 3 * l;
        if (lane_on) {
#pragma unroll
          for (int r = 0; r < 6; ++r) { row[r * KS + 0] = WF[r * 3]; row[r * KS + 1] = WF[r * 3 + 1]; row[r * KS + 2] = WF[r * 3 + 2]; }
        }
        if (SYM) {
#pragma unroll
          for (int r = 0; r < 6; ++r) csum[r] += WF[r * 3] * d.dl[6] + WF[r * 3 + 1] * d.dl[7] + WF[r * 3 + 2] * d.dl[8];
        }
      }
    };
    // (always inlined: as a call, in the fisheye instantiation, every accumulator it touches by reference would live in scratch memory)
    auto park = [&](int c0, const int4& ra, const int4& rb, const Dat& d) __attribute__((always_inline)) {
      const bool valid = lane_on && (c0 + l) < it.n_lm;
      const unsigned xo = valid ? slot_of((unsigned)ra.z, (unsigned)ra.w) : kAbsent;
      wave_sync();   // the previous chunk's MFMA reads are done
      side_rows(0, xo != kAbsent, ra, xo, d.ex, d, shA);
      if (!SYM) {
        const unsigned yo = valid ? slot_of((unsigned)rb.x, (unsigned)rb.y) : kAbsent;
        side_rows(1, yo != kAbsent, ra, yo, d.ey, d, shB);
      }
    };
    // The tile counts are compile-time constants inside each instantiation (dispatched once per chunk): with run-time tile tests
    // every MFMA and every operand read sat behind its own scalar branch.
    auto multiply_t = [&](auto txc, auto tyc) {
      constexpr int TXc = decltype(txc)::value, TYc = decltype(tyc)::value;
      const double* imgB = SYM ? shA : shB;
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) {
        double a[3], b[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          a[t] = (t < TXc) ? shA[(16 * t + mrow) * KS + 4 * ks + mk] : 0.0;
          b[t] = SYM ? a[t] : ((t < TYc) ? imgB[(16 * t + mrow) * KS + 4 * ks + mk] : 0.0);
        }
#pragma unroll
        for (int ti = 0; ti < 3; ++ti)
#pragma unroll
          for (int tj = SYM ? ti : 0; tj < 3; ++tj)
            if (ti < TXc && tj < TYc) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
        if (ks & 1) __builtin_amdgcn_sched_barrier(0);   // operands of two k-steps in flight at most (the prefetched chunk needs the registers)
      }
    };
    auto multiply_cross = [&]() {
      // cross items keep run-time tile tests: nine instantiations cost registers for no gain
#pragma unroll
      for (int ks = 0; ks < KST; ++ks) {
        double a[3], b[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          a[t] = (t < TX) ? shA[(16 * t + mrow) * KS + 4 * ks + mk] : 0.0;
          b[t] = (t < TY) ? shB[(16 * t + mrow) * KS + 4 * ks + mk] : 0.0;
        }
#pragma unroll
        for (int ti = 0; ti < 3; ++ti)
#pragma unroll
          for (int tj = 0; tj < 3; ++tj)
            if (ti < TX && tj < TY) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
      }
    };
    int4 rc0, rc1, rn0, rn1;
    Dat d;
    load_rec(0, rc0, rc1);
    load_rec(LM, rn0, rn1);
    load_dat(rc0, rc1, d);
    wave_sync();   // the staged poses are visible
    // The chunk loop is instantiated per tile count (dispatched ONCE per item): a dispatch inside the loop makes the compiler
    // reconcile the register assignment of the accumulators at every merge (dozens of 64-bit moves per chunk).
    auto chunk_loop = [&](auto txc) {
      for (int c0 = 0; c0 < it.n_lm; c0 += LM) {
        park(c0, rc0, rc1, d);                 // edge descriptions, W F rows of chunk c0 -> LDS (consumes d)
        wave_sync();
        rc0 = rn0; rc1 = rn1;                  // loaded one iteration ago
        load_dat(rc0, rc1, d);                 // chunk c0 + 1: in flight during the MFMAs below
        load_rec(c0 + 2 * LM, rn0, rn1);
        if (do_schur) {
          if (SYM) multiply_t(txc, txc); else multiply_cross();
        }
      }
    };
    if (SYM && do_schur) {   // TX == TY
      if (PS > 5 && TX == 3) chunk_loop(integral_constant<int, 3>{});
      else if (PS > 2 && TX == 2) chunk_loop(integral_constant<int, 2>{});
      else chunk_loop(integral_constant<int, 1>{});
    } else {
      chunk_loop(integral_constant<int, 0>{});
    }
    wave_sync();
    if (do_schur) {
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
        // rhs term of row pose s: sum over the landmark lanes in fixed order
        if (lane_on) {
#pragma unroll
          for (int r = 0; r < 6; ++r) shA[l * (6 * PS) + 6 * s + r] = csum[r];
        }
        wave_sync();
        if (lane < 6 * PS) {
          double v = 0.0;
#pragma unroll
          for (int k = 0; k < LM; ++k) v += shA[k * (6 * PS) + lane];
          const int sa = lane / 6;
          const int slot = shP[64 + sa];
          if (slot >= 0) bv.ccontrib[(size_t)slot * 6 + (lane - 6 * sa)] = v;
        }
        wave_sync();
      }
    }
    if (SYM && do_hpp) {
      // Hpp / b_p of row pose s: sum over the landmark lanes in fixed order (two halves through the image buffer)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int k0 = half * 14, nk = half ? 13 : 14;
        wave_sync();
#pragma unroll
        for (int k = 0; k < 14; ++k) {
          const int kk = k0 + k;
          if (k < nk) shA[k * 64 + lane] = (kk < 21) ? H[kk < 21 ? kk : 0] : hb[(kk - 21) < 0 ? 0 : ((kk - 21) > 5 ? 5 : (kk - 21))];
        }
        wave_sync();
        for (int o = lane; o < PS * nk; o += 64) {
          const int k = o / PS, sa = o - k * PS;
          const int slot = shP[64 + sa];
          if (slot >= 0) {
            double v = 0.0;
#pragma unroll
            for (int ll = 0; ll < LM; ++ll) v += shA[k * 64 + ll * PS + sa];
            bv.hcontrib[(size_t)slot * 27 + k0 + k] = v;
          }
        }
      }
    }
  };
  if constexpr (SYM && !KB8) {
    if (nx <= 4) body(integral_constant<int, 16>{}, integral_constant<int, 4>{});
    else if (nx == 5) body(integral_constant<int, 12>{}, integral_constant<int, 5>{});
    else body(integral_constant<int, kSiLm>{}, integral_constant<int, 8>{});
  } else {
    body(integral_constant<int, kSiLm>{}, integral_constant<int, 8>{});
  }
}

// --------------------------------------------------------------------------------------------
// k_schur_reduce: S(i,j) = [i == j] (Hpp_i + lambda I) - sum of the block's contributions,
// b_s(i) = b_p(i) - sum of the pose's rhs contributions; 7 blocks of S per 256-thread group,
// one thread per entry, contributions in plan order.
// --------------------------------------------------------------------------------------------
// DEEP: 24 contributions in flight where a block has that many -- a call with few windows cuts its Schur items small (item_max_lm) and
// has four times as many contributions per block; in a large batch (about six per block) the extra registers only cost (0.315 -> 0.33 ms).
template <bool DEEP>
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
    int k = 0;
    if constexpr (DEEP) {
      for (; k + 24 <= rb.count; k += 24) {
        double t[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) t[u] = c[(size_t)(k + u) * 6];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 24; ++u) v -= t[u];
      }
    }
    for (; k + 8 <= rb.count; k += 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = c[(size_t)(k + u) * 6];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) v -= t[u];
    }
    for (; k < rb.count; ++k) v -= c[(size_t)k * 6];
    bv.bs[gp * 6 + el] = v;
    return;
  }
  // A block above the column envelope of S (no landmark joins keyframe i to j or to any keyframe before i, k_schur_env) is zero, is
  // never touched by the envelope factorisation of k_solve and was zeroed at upload: nothing to write.
  if (bv.pose_lo && i < bv.pose_lo[wd.fpose_off + j]) return;
  const int r = el / 6, cc = el - 6 * r;
  double v = 0.0;
  if (i == j) v = bv.Hpp[((size_t)wd.fpose_off + i) * 36 + el] + ((r == cc) ? st.lambda : 0.0);
  const double* c = bv.contrib + (size_t)rb.start * 36 + el;
  int k = 0;
  if constexpr (DEEP) {
    for (; k + 24 <= rb.count; k += 24) {
      double t[24];
#pragma unroll
      for (int u = 0; u < 24; ++u) t[u] = c[(size_t)(k + u) * 36];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 24; ++u) v -= t[u];
    }
  }
  for (; k + 8 <= rb.count; k += 8) {   // eight contributions in flight (one load per iteration left every memory round trip exposed),
    double t[8];                        // subtracted in plan order: the sum is the same number as before
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = c[(size_t)(k + u) * 36];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) v -= t[u];
  }
  for (; k < rb.count; ++k) v -= c[(size_t)k * 36];
  bv.S[wd.S_off + (size_t)(6 * i + r) * wd.n + 6 * j + cc] = v;
}

// k_pose_reduce: Hpp_i (full symmetric 6x6), b_p(i) and the largest diagonal entry of pose i from the item
// contributions, in plan order.  32 lanes per optimisable pose: lane k < 27 sums entry k of the contributions
// (one contribution = 27 contiguous doubles -> coalesced, the loads of successive contributions are independent).
// mode as in k_schur_fused: 0 = the first round of optimize(), 1 = rounds that opened a later iteration.
__global__ __launch_bounds__(64) void k_pose_reduce(BatchView bv, int mode) {
  const int gp = blockIdx.x * 2 + (threadIdx.x >> 5);
  const int k = threadIdx.x & 31;
  if (gp >= bv.n_fposes) return;
  const int w = bv.fpose_win[gp];
  const LmView st = lm_view(bv.lm, w);
  if (!st.active || !(mode == 0 ? st.need_lin != 0 : (st.lin_now != 0 && st.iter > 0))) return;
  const I2 rg = bv.pose_crange[gp];
  double a = 0.0;
  if (k < 27) {
    const double* src = bv.hcontrib + (size_t)rg.x * 27 + k;
    int c = 0;
    for (; c + 16 <= rg.y; c += 16) {   // sixteen contributions in flight (a single window cut into small items has ~100 per pose), added in plan order
      double t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = src[(size_t)(c + u) * 27];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 16; ++u) a += t[u];
    }
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
  wave_sync();
  if (k == 0) {
    const double* d = shd + (threadIdx.x & 32);
    bv.dmax_pose[gp] = fmax(fmax(fmax(d[0], d[6]), fmax(d[11], d[15])), fmax(d[18], d[20]));
  }
}

// Tail of the reduced-system solve of one window: x_p out, pose update T <- exp(x) T into the trial buffer and the pose part of
// computeScale.  `xs`: the solution (n doubles, LDS or global), `shw`: NT/64 doubles of LDS scratch.
template <int NT>
__device__ void solve_tail(BatchView& bv, const WinDesc& wd, LmState& st, const double* xs, double* shw, bool ok) {
  const int tid = threadIdx.x, n = wd.n;
  double* xp = bv.xp + (size_t)wd.fpose_off * 6;
  for (int k = tid; k < n; k += NT) xp[k] = xs[k];   // zeros when the factorisation failed
  __syncthreads();
  const int cur = st.sel, tr_sel = st.sel ^ 1;
  const double lambda = st.lambda;
  double sc = 0.0;
  for (int i = tid; i < wd.P; i += NT) {
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
    for (int k = 0; k < NT / 64; ++k) tot += shw[k];
    st.scale_pose = tot; st.solve_ok = ok ? 1 : 0;
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
  double* A = bv.S + wd.S_off;
  double* rhs = bv.bs + (size_t)wd.fpose_off * 6;
  double *xs, *shw;
  const bool ok = ldlt_solve_block<NB, kSolveThreads>(A, rhs, n, W, sh, xs, shw, bv.pose_lo ? bv.pose_lo + wd.fpose_off : nullptr);
  solve_tail<kSolveThreads>(bv, wd, st, xs, shw, ok);
}

// One reduced system too large for the LDS-resident factorisation (global BA of a long session): big_solve.h has factored and
// solved it in global memory, the solution sits in the window's `bs`; this is the tail of k_solve for that window.
__global__ __launch_bounds__(256) void k_big_finish(BatchView bv, int w, const int* fail) {
  __shared__ double shw[8];
  const WinDesc& wd = bv.win[w];
  LmState& st = bv.lm[w];
  if (!st.active) return;
  double* x = bv.bs + (size_t)wd.fpose_off * 6;
  const bool ok = *fail == 0;
  if (!ok) {
    for (int k = threadIdx.x; k < wd.n; k += 256) x[k] = 0.0;    // zeros when the factorisation failed
    __syncthreads();
  }
  solve_tail<256>(bv, wd, st, x, shw, ok);
}

// --------------------------------------------------------------------------------------------
// k_backsub: x_l = (Hll + lambda I)^-1 (b_l - Hpl^T x_p) = F F^T (...), X_trial = X + x_l, landmark part of computeScale
// (block_solver.hpp:461-483, sparse_block_matrix_ccs.h:103-129).  Landmark-major (see above): lane per edge for the products
// -Hpl^T x_p = -R^T Q (D x_p) formed from the edge description (lba_math.h), lane per landmark for the ordered sum; the
// optimisable poses of the window and x_p sit in LDS.
// --------------------------------------------------------------------------------------------
template <bool KB8, bool F32>
__global__ __launch_bounds__(kBlock) void k_backsub(BatchView bv) {
  extern __shared__ __attribute__((aligned(16))) double sh_bs[];  // [3*256] partials, [4] reduce, [256*3] landmarks, [n] x_p, [P*21] poses
  double* sh_c = sh_bs;
  double* sh4 = sh_bs + 3 * kChunkEdges;
  double* sh_X = sh4 + 4;
  double* sh_x = sh_X + 3 * kBlock;
  const Chunk ch = bv.chunks[blockIdx.x];
  const WinDesc wd = bv.win[ch.win];   // by value: the fields stay in SGPRs across the kernel's stores
  const LmView st = lm_view(bv.lm, ch.win);
  if (!st.active) return;
  const int tid = threadIdx.x;
  const int n = wd.n;
  const double* poses = bv.pose_state[st.sel] + (size_t)wd.pose_off * 7;
  const double* cams = bv.pose_cam + (size_t)wd.pose_off * 5;
  const double* pts = bv.pt_state[st.sel] + (size_t)wd.pt_off * 3;
  const bool staged = wd.P <= kLdsPoses;
  const double* xp = bv.xp + (size_t)wd.fpose_off * 6;
  double* sh_pose = sh_x + (staged ? n : 0);
  const int* lmo = bv.lm_off + wd.lmoff_off;
  const int e0 = lmo[ch.lm0], e1 = lmo[ch.lm1];
  const int nl = ch.lm1 - ch.lm0;
  const double lambda = st.lambda;
  double acc[3] = {0, 0, 0};
  int my_lo = 0, my_hi = 0;
  double F[6], b3[3], Xcur[3];
#pragma unroll
  for (int k = 0; k < 6; ++k) F[k] = 0.0;
  b3[0] = b3[1] = b3[2] = 0.0; Xcur[0] = Xcur[1] = Xcur[2] = 0.0;
  if (tid < nl) {
    // the landmark's own data is requested with everything else
    const size_t gl = (size_t)wd.pt_off + ch.lm0 + tid;
    my_lo = lmo[ch.lm0 + tid]; my_hi = lmo[ch.lm0 + tid + 1];
#pragma unroll
    for (int k = 0; k < 6; ++k) F[k] = bv.DL[gl * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { b3[k] = bv.bl[gl * 3 + k]; Xcur[k] = pts[(size_t)(ch.lm0 + tid) * 3 + k]; }
  }
  for (int base = e0; base < e1; base += kChunkMaxEdges) {
    LaneEdges le;
    load_lane_edges<F32>(bv, wd, base, e1, tid, le);
    if (base == e0) {
      stage_points(sh_X, pts, ch.lm0, nl, tid);
      if (staged) {
        for (int k = tid; k < n; k += kBlock) sh_x[k] = xp[k];
        stage_poses(sh_pose, poses, cams, wd.P, tid, kBlock);
      }
      __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
      const int pbase = base + p * kChunkEdges;
      if (pbase >= e1) break;     // uniform over the block
      const int e = pbase + tid;
      double c[3] = {0, 0, 0};
      if (e < e1 && le.ip[p] < wd.P) {
        const int ip = le.ip[p];
        double qt[7], R[9], cam[5], X[3], Xc[3], Q[6], g[3], rho0, x6[6];
        fetch_pose(staged, sh_pose, poses, cams, ip, qt, cam, R);
#pragma unroll
        for (int k = 0; k < 6; ++k) x6[k] = staged ? sh_x[6 * ip + k] : xp[6 * ip + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) X[k] = sh_X[(le.il[p] - ch.lm0) * 3 + k];
        const double rec[4] = {le.ra[p].x, le.ra[p].y, le.rb[p].x, le.rb[p].y};
        win_edge_core<KB8>(wd, bv, (size_t)wd.edge_off + e, rec, qt, cam, R, X, Xc, Q, g, rho0);
        dev::core_backsub<KB8>(Xc, Q, R, x6, c);
      }
      sh_c[tid] = c[0]; sh_c[kChunkEdges + tid] = c[1]; sh_c[2 * kChunkEdges + tid] = c[2];
      __syncthreads();
      if (tid < nl) {
        const int lo = max(my_lo, pbase), hi = min(my_hi, pbase + kChunkEdges);
        for (int x = lo; x < hi; ++x) {
          acc[0] += sh_c[x - pbase]; acc[1] += sh_c[kChunkEdges + x - pbase]; acc[2] += sh_c[2 * kChunkEdges + x - pbase];
        }
      }
      __syncthreads();
    }
  }
  double sc = 0.0;
  if (tid < nl) {
    const size_t gl = (size_t)wd.pt_off + ch.lm0 + tid;
    const double b0 = b3[0], b1 = b3[1], b2 = b3[2];
    const double c0 = b0 + acc[0], c1 = b1 + acc[1], c2 = b2 + acc[2];
    double xl[3];
    if (st.solve_ok) {
      // x_l = F (F^T c)
      const double t0 = F[0] * c0, t1 = F[1] * c0 + F[3] * c1, t2 = F[2] * c0 + F[4] * c1 + F[5] * c2;
      xl[0] = F[0] * t0 + F[1] * t1 + F[2] * t2;
      xl[1] = F[3] * t1 + F[4] * t2;
      xl[2] = F[5] * t2;
    } else {
      xl[0] = xl[1] = xl[2] = 0.0;
    }
    double* Xt = bv.pt_state[st.sel ^ 1] + gl * 3;
    Xt[0] = Xcur[0] + xl[0]; Xt[1] = Xcur[1] + xl[1]; Xt[2] = Xcur[2] + xl[2];
    sc = xl[0] * (lambda * xl[0] + b0) + xl[1] * (lambda * xl[1] + b1) + xl[2] * (lambda * xl[2] + b2);
  }
  sc = block_sum(sc, sh4);
  if (tid == 0) bv.scale_part[blockIdx.x] = sc;
  // ---- the trial residual of the chunk's edges (computeActiveErrors + activeRobustChi2 at the trial estimates, what k_residual did in
  // a pass of its own until round 3): the chunk's trial landmarks are in this block's registers, the trial poses were written by
  // k_solve before this launch.  Same lane <-> edge mapping and the same sums as k_residual: the chunk's chi2 is the same number.
  __syncthreads();
  if (tid < nl) {
    const double* Xt = bv.pt_state[st.sel ^ 1] + ((size_t)wd.pt_off + ch.lm0 + tid) * 3;   // (just written by this thread)
    sh_X[tid * 3] = Xt[0]; sh_X[tid * 3 + 1] = Xt[1]; sh_X[tid * 3 + 2] = Xt[2];
  }
  const double* poses_t = bv.pose_state[st.sel ^ 1] + (size_t)wd.pose_off * 7;
  const bool staged_t = (wd.P + wd.F) <= kLdsPoses;
  double* sh_pose_t = sh_pose + (staged ? wd.P * kPoseRec : 0);
  if (staged_t) stage_poses(sh_pose_t, poses_t, cams, wd.P + wd.F, tid, kBlock);
  __syncthreads();
  double chi_acc = 0.0;
  for (int base = e0; base < e1; base += kChunkMaxEdges) {
    LaneEdges le;
    load_lane_edges<F32>(bv, wd, base, e1, tid, le);
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
      const int e = base + p * kChunkEdges + tid;
      if (e < e1) {
        double qt[7], cam[5], R[9], X[3];
        fetch_pose(staged_t, sh_pose_t, poses_t, cams, le.ip[p], qt, cam, R);
#pragma unroll
        for (int k = 0; k < 3; ++k) X[k] = sh_X[(le.il[p] - ch.lm0) * 3 + k];
        const double rec[4] = {le.ra[p].x, le.ra[p].y, le.rb[p].x, le.rb[p].y};
        double cl, cr;
        chi_acc += win_edge_rho<KB8>(wd, bv, (size_t)wd.edge_off + e, rec, qt, cam, R, X, cl, cr);
      }
    }
  }
  const double chi = block_sum(chi_acc, sh4);
  if (tid == 0) bv.chi_part[blockIdx.x] = chi;
}

// --------------------------------------------------------------------------------------------
// k_control: one wavefront per window.
//   phase 0 (after the landmark side of a linearisation): open the iteration (currentChi, lambda init).
//   phase 1 (after the trial residual): gain ratio, accept / reject, stop rules.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_control(BatchView bv, int phase) {
  const int w = blockIdx.x;
  const WinDesc& wd = bv.win[w];
  LmState& st = bv.lm[w];
  const int lane = threadIdx.x;
  // the count of active windows is taken by phase 1; phase 0 of the same round (an earlier launch, no increments of its own) clears it
  // (a hipMemsetAsync per round was a 4.4 us fill kernel of its own: 2 % of a single window's optimize())
  if (phase == 0 && w == 0 && lane == 0) *bv.n_active = 0;
  if (!st.active) return;
  // robust chi2 of the state evaluated last = sum of the partials (fixed order): chunk partials of k_lin_lm after a
  // linearisation, of k_residual after a trial
  double chi = 0.0;
  if (phase == 0) { for (int c = lane; c < wd.n_chunks; c += 64) chi += bv.chi_lin[wd.chunk_off + c]; }
  else { for (int c = lane; c < wd.n_chunks * kResidualSplit; c += 64) chi += bv.chi_part[(size_t)wd.chunk_off * kResidualSplit + c]; }
  chi = dev::wave_sum(chi);
  if (phase == 0) {
    if (!st.need_lin) return;
    double dm = 0.0;
    if (st.iter == 0) {
      for (int c = lane; c < wd.n_chunks; c += 64) dm = fmax(dm, bv.dmax_lin[wd.chunk_off + c]);
      for (int p = lane; p < wd.P; p += 64) dm = fmax(dm, bv.dmax_pose[wd.fpose_off + p]);
      dm = dev::wave_max(dm);
    }
    if (lane == 0) {
      st.currentChi = chi; st.tempChi = chi; st.iniChi = chi;
      if (st.iter == 0) {
        st.chi2_initial = chi;
        st.lambda = (wd.lambda_init > 0) ? wd.lambda_init : kTau * dm;  // computeLambdaInit
        st.ni = 2.0; st.nBad = 0;
        st.need_dl = 1;   // the landmark factors could not be formed before lambda was known
      }
      st.rho = 0.0; st.qmax = 0; st.need_lin = 0; st.lin_now = 1;
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
  st.lin_now = 0;
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
  st.need_dl = again ? 1 : 0;   // another trial of this iteration: same Hll, new lambda
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

// Column envelope of every window's reduced system, once per upload: block (i, j) of S is structurally non-zero when a landmark is
// seen by both keyframes, i.e. when the plan gave it a contribution (RBlk::count > 0); the diagonal always is (Hpp + lambda I).
// pose_lo[j] = smallest such i.  The factorisation of k_solve keeps this envelope and skips the tile columns outside it (ldlt_block.h).
__global__ __launch_bounds__(256) void k_schur_env(BatchView bv) {
  __shared__ int sh_off[4];
  const int w = blockIdx.x, tid = threadIdx.x;
  const WinDesc wd = bv.win[w];
  int off = 0;
  for (int k = tid; k < w; k += 256) { const int Pk = bv.win[k].P; off += Pk * (Pk + 1) / 2 + Pk; }
  for (int o = 32; o > 0; o >>= 1) off += __shfl_xor(off, o, 64);
  if ((tid & 63) == 0) sh_off[tid >> 6] = off;
  __syncthreads();
  const int base = sh_off[0] + sh_off[1] + sh_off[2] + sh_off[3];
  const int P = wd.P;
  for (int j = tid; j < P; j += 256) {
    int lo = j;
    for (int i = 0; i < j; ++i) {
      if (bv.rblk[base + i * P - i * (i - 1) / 2 + (j - i)].count > 0) { lo = i; break; }
    }
    bv.pose_lo[wd.fpose_off + j] = lo;
  }
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
  st.need_lin = st.active; st.lin_now = 0; st.need_dl = 0;
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
// levenberg.cpp:123-147) and isDepthPositive from the FINAL estimates (Optimizer.cc:1425), in the caller's edge order.
// A merged fisheye-rig edge reports to both of its caller edges.
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
    double qt[7], cam[5], X[3], rec[4];
#pragma unroll
    for (int k = 0; k < 5; ++k) cam[k] = bv.pose_cam[((size_t)wd.pose_off + ip) * 5 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) rec[k] = bv.e_rec[ge * 4 + k];
    double chi_l = 0.0, chi_r = 0.0;
    if (evaluated) {
      const int s = st.last_eval_sel;
#pragma unroll
      for (int k = 0; k < 7; ++k) qt[k] = bv.pose_state[s][((size_t)wd.pose_off + ip) * 7 + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) X[k] = bv.pt_state[s][((size_t)wd.pt_off + il) * 3 + k];
      double Re[9];
      dev::quat_to_R(qt, Re);
      (void)win_edge_rho<KB8>(wd, bv, ge, rec, qt, cam, Re, X, chi_l, chi_r);
    }
    const int f = st.sel;
#pragma unroll
    for (int k = 0; k < 7; ++k) qt[k] = bv.pose_state[f][((size_t)wd.pose_off + ip) * 7 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) X[k] = bv.pt_state[f][((size_t)wd.pt_off + il) * 3 + k];
    double rot[3];
    dev::quat_rotate(qt, X, rot);
    const double Xl[3] = {rot[0] + qt[4], rot[1] + qt[5], rot[2] + qt[6]};
    const size_t go = (size_t)wd.out_off + bv.e_orig[ge];
    int kind = kKindMono;
    if (KB8 && wd.kb8_on) kind = bv.e_kind[ge];
    double zr = 0.0;
    if (KB8 && kind >= kKindBody) {   // isDepthPositive of the body edge: ((mTrl * T).map(X))(2) > 0
      double Xr[3];
      dev::quat_rotate(wd.trl, Xl, Xr);
      zr = Xr[2] + wd.trl[6];
    }
    if (kind == kKindBody) {
      bv.out_chi2[go] = chi_r;
      bv.out_depth[go] = zr > 0.0 ? 1 : 0;
    } else {
      bv.out_chi2[go] = chi_l;
      bv.out_depth[go] = (Xl[2] > 0.0) ? 1 : 0;
      if (kind == kKindBoth) {
        const size_t go2 = (size_t)wd.out_off + bv.e_orig2[ge];
        bv.out_chi2[go2] = chi_r;
        bv.out_depth[go2] = zr > 0.0 ? 1 : 0;
      }
    }
  }
}

// Debug / parity aid (osh_lba_linearize): the 6x3 blocks Hpl = D^T Q R of the optimisable-pose edges, which the
// optimisation itself never stores, in sorted edge order.
template <bool KB8>
__global__ __launch_bounds__(kBlock) void k_debug_hpl(BatchView bv, double* out) {
  const Chunk ch = bv.chunks[blockIdx.x];
  const WinDesc wd = bv.win[ch.win];
  const LmView st = lm_view(bv.lm, ch.win);
  const int* lmo = bv.lm_off + wd.lmoff_off;
  const int e0 = lmo[ch.lm0], e1 = lmo[ch.lm1];
  const double* poses = bv.pose_state[st.sel] + (size_t)wd.pose_off * 7;
  const double* cams = bv.pose_cam + (size_t)wd.pose_off * 5;
  const double* pts = bv.pt_state[st.sel] + (size_t)wd.pt_off * 3;
  for (int e = e0 + threadIdx.x; e < e1; e += kBlock) {
    const size_t ge = (size_t)wd.edge_off + e;
    const int ip = bv.e_pose[ge], il = bv.e_point[ge];
    double W[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) W[k] = 0.0;
    if (ip < wd.P) {
      double qt[7], cam[5], R[9], X[3], rec[4], Xc[3], Q[6], g[3], rho0;
      fetch_pose(false, nullptr, poses, cams, ip, qt, cam, R);
#pragma unroll
      for (int k = 0; k < 3; ++k) X[k] = pts[(size_t)il * 3 + k];
#pragma unroll
      for (int k = 0; k < 4; ++k) rec[k] = bv.e_rec[ge * 4 + k];
      win_edge_core<KB8>(wd, bv, ge, rec, qt, cam, R, X, Xc, Q, g, rho0);
      const double ident[6] = {1.0, 0.0, 0.0, 1.0, 0.0, 1.0};
      dev::core_WF<KB8>(Xc, Q, R, ident, W);
    }
#pragma unroll
    for (int k = 0; k < 18; ++k) out[ge * 18 + k] = W[k];
  }
}

}  // namespace osh

// =============================================================================================
// Host driver
// =============================================================================================
using namespace osh;

namespace osh {

// Final estimates in the caller's order, contiguous per batch, so that the download is a handful of large copies:
// optimisable poses of every window (the state buffer the controller ended on) and the landmarks with the upload's
// renumbering undone.
__global__ __launch_bounds__(256) void k_gather_out(BatchView bv, const int* lm_perm, int n_points_total, double* out_pose, double* out_pts,
                                                    const int* pt_win) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_points_total) {
    const int w = pt_win[t];
    const WinDesc& wd = bv.win[w];
    const int sel = bv.lm[w].sel;
    const int jn = t - wd.pt_off;
    const size_t dst = (size_t)wd.pt_off + lm_perm[t];
#pragma unroll
    for (int k = 0; k < 3; ++k) out_pts[dst * 3 + k] = bv.pt_state[sel][(size_t)(wd.pt_off + jn) * 3 + k];
  }
  if (t < bv.n_fposes) {
    const int w = bv.fpose_win[t];
    const WinDesc& wd = bv.win[w];
    const int sel = bv.lm[w].sel;
    const int i = t - wd.fpose_off;
#pragma unroll
    for (int k = 0; k < 7; ++k) out_pose[(size_t)t * 7 + k] = bv.pose_state[sel][((size_t)wd.pose_off + i) * 7 + k];
  }
}

// float32 observation records -> the 32-byte records the kernels read (lba_pack.h: rec_f32)
__global__ __launch_bounds__(256) void k_widen_rec(const float4* src, double* dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = src[i];
  double2* d = reinterpret_cast<double2*>(dst + i * 4);
  d[0] = make_double2((double)v.x, (double)v.y);
  d[1] = make_double2((double)v.z, (double)v.w);
}

}  // namespace osh

struct osh_lba_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  void* attach[4] = {nullptr, nullptr, nullptr, nullptr};
  void (*attach_free[4])(void*) = {nullptr, nullptr, nullptr, nullptr};
  KernelTimer timer;
  // host-side batch description
  int n_windows = 0;
  PackedBatch pb;
  std::vector<const volatile unsigned char*> stop_ptr;
  bool any_stop = false;
  int solve_nb = 24, solve_W = 0, solve_threads = kSolveThreadsLatency;
  bool solve_big = false;            // a window's reduced system exceeds the LDS-resident factorisation: big_solve.h
  DevBuf d_bigV, d_bigz, d_bigfail;
  size_t solve_lds = 0, backsub_lds = 0, lin_lds = 0, resid_lds = 0;
  double upload_pack_ms = 0, upload_copy_ms = 0;
  // edge kernels of the batch's camera models: the KannalaBrandt8 instantiations only when a window asks for them
  void (*kp_lin_lm)(BatchView) = nullptr;
  void (*kp_schur_sym)(BatchView, int, int) = nullptr;
  void (*kp_schur_cross)(BatchView, int, int) = nullptr;
  void (*kp_residual)(BatchView) = nullptr;
  void (*kp_finalize)(BatchView) = nullptr;
  void (*kp_backsub)(BatchView) = nullptr;
  void (*kp_debug_hpl)(BatchView, double*) = nullptr;
  // staging + device buffers
  PinBuf h_arena[2], h_out;
  DevBuf d_arena[2], d_ptwin, d_erec;
  DevBuf d_lm, d_pose[2], d_pt[2], d_hcontrib, d_chi_lin, d_dmax_lin, d_contrib, d_ccontrib;
  DevBuf d_Hll, d_bl, d_DL, d_Hpp, d_bp, d_S, d_bs, d_xp, d_chi, d_scale, d_dmaxp, d_nactive, d_out_chi2, d_out_depth, d_stop;
  DevBuf d_out_pose, d_out_pts, d_dbg, d_pose_lo;
  int* h_nactive = nullptr;          // pinned
  unsigned char* h_stop = nullptr;   // pinned [n_windows]
  size_t h_stop_cap = 0;
  BatchView bv{};
  bool optimized = false;
  DevPackState dpack;                // lba_pack_device.hip: the packer on the device
  int pack_mode = -1;                // -1: device unless the batch needs the host packer / the environment says otherwise; 0 device; 1 host
  bool device_packed = false;        // the last upload's arenas exist on the device only (pb.arena[] are null)
  template <class T> T* dsec(int s) const { return reinterpret_cast<T*>(static_cast<unsigned char*>(d_arena[pb.sec_arena(s)].p) + pb.sec_off(s)); }
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

// used by liba_device.hip / pose_device.hip: those paths share the context's device and stream
extern "C" int osh_lba_stream(osh_lba_ctx* c, int* device, hipStream_t* stream) {
  if (!c) { set_error("null context"); return OSH_ERR_INVALID; }
  *device = c->device; *stream = c->stream;
  return OSH_OK;
}

// State another translation unit keeps with the context (liba_device.hip: its staging and work buffers): created on first use,
// released by osh_lba_destroy -- not by a thread_local destructor at process exit, when the HIP runtime may already be gone.
extern "C" void** osh_lba_attachment(osh_lba_ctx* c, int slot, void (*free_fn)(void*)) {
  if (!c || slot < 0 || slot >= 4) return nullptr;
  c->attach_free[slot] = free_fn;
  return &c->attach[slot];
}

extern "C" void osh_lba_destroy(osh_lba_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int k = 0; k < 4; ++k) if (c->attach[k] && c->attach_free[k]) { c->attach_free[k](c->attach[k]); c->attach[k] = nullptr; }
  c->timer.destroy();
  c->dpack.release_events();
  if (c->h_nactive) (void)hipHostFree(c->h_nactive);
  if (c->h_stop) (void)hipHostFree(c->h_stop);
  hipStream_t s = c->stream;
  delete c;   // DevBuf / PinBuf members release their memory
  if (s) (void)hipStreamDestroy(s);
}

extern "C" int osh_lba_upload(osh_lba_ctx* c, int32_t nw, const osh_lba_problem* pr) {
  if (!c || nw <= 0 || !pr) { set_error("osh_lba_upload: bad arguments"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  OSH_HIP(hipStreamSynchronize(c->stream));
  c->optimized = false;
  c->n_windows = 0;                  // stays 0 (nothing uploaded) unless this upload completes
  c->stop_ptr.assign(nw, nullptr);
  c->any_stop = false;
  for (int w = 0; w < nw; ++w) { c->stop_ptr[w] = pr[w].stop_flag; if (pr[w].stop_flag) c->any_stop = true; }

  // ---- packing: on the device (lba_pack_device.hip) from the caller's arrays in the caller's order, or -- a handful of windows,
  // OSH_LBA_PACK=host -- by host threads straight into pinned staging (lba_pack.h) and one copy per arena
  const auto t0 = std::chrono::steady_clock::now();
  PackedBatch& pb = c->pb;
  hipStream_t s = c->stream;
  // Default rule: batches go to the device packer (one thread block per window: 512 windows pack in 6 ms of kernels against 65 ms of
  // 16 host threads); a handful of windows stay on the host, where one thread packs a 50-keyframe window in 1.3 ms -- below the
  // latency of three one-block kernels and two round trips (1.9 ms, profiles/pack_device.py).
  bool on_device = device_pack_supported(nw, pr);
  if (c->pack_mode >= 0) on_device = on_device && c->pack_mode == 0;
  else if (const char* e = std::getenv("OSH_LBA_PACK")) on_device = on_device && std::strcmp(e, "host") != 0;
  else on_device = on_device && nw >= kDevicePackMinWindows;
  c->device_packed = on_device;
  if (on_device) {
    const int rc = device_pack_batch(c->dpack, s, nw, pr, default_pack_threads(nw), pb, c->d_arena, c->d_ptwin);
    if (rc != OSH_OK) { set_error("%s", pb.msg); return rc; }
  } else {
    const int rc = pack_batch(nw, pr, [&](int which, size_t bytes) { return c->h_arena[which].reserve(bytes); }, default_pack_threads(nw), pb);
    if (rc != OSH_OK) { set_error("%s", pb.msg); return rc; }
  }
  const auto t1 = std::chrono::steady_clock::now();
  if (!on_device) {
    for (int a = 0; a < 2; ++a) {
      OSH_TRY(c->d_arena[a].reserve(pb.arena_bytes[a]));
      OSH_HIP(hipMemcpyAsync(c->d_arena[a].p, pb.arena[a], pb.arena_bytes[a], hipMemcpyHostToDevice, s));
    }
  }
  const size_t NP = pb.NP, NFP = pb.NFP, NL = pb.NL, NOUT = pb.NOUT;

  // ---- LDS budgets
  {
    const int staged = std::min(pb.np_max, kLdsPoses);
    c->lin_lds = (size_t)(9 * kChunkEdges + 4 + 3 * kBlock + staged * kPoseRec) * sizeof(double);
    c->resid_lds = (size_t)(4 + 3 * kBlock + staged * kPoseRec) * sizeof(double);
    const int pmax = pb.n_max / 6;
    c->backsub_lds = (size_t)(3 * kChunkEdges + 4 + 3 * kBlock + (pmax <= kLdsPoses ? pb.n_max + pmax * kPoseRec : 0) + staged * kPoseRec) * sizeof(double);
    // One block per window with the widest panel that fits.  A batch takes 256-thread blocks: the one LDS panel of ldlt_block.h
    // (70 KB for 50 keyframes) lets two windows share a CU, and the factorisation is latency bound, so they overlap (measured:
    // 5.5 ms of solve per 512-window step against 6.0 with 512-thread blocks, profiles/solve_sweep.sh); a few windows take 512 threads.
    c->solve_threads = nw >= 128 ? kSolveThreadsBatch : kSolveThreadsLatency;
    auto need = [&](int b) { return ldlt_lds_doubles(b, ldlt_row_stride(pb.n_max), c->solve_threads) * sizeof(double); };
    int nb = 24;
    while (nb > 6 && need(nb) > (size_t)150 * 1024) nb /= 2;  // 24 -> 12 -> 6 (template instantiations of k_solve)
    if (const char* e = std::getenv("OSH_LBA_SOLVE_THREADS")) { const int t = std::atoi(e); if (t == kSolveThreadsBatch || t == kSolveThreadsLatency) c->solve_threads = t; }
    if (const char* e = std::getenv("OSH_LBA_SOLVE_NB")) { const int b = std::atoi(e); if ((b == 24 || b == 12 || b == 6) && b <= nb) nb = b; }
    c->solve_big = need(nb) > (size_t)150 * 1024;
    if (c->solve_big) {
      // beyond ~240 optimisable poses (global BA of a long session) the system is factored in global memory, one window at a time
      if (pb.n_max / 6 > kBigMaxPoses) {
        set_error("window with %d optimisable poses: the dense reduced system is limited to %d poses", pb.n_max / 6, kBigMaxPoses);
        return OSH_ERR_UNSUPPORTED;
      }
      OSH_TRY(c->d_bigV.reserve((size_t)kBigNB * pb.n_max * 8)); OSH_TRY(c->d_bigz.reserve((size_t)pb.n_max * 8)); OSH_TRY(c->d_bigfail.reserve(sizeof(int)));
      nb = 6;
    }
    c->solve_nb = nb;
    c->solve_W = ldlt_row_stride(pb.n_max);
    c->solve_lds = c->solve_big ? 0 : need(nb);
  }

  // ---- work buffers
  auto R = [&](DevBuf& b, size_t bytes) { return b.reserve(std::max<size_t>(bytes, 8)); };
  OSH_TRY(R(c->d_lm, nw * sizeof(LmState)));
  for (int k = 0; k < 2; ++k) { OSH_TRY(R(c->d_pose[k], NP * 7 * 8)); OSH_TRY(R(c->d_pt[k], NL * 3 * 8)); }
  OSH_TRY(R(c->d_contrib, pb.n_contrib * 36 * 8)); OSH_TRY(R(c->d_ccontrib, pb.n_ccontrib * 6 * 8));
  OSH_TRY(R(c->d_hcontrib, pb.n_ccontrib * 27 * 8)); OSH_TRY(R(c->d_chi_lin, pb.n_chunks * 8)); OSH_TRY(R(c->d_dmax_lin, pb.n_chunks * 8));
  OSH_TRY(R(c->d_Hll, NL * 6 * 8)); OSH_TRY(R(c->d_bl, NL * 3 * 8)); OSH_TRY(R(c->d_DL, NL * 9 * 8));
  OSH_TRY(R(c->d_Hpp, NFP * 36 * 8)); OSH_TRY(R(c->d_bp, NFP * 6 * 8)); OSH_TRY(R(c->d_S, pb.S_total * 8));
  OSH_TRY(R(c->d_bs, NFP * 6 * 8)); OSH_TRY(R(c->d_xp, NFP * 6 * 8));
  OSH_TRY(R(c->d_chi, pb.n_chunks * kResidualSplit * 8)); OSH_TRY(R(c->d_scale, pb.n_chunks * 8));
  OSH_TRY(R(c->d_dmaxp, NFP * 8)); OSH_TRY(R(c->d_nactive, sizeof(int)));
  OSH_TRY(R(c->d_out_chi2, NOUT * 8)); OSH_TRY(R(c->d_out_depth, NOUT)); OSH_TRY(R(c->d_stop, nw));
  OSH_TRY(R(c->d_out_pose, NFP * 7 * 8)); OSH_TRY(R(c->d_out_pts, NL * 3 * 8)); OSH_TRY(R(c->d_ptwin, NL * 4)); OSH_TRY(R(c->d_pose_lo, NFP * 4));
  if (c->h_stop_cap < (size_t)nw) {
    if (c->h_stop) (void)hipHostFree(c->h_stop);
    OSH_HIP(hipHostMalloc((void**)&c->h_stop, nw));
    c->h_stop_cap = nw;
  }
  OSH_HIP(hipMemsetAsync(c->d_xp.p, 0, std::max<size_t>(NFP * 6 * 8, 8), s));
  if (!on_device) {
    // window of every landmark (for k_gather_out): the host packer writes it into the tail of the pinned output staging (reused
    // by the download later); the device packer has filled d_ptwin itself
    int* h_ptwin = static_cast<int*>(c->h_out.reserve(std::max<size_t>(NL * 4, 8)));
    if (!h_ptwin) { set_error("cannot allocate pinned staging"); return OSH_ERR_DEVICE; }
    for (int w = 0; w < nw; ++w) std::fill(h_ptwin + pb.win[w].pt_off, h_ptwin + pb.win[w].pt_off + pb.win[w].L, w);
    if (NL) OSH_HIP(hipMemcpyAsync(c->d_ptwin.p, h_ptwin, NL * 4, hipMemcpyHostToDevice, s));
  }

  BatchView& bv = c->bv;
  bv.n_windows = nw; bv.n_chunks = (int)pb.n_chunks; bv.n_fposes = (int)NFP;
  bv.win = c->dsec<WinDesc>(PackedBatch::WIN); bv.lm = c->d_lm.as<LmState>(); bv.chunks = c->dsec<Chunk>(PackedBatch::CHUNKS);
  bv.fpose_win = c->dsec<int>(PackedBatch::FPW);
  for (int k = 0; k < 2; ++k) { bv.pose_state[k] = c->d_pose[k].as<double>(); bv.pt_state[k] = c->d_pt[k].as<double>(); }
  bv.pose_cam = c->dsec<double>(PackedBatch::CAM);
  bv.e_pose = c->dsec<int>(PackedBatch::EPOSE); bv.e_point = c->dsec<int>(PackedBatch::EPOINT);
  bv.e_kind = c->dsec<unsigned char>(PackedBatch::EKIND);
  bv.e_rec = c->dsec<double>(PackedBatch::EREC); bv.e_rec2 = c->dsec<double>(PackedBatch::EREC2);
  if (pb.rec_f32) {
    OSH_TRY(R(c->d_erec, pb.NE * 32));
    if (pb.NE) {
      hipLaunchKernelGGL(k_widen_rec, dim3((unsigned)((pb.NE + 255) / 256)), dim3(256), 0, s, c->dsec<float4>(PackedBatch::EREC), c->d_erec.as<double>(), pb.NE);
      OSH_TRY(launch_check("k_widen_rec"));
    }
    bv.e_rec = c->d_erec.as<double>();
  }
  bv.e_rec32 = (pb.rec_f32 && !std::getenv("OSH_LBA_NO_F32_RECORDS")) ? c->dsec<float4>(PackedBatch::EREC) : nullptr;
  bv.e_orig = c->dsec<int>(PackedBatch::EORIG); bv.e_orig2 = c->dsec<int>(PackedBatch::EORIG2);
  bv.lm_off = c->dsec<int>(PackedBatch::LMOFF);
  bv.sitems = c->dsec<SItem>(PackedBatch::ITEMS); bv.srecs = c->dsec<SRec>(PackedBatch::RECS);
  bv.spair = c->dsec<int>(PackedBatch::SPAIR); bv.scslot = c->dsec<int>(PackedBatch::SCSLOT);
  bv.sposex = c->dsec<int>(PackedBatch::POSEX); bv.sposey = c->dsec<int>(PackedBatch::POSEY);
  bv.rblk = c->dsec<RBlk>(PackedBatch::RBLK); bv.n_rblk = (int)pb.n_rblk;
  bv.pose_crange = c->dsec<I2>(PackedBatch::CRANGE);
  bv.contrib = c->d_contrib.as<double>(); bv.ccontrib = c->d_ccontrib.as<double>(); bv.hcontrib = c->d_hcontrib.as<double>();
  bv.chi_lin = c->d_chi_lin.as<double>(); bv.dmax_lin = c->d_dmax_lin.as<double>();
  bv.Hll = c->d_Hll.as<double>(); bv.bl = c->d_bl.as<double>(); bv.DL = c->d_DL.as<double>();
  bv.Hpp = c->d_Hpp.as<double>(); bv.bp = c->d_bp.as<double>(); bv.S = c->d_S.as<double>();
  bv.bs = c->d_bs.as<double>(); bv.xp = c->d_xp.as<double>();
  bv.chi_part = c->d_chi.as<double>(); bv.scale_part = c->d_scale.as<double>();
  bv.dmax_pose = c->d_dmaxp.as<double>();
  bv.n_active = c->d_nactive.as<int>();
  bv.out_chi2 = c->d_out_chi2.as<double>(); bv.out_depth = c->d_out_depth.as<unsigned char>();
  bv.pose_lo = c->d_pose_lo.as<int>();
  hipLaunchKernelGGL(k_schur_env, dim3((unsigned)nw), dim3(256), 0, s, bv);
  OSH_TRY(launch_check("k_schur_env"));
  // dense factorisations (OSH_LBA_DENSE_SOLVE, and big_solve.h for maps beyond the LDS-resident solve) read and fill every block
  if (std::getenv("OSH_LBA_DENSE_SOLVE") || c->solve_big) bv.pose_lo = nullptr;
  else OSH_HIP(hipMemsetAsync(c->d_S.p, 0, std::max<size_t>(pb.S_total * 8, 8), s));   // the blocks above the envelope stay zero from here on
  OSH_HIP(hipStreamSynchronize(s));   // the staging arenas may be rewritten by the next upload
  const auto t2 = std::chrono::steady_clock::now();
  c->upload_pack_ms = on_device ? c->dpack.host_ms : std::chrono::duration<double, std::milli>(t1 - t0).count();
  c->upload_copy_ms = (on_device ? c->dpack.device_ms : 0.0) + std::chrono::duration<double, std::milli>(t2 - t1).count();

  const bool f32 = bv.e_rec32 != nullptr;
  if (pb.has_kb8) {
    c->kp_lin_lm = f32 ? k_lin_lm<true, true> : k_lin_lm<true, false>; c->kp_schur_sym = k_schur_fused<true, true>; c->kp_schur_cross = k_schur_fused<false, true>;
    c->kp_residual = f32 ? k_residual<true, true> : k_residual<true, false>; c->kp_finalize = k_finalize<true>;
    c->kp_backsub = f32 ? k_backsub<true, true> : k_backsub<true, false>; c->kp_debug_hpl = k_debug_hpl<true>;
  } else {
    c->kp_lin_lm = f32 ? k_lin_lm<false, true> : k_lin_lm<false, false>; c->kp_schur_sym = k_schur_fused<true, false>; c->kp_schur_cross = k_schur_fused<false, false>;
    c->kp_residual = f32 ? k_residual<false, true> : k_residual<false, false>; c->kp_finalize = k_finalize<false>;
    c->kp_backsub = f32 ? k_backsub<false, true> : k_backsub<false, false>; c->kp_debug_hpl = k_debug_hpl<false>;
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
#define OSH_LM_ATTR(K) \
    OSH_HIP(hipFuncSetAttribute((const void*)K<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); \
    OSH_HIP(hipFuncSetAttribute((const void*)K<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); \
    OSH_HIP(hipFuncSetAttribute((const void*)K<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); \
    OSH_HIP(hipFuncSetAttribute((const void*)K<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    OSH_LM_ATTR(k_backsub) OSH_LM_ATTR(k_lin_lm) OSH_LM_ATTR(k_residual)
#undef OSH_LM_ATTR
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
  if (c->solve_big) {
    for (int w = 0; w < c->n_windows; ++w) {
      const WinDesc& d = c->pb.win[w];
      if (d.n == 0) continue;
      BigSolve g;
      g.A = c->d_S.as<double>() + d.S_off; g.b = c->d_bs.as<double>() + (size_t)d.fpose_off * 6;
      g.V = c->d_bigV.as<double>(); g.z = c->d_bigz.as<double>(); g.fail = c->d_bigfail.as<int>();
      g.active = &(c->d_lm.as<LmState>() + w)->active;
      g.n = d.n;
      OSH_HIP(hipMemsetAsync(g.fail, 0, sizeof(int), s));
      for (int k0 = 0; k0 < d.n; k0 += kBigNB) {
        hipLaunchKernelGGL(k_big_diag, dim3(1), dim3(64), 0, s, g, k0);
        const int tr = d.n - (k0 + kBigNB);   // trailing rows
        if (tr <= 0) break;
        hipLaunchKernelGGL(k_big_panel, dim3((unsigned)((tr + 255) / 256)), dim3(256), 0, s, g, k0);
        const unsigned T = (unsigned)((tr + kBigTile - 1) / kBigTile);
        hipLaunchKernelGGL(k_big_update, dim3(T, T), dim3(256), 0, s, g, k0);
      }
      hipLaunchKernelGGL(k_big_back, dim3(1), dim3(1024), 0, s, g);
      hipLaunchKernelGGL(k_big_finish, dim3(1), dim3(256), 0, s, c->bv, w, g.fail);
    }
    return launch_check("k_big_solve");
  }
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
  const PackedBatch& pb = c->pb;
  for (int k = 0; k < 2; ++k) {
    if (pb.NP) OSH_HIP(hipMemcpyAsync(c->d_pose[k].p, c->dsec<double>(PackedBatch::POSE), pb.NP * 7 * 8, hipMemcpyDeviceToDevice, s));
    if (pb.NL) OSH_HIP(hipMemcpyAsync(c->d_pt[k].p, c->dsec<double>(PackedBatch::PT), pb.NL * 3 * 8, hipMemcpyDeviceToDevice, s));
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

// The first linearisation of optimize(): landmark side, pose side alone (computeLambdaInit needs the diagonal of Hpp before
// any Schur product can be formed), controller phase 0 (lambda), then the landmark factors for that lambda.
static int first_linearisation(osh_lba_ctx* c) {
  hipStream_t s = c->stream;
  const PackedBatch& pb = c->pb;
  LAUNCH(OSH_K_LINEARIZE, c->kp_lin_lm, pb.n_chunks, kBlock, c->lin_lds, c->bv);
  LAUNCH(OSH_K_LIN_POSE, c->kp_schur_sym, pb.n_sym, 64, 0, c->bv, 0, 0);
  LAUNCH(OSH_K_POSE_HESS, k_pose_reduce, (pb.NFP + 1) / 2, 64, 0, c->bv, 0);
  LAUNCH(OSH_K_CONTROL, k_control, c->n_windows, 64, 0, c->bv, 0);
  LAUNCH(OSH_K_LIN_AUX, c->kp_lin_lm, pb.n_chunks, kBlock, c->lin_lds, c->bv);
  return OSH_OK;
}

// One LM trial of every active window from the Schur products to the trial residual.
static int trial_kernels(osh_lba_ctx* c, bool later_round) {
  hipStream_t s = c->stream;
  const PackedBatch& pb = c->pb;
  LAUNCH(OSH_K_SCHUR, c->kp_schur_sym, pb.n_sym, 64, 0, c->bv, 0, 1);
  LAUNCH(OSH_K_SCHUR_CROSS, c->kp_schur_cross, pb.n_items - pb.n_sym, 64, 0, c->bv, (int)pb.n_sym, 1);
  if (later_round) LAUNCH(OSH_K_POSE_HESS, k_pose_reduce, (pb.NFP + 1) / 2, 64, 0, c->bv, 1);
  if (c->n_windows <= 8) { LAUNCH(OSH_K_SCHUR_REDUCE, k_schur_reduce<true>, (pb.n_rblk + 6) / 7, 256, 0, c->bv); } else { LAUNCH(OSH_K_SCHUR_REDUCE, k_schur_reduce<false>, (pb.n_rblk + 6) / 7, 256, 0, c->bv); }
  {
    const bool _t = c->timer.begin(OSH_K_SOLVE, s);
    OSH_TRY(launch_solve(c, s));
    if (_t) c->timer.end(s);
  }
  LAUNCH(OSH_K_BACKSUB, c->kp_backsub, pb.n_chunks, kBlock, c->backsub_lds, c->bv);
  return OSH_OK;
}

extern "C" int osh_lba_optimize(osh_lba_ctx* c) {
  if (!c || c->n_windows <= 0) { set_error("osh_lba_optimize: nothing uploaded"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const PackedBatch& pb = c->pb;
  if (c->timer.enabled) OSH_TRY(c->timer.init());
  OSH_TRY(reset_state(c));
  int n_active = 0;
  OSH_TRY(read_nactive(c, &n_active));
  int max_iter = 0;
  for (const WinDesc& d : pb.win) max_iter = std::max(max_iter, d.max_iter);
  const long max_rounds = (long)max_iter * kMaxTrials + 1;
  for (long round = 0; round < max_rounds && n_active > 0; ++round) {
    if (round == 0) {
      OSH_TRY(first_linearisation(c));
    } else {
      // windows opening an iteration: landmark side + factors; windows repeating a trial: factors for the new lambda
      LAUNCH(OSH_K_LINEARIZE, c->kp_lin_lm, pb.n_chunks, kBlock, c->lin_lds, c->bv);
      LAUNCH(OSH_K_CONTROL, k_control, c->n_windows, 64, 0, c->bv, 0);
    }
    OSH_TRY(trial_kernels(c, round > 0));   // (k_backsub ends with the trial residual of its chunk: no k_residual pass)
    if (c->any_stop) {
      // terminate() is polled after every trial (levenberg.cpp:149) and before every iteration
      snapshot_stop(c);
      OSH_HIP(hipMemcpyAsync(c->d_stop.p, c->h_stop, c->n_windows, hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(k_set_stop, dim3((c->n_windows + 63) / 64), dim3(64), 0, s, c->bv, c->d_stop.as<unsigned char>());
      OSH_TRY(launch_check("k_set_stop"));
    }
    LAUNCH(OSH_K_CONTROL, k_control, c->n_windows, 64, 0, c->bv, 1);   // (counts the windows still active; cleared by this round's phase 0)
    // A round is one trial; an iteration takes at least one, so no window can finish before `max_iter` rounds unless it converges
    // early -- and every kernel of a round leaves at once for a window that is no longer active.  The first max_iter rounds are
    // therefore queued without waiting for the count of active windows (ten host round trips of an optimize(10) were 0.2 ms of a
    // single window's 2.9 ms); the count is read after them, and after every further round (windows whose trials were rejected).
    // With a stop flag to poll the host looks in after every round, as before.
    if (c->any_stop || round + 1 >= (long)max_iter) {
      OSH_TRY(read_nactive(c, &n_active));
      if (c->timer.enabled) c->timer.collect();
    }
  }
  if (pb.n_chunks) {
    hipLaunchKernelGGL(c->kp_finalize, dim3((unsigned)pb.n_chunks), dim3(kBlock), 0, s, c->bv);
    OSH_TRY(launch_check("k_finalize"));
  }
  {
    const size_t n = std::max(pb.NL, pb.NFP);
    if (n) {
      hipLaunchKernelGGL(k_gather_out, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->bv, c->dsec<int>(PackedBatch::LMPERM), (int)pb.NL,
                         c->d_out_pose.as<double>(), c->d_out_pts.as<double>(), c->d_ptwin.as<int>());
      OSH_TRY(launch_check("k_gather_out"));
    }
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
  const PackedBatch& pb = c->pb;
  // one pinned staging buffer, five large copies, then per-window memcpy into the caller's arrays
  bool want_pose = false, want_pts = false, want_chi = false, want_depth = false;
  for (int w = 0; w < nw; ++w) { want_pose |= res[w].pose_qt != nullptr; want_pts |= res[w].points != nullptr; want_chi |= res[w].edge_chi2 != nullptr; want_depth |= res[w].edge_depth_pos != nullptr; }
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_lm = 0, o_pose = o_lm + al((size_t)nw * sizeof(LmState)), o_pts = o_pose + al(pb.NFP * 7 * 8), o_chi = o_pts + al(pb.NL * 3 * 8),
               o_depth = o_chi + al(pb.NOUT * 8), total = o_depth + al(pb.NOUT);
  unsigned char* h = static_cast<unsigned char*>(c->h_out.reserve(total));
  if (!h) { set_error("cannot allocate %zu bytes of pinned staging", total); return OSH_ERR_DEVICE; }
  OSH_HIP(hipMemcpyAsync(h + o_lm, c->d_lm.p, (size_t)nw * sizeof(LmState), hipMemcpyDeviceToHost, s));
  if (want_pose && pb.NFP) OSH_HIP(hipMemcpyAsync(h + o_pose, c->d_out_pose.p, pb.NFP * 7 * 8, hipMemcpyDeviceToHost, s));
  if (want_pts && pb.NL) OSH_HIP(hipMemcpyAsync(h + o_pts, c->d_out_pts.p, pb.NL * 3 * 8, hipMemcpyDeviceToHost, s));
  if (want_chi && pb.NOUT) OSH_HIP(hipMemcpyAsync(h + o_chi, c->d_out_chi2.p, pb.NOUT * 8, hipMemcpyDeviceToHost, s));
  if (want_depth && pb.NOUT) OSH_HIP(hipMemcpyAsync(h + o_depth, c->d_out_depth.p, pb.NOUT, hipMemcpyDeviceToHost, s));
  OSH_HIP(hipStreamSynchronize(s));
  const LmState* h_lm = reinterpret_cast<const LmState*>(h + o_lm);
  auto copy_window = [&](int w) {
    const WinDesc& d = pb.win[w];
    const LmState& st = h_lm[w];
    osh_lba_result& r = res[w];
    if (r.pose_qt && d.P) std::memcpy(r.pose_qt, h + o_pose + (size_t)d.fpose_off * 7 * 8, (size_t)d.P * 7 * 8);
    if (r.points && d.L) std::memcpy(r.points, h + o_pts + (size_t)d.pt_off * 3 * 8, (size_t)d.L * 3 * 8);
    if (r.edge_chi2 && d.in_edges) std::memcpy(r.edge_chi2, h + o_chi + (size_t)d.out_off * 8, (size_t)d.in_edges * 8);
    if (r.edge_depth_pos && d.in_edges) std::memcpy(r.edge_depth_pos, h + o_depth + (size_t)d.out_off, (size_t)d.in_edges);
    r.status = OSH_OK; r.iterations = st.iterations; r.trials = st.trials; r.n_trace = st.n_trace;
    r.chi2_initial = st.chi2_initial;
    for (int k = 0; k < st.n_trace; ++k) { r.chi2_trace[k] = st.chi2_trace[k]; r.lambda_trace[k] = st.lambda_trace[k]; r.trials_trace[k] = st.trials_trace[k]; }
  };
  const int n_threads = default_pack_threads(nw);
  if (n_threads <= 1) {
    for (int w = 0; w < nw; ++w) copy_window(w);
  } else {
    std::atomic<int> next{0};
    auto worker = [&]() { for (int w = next.fetch_add(1); w < nw; w = next.fetch_add(1)) copy_window(w); };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (std::thread& t : pool) t.join();
  }
  return OSH_OK;
}

extern "C" int osh_lba_solve(osh_lba_ctx* c, int32_t nw, const osh_lba_problem* pr, osh_lba_result* res) {
  OSH_TRY(osh_lba_upload(c, nw, pr));
  OSH_TRY(osh_lba_optimize(c));
  return osh_lba_download(c, nw, res);
}

// Host view of an int section of arena 0: the host packer's staging, or a copy from the device when the batch was packed there.
static int host_ints(osh_lba_ctx* c, int sec, size_t count, std::vector<int>& store, const int*& out) {
  if (!c->device_packed) { out = c->pb.sec<int>(sec); return OSH_OK; }
  store.resize(std::max<size_t>(count, 1));
  if (count) OSH_HIP(hipMemcpy(store.data(), c->dsec<int>(sec), count * 4, hipMemcpyDeviceToHost));
  out = store.data();
  return OSH_OK;
}

// Debug / parity aid: one linearisation of `window` at the uploaded estimates.
extern "C" int osh_lba_linearize(osh_lba_ctx* c, int32_t window, double* Hpp, double* bp, double* Hll, double* bl,
                                 double* Hpl, double* chi2, double* robust_chi2) {
  if (!c || window < 0 || window >= c->n_windows) { set_error("osh_lba_linearize: bad window"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const PackedBatch& pb = c->pb;
  OSH_TRY(reset_state(c));
  OSH_TRY(first_linearisation(c));
  OSH_HIP(hipStreamSynchronize(s));
  const WinDesc& d = pb.win[window];
  std::vector<LmState> h_lm(c->n_windows);
  OSH_HIP(hipMemcpy(h_lm.data(), c->d_lm.p, c->n_windows * sizeof(LmState), hipMemcpyDeviceToHost));
  if (robust_chi2) *robust_chi2 = h_lm[window].chi2_initial;
  if (Hpp && d.P) OSH_HIP(hipMemcpy(Hpp, c->d_Hpp.as<double>() + (size_t)d.fpose_off * 36, (size_t)d.P * 36 * 8, hipMemcpyDeviceToHost));
  if (bp && d.P) OSH_HIP(hipMemcpy(bp, c->d_bp.as<double>() + (size_t)d.fpose_off * 6, (size_t)d.P * 6 * 8, hipMemcpyDeviceToHost));
  std::vector<int> perm_store, eo_store, eo2_store;
  const int* perm = nullptr;   // new -> caller landmark index
  OSH_TRY(host_ints(c, PackedBatch::LMPERM, pb.NL, perm_store, perm));
  perm += d.pt_off;
  if (bl && d.L) {
    std::vector<double> t((size_t)d.L * 3);
    OSH_HIP(hipMemcpy(t.data(), c->d_bl.as<double>() + (size_t)d.pt_off * 3, t.size() * 8, hipMemcpyDeviceToHost));
    for (int j = 0; j < d.L; ++j) std::memcpy(bl + (size_t)perm[j] * 3, &t[(size_t)j * 3], 24);
  }
  if (Hll && d.L) {
    std::vector<double> up((size_t)d.L * 6);
    OSH_HIP(hipMemcpy(up.data(), c->d_Hll.as<double>() + (size_t)d.pt_off * 6, up.size() * 8, hipMemcpyDeviceToHost));
    for (int j = 0; j < d.L; ++j) {
      const double* u = &up[(size_t)j * 6];
      double* o = Hll + (size_t)perm[j] * 9;
      o[0] = u[0]; o[1] = u[1]; o[2] = u[2]; o[3] = u[1]; o[4] = u[3]; o[5] = u[4]; o[6] = u[2]; o[7] = u[4]; o[8] = u[5];
    }
  }
  if (Hpl && d.in_edges) {
    OSH_TRY(c->d_dbg.reserve(std::max<size_t>(pb.NE * 18 * 8, 8)));
    hipLaunchKernelGGL(c->kp_debug_hpl, dim3((unsigned)pb.n_chunks), dim3(kBlock), 0, s, c->bv, c->d_dbg.as<double>());
    OSH_TRY(launch_check("k_debug_hpl"));
    OSH_HIP(hipStreamSynchronize(s));
    std::vector<double> hs((size_t)d.E * 18);
    if (d.E) OSH_HIP(hipMemcpy(hs.data(), c->d_dbg.as<double>() + (size_t)d.edge_off * 18, hs.size() * 8, hipMemcpyDeviceToHost));
    std::memset(Hpl, 0, (size_t)d.in_edges * 18 * 8);
    const int *eo = nullptr, *eo2 = nullptr;
    OSH_TRY(host_ints(c, PackedBatch::EORIG, pb.NE, eo_store, eo));
    eo += d.edge_off;
    if (pb.has_rig) { OSH_TRY(host_ints(c, PackedBatch::EORIG2, pb.NE, eo2_store, eo2)); eo2 += d.edge_off; }
    // the two edges of a merged fisheye-rig pair share one Hessian block: both report it (their sum)
    for (int x = 0; x < d.E; ++x) {
      std::memcpy(Hpl + (size_t)eo[x] * 18, &hs[(size_t)x * 18], 18 * 8);
      if (eo2 && eo2[x] >= 0) std::memcpy(Hpl + (size_t)eo2[x] * 18, &hs[(size_t)x * 18], 18 * 8);
    }
  }
  if (chi2 && d.in_edges) {
    // mark every window evaluated so k_finalize emits chi2 of the current state
    for (auto& st : h_lm) { st.iterations = 1; st.last_eval_sel = st.sel; }
    OSH_HIP(hipMemcpy(c->d_lm.p, h_lm.data(), c->n_windows * sizeof(LmState), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(c->kp_finalize, dim3((unsigned)pb.n_chunks), dim3(kBlock), 0, s, c->bv);
    OSH_TRY(launch_check("k_finalize"));
    OSH_HIP(hipStreamSynchronize(s));
    OSH_HIP(hipMemcpy(chi2, c->d_out_chi2.as<double>() + d.out_off, (size_t)d.in_edges * 8, hipMemcpyDeviceToHost));
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
  const PackedBatch& pb = c->pb;
  OSH_TRY(reset_state(c));
  LAUNCH(OSH_K_LINEARIZE, c->kp_lin_lm, pb.n_chunks, kBlock, c->lin_lds, c->bv);
  LAUNCH(OSH_K_LIN_POSE, c->kp_schur_sym, pb.n_sym, 64, 0, c->bv, 0, 0);
  LAUNCH(OSH_K_POSE_HESS, k_pose_reduce, (pb.NFP + 1) / 2, 64, 0, c->bv, 0);
  LAUNCH(OSH_K_CONTROL, k_control, c->n_windows, 64, 0, c->bv, 0);
  OSH_HIP(hipStreamSynchronize(s));
  std::vector<LmState> h_lm(c->n_windows);
  OSH_HIP(hipMemcpy(h_lm.data(), c->d_lm.p, c->n_windows * sizeof(LmState), hipMemcpyDeviceToHost));
  for (auto& st : h_lm) { st.lambda = lambda; st.need_dl = 1; }
  OSH_HIP(hipMemcpy(c->d_lm.p, h_lm.data(), c->n_windows * sizeof(LmState), hipMemcpyHostToDevice));
  const WinDesc& d = pb.win[window];
  LAUNCH(OSH_K_LIN_AUX, c->kp_lin_lm, pb.n_chunks, kBlock, c->lin_lds, c->bv);
  LAUNCH(OSH_K_SCHUR, c->kp_schur_sym, pb.n_sym, 64, 0, c->bv, 0, 1);
  LAUNCH(OSH_K_SCHUR_CROSS, c->kp_schur_cross, pb.n_items - pb.n_sym, 64, 0, c->bv, (int)pb.n_sym, 1);
  if (c->n_windows <= 8) { LAUNCH(OSH_K_SCHUR_REDUCE, k_schur_reduce<true>, (pb.n_rblk + 6) / 7, 256, 0, c->bv); } else { LAUNCH(OSH_K_SCHUR_REDUCE, k_schur_reduce<false>, (pb.n_rblk + 6) / 7, 256, 0, c->bv); }
  OSH_HIP(hipStreamSynchronize(s));
  if (S && d.n) OSH_HIP(hipMemcpy(S, c->d_S.as<double>() + d.S_off, (size_t)d.n * d.n * 8, hipMemcpyDeviceToHost));
  if (bs && d.n) OSH_HIP(hipMemcpy(bs, c->d_bs.as<double>() + (size_t)d.fpose_off * 6, (size_t)d.n * 8, hipMemcpyDeviceToHost));
  OSH_TRY(launch_solve(c, s));
  hipLaunchKernelGGL(c->kp_backsub, dim3((unsigned)pb.n_chunks), dim3(kBlock), c->backsub_lds, s, c->bv);
  OSH_TRY(launch_check("k_backsub"));
  OSH_HIP(hipStreamSynchronize(s));
  if (x) {
    if (d.n) OSH_HIP(hipMemcpy(x, c->d_xp.as<double>() + (size_t)d.fpose_off * 6, (size_t)d.n * 8, hipMemcpyDeviceToHost));
    // x_l = X_trial - X_cur, back in the caller's landmark order
    std::vector<double> a((size_t)d.L * 3), b((size_t)d.L * 3);
    if (d.L) {
      std::vector<int> perm_store;
      const int* perm = nullptr;
      OSH_TRY(host_ints(c, PackedBatch::LMPERM, pb.NL, perm_store, perm));
      perm += d.pt_off;
      OSH_HIP(hipMemcpy(a.data(), c->d_pt[1].as<double>() + (size_t)d.pt_off * 3, a.size() * 8, hipMemcpyDeviceToHost));
      OSH_HIP(hipMemcpy(b.data(), c->d_pt[0].as<double>() + (size_t)d.pt_off * 3, b.size() * 8, hipMemcpyDeviceToHost));
      for (int j = 0; j < d.L; ++j)
        for (int k = 0; k < 3; ++k) x[d.n + (size_t)perm[j] * 3 + k] = a[(size_t)j * 3 + k] - b[(size_t)j * 3 + k];
    }
  }
  c->optimized = false;
  return OSH_OK;
}

// 0: pack uploads on the device (default), 1: on the host (lba_pack.h), -1: default
// rules (OSH_LBA_PACK=host in the environment selects the host packer)
extern "C" int osh_lba_set_pack_mode(osh_lba_ctx* c, int mode) {
  if (!c || mode < -1 || mode > 1) { set_error("osh_lba_set_pack_mode: bad arguments"); return OSH_ERR_INVALID; }
  c->pack_mode = mode;
  return OSH_OK;
}

// Test hook: packs the problems with the device packer AND the host packer and compares every section of the two layouts.
extern "C" int osh_lba_pack_compare(osh_lba_ctx* c, int32_t nw, const osh_lba_problem* pr, int64_t stats[4]) {
  if (!c || nw <= 0 || !pr) { set_error("osh_lba_pack_compare: bad arguments"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  OSH_HIP(hipStreamSynchronize(c->stream));
  return device_pack_compare(c->dpack, c->stream, nw, pr, stats);
}

extern "C" int osh_lba_set_profiling(osh_lba_ctx* c, int enable) {
  if (!c) return OSH_ERR_INVALID;
  OSH_HIP(hipSetDevice(c->device));
  if (enable) OSH_TRY(c->timer.init());
  c->timer.enabled = enable != 0;
  c->dpack.timing = enable != 0;
  c->timer.reset();
  return OSH_OK;
}

extern "C" int osh_lba_get_profile(osh_lba_ctx* c, int64_t launches[OSH_K_COUNT], double total_ms[OSH_K_COUNT]) {
  if (!c || !launches || !total_ms) return OSH_ERR_INVALID;
  for (int k = 0; k < OSH_K_COUNT; ++k) { launches[k] = c->timer.launches[k]; total_ms[k] = c->timer.total_ms[k]; }
  return OSH_OK;
}

extern "C" int osh_lba_get_plan_stats(osh_lba_ctx* c, int64_t stats[8]) {
  if (!c || !stats || c->n_windows <= 0) { set_error("osh_lba_get_plan_stats: nothing uploaded"); return OSH_ERR_INVALID; }
  stats[0] = (int64_t)c->pb.n_items; stats[1] = (int64_t)c->pb.n_sym; stats[2] = c->pb.tile_steps; stats[3] = c->pb.pair_blocks;
  stats[4] = (int64_t)c->pb.n_contrib; stats[5] = (int64_t)c->pb.n_rblk; stats[6] = (int64_t)c->pb.n_recs; stats[7] = (int64_t)c->pb.n_ccontrib;
  return OSH_OK;
}

// Device-side cost of the last upload packed on the device (HIP events; enable with osh_lba_set_profiling before the upload):
// ms[0] H2D of the staged problem, ms[1..3] k_pack_pre1 / k_pack_pre2 / k_pack_post, ms[4] staged bytes, ms[5] 1 if the batch was packed on the
// device, ms[6..29] shader-clock cycles per phase of the three kernels (mean over the windows)
extern "C" int osh_lba_get_pack_profile(osh_lba_ctx* c, double ms[30]) {
  if (!c || !ms) return OSH_ERR_INVALID;
  for (int k = 0; k < 4; ++k) ms[k] = c->dpack.ev_ms[k];
  ms[4] = (double)c->dpack.raw_bytes; ms[5] = c->device_packed ? 1.0 : 0.0;
  for (int k = 0; k < 24; ++k) ms[6 + k] = c->dpack.cyc_mean[k];
  return OSH_OK;
}

extern "C" int osh_lba_get_upload_times(osh_lba_ctx* c, double ms[2]) {
  if (!c || !ms) return OSH_ERR_INVALID;
  ms[0] = c->upload_pack_ms; ms[1] = c->upload_copy_ms;
  return OSH_OK;
}

extern "C" const char* osh_lba_kernel_name(int k) {
  // pinhole instantiations reading float32 records (<true, ..> for a fisheye batch, <.., false> when a record is not exact in float32)
  static const char* names[OSH_K_COUNT] = {"k_lin_lm<false, true>", "k_pose_reduce", "k_schur_fused<true, false> (mode 1)", "k_solve", "k_backsub<false, true>",
                                           "k_residual<false, true>", "k_control", "k_schur_reduce<false>", "k_schur_fused<false, false>",
                                           "k_lin_lm<false, true> (factors only)", "k_schur_fused<true, false> (mode 0)"};
  return (k >= 0 && k < OSH_K_COUNT) ? names[k] : "?";
}
