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

constexpr int kLT = 512;      // threads of the persistent block
constexpr int kLNB = 24;      // LDL^T panel width
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
  double* linkJ;              // [NL*(216+81+9)] J(9x24), W(9x9), -W r (9)
  double* out_chi2; unsigned char* out_depth;
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

// robust chi2 of the state in buffer `sel` (computeActiveErrors + activeRobustChi2, inertial edges first)
__device__ double eval_chi2(const LibaView& v, const LibaDesc& d, int sel, double* shw) {
  const int tid = threadIdx.x;
  const double* poses = v.pose[sel] + (size_t)d.pose_off * 24;
  const double* vba = v.vba[sel] + (size_t)d.vel_off * 9;
  const double* pts = v.pts[sel] + (size_t)d.pt_off * 3;
  double acc = 0.0;
  for (int l = tid; l < d.NL; l += kLT) {
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
  const int* lmo = v.lm_off + d.lmoff_off;
  for (int j = tid; j < d.L; j += kLT) {
    const double* X = pts + 3 * (size_t)j;
    for (int e = lmo[j]; e < lmo[j + 1]; ++e) {
      const size_t ge = (size_t)d.edge_off + e;
      const int kind = v.e_kind[ge];
      VisEval ev;
      vis_residual(d, kind, poses + 24 * (size_t)v.e_pose[ge], X, v.e_obs + ge * 3, v.e_info[ge], ev);
      double r0, r1;
      dev::huber(ev.chi2, kind == OSH_EDGE_STEREO ? d.huber_stereo : d.huber_mono, r0, r1);
      acc += r0;
    }
  }
  return blk_sum(acc, shw);
}

__global__ __launch_bounds__(kLT) void k_liba(LibaView v, int W) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const LibaDesc& d = v.desc[blockIdx.x];
  LibaOut& out = v.out[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = d.N, n = d.n, L = d.L;
  // LDS carve: [0, ldlt) the LDL^T scratch (reused as general scratch between solves), then control words
  double* shw = sh + liba_scratch_doubles(W);             // [kLT/64] reductions
  double* H = v.H + d.H_off; double* S = v.S + d.H_off;
  double* b = v.b + d.b_off; double* bs = v.bs + d.b_off; double* xg = v.x + d.b_off;
  double* Hpl = v.Hpl + (size_t)d.edge_off * 18;
  double* BD = v.BD + (size_t)d.edge_off * 18;
  double* Hll = v.Hll + (size_t)d.pt_off * 6; double* bl = v.bl + (size_t)d.pt_off * 3; double* dinv = v.dinv + (size_t)d.pt_off * 9;
  const int* lmo = v.lm_off + d.lmoff_off;
  const int* po = v.pel_off + d.peloff_off;
  const int* lmpe = v.lm_pose_edge + d.lmpose_off;
  double* linkJ = v.linkJ + (size_t)d.link_off * 306;

  int sel = 0, eval_sel = 0;
  double lambda = -1.0, ni = 2.0;
  int nBad = 0, cj = 0, trials_total = 0, n_trace = 0;
  bool ok = true;
  const double chi_init = eval_chi2(v, d, 0, shw);
  if (tid == 0) out.chi2_initial = chi_init;
  double last_chi = chi_init;   // activeRobustChi2() of the errors evaluated last (err_end)

  for (int it = 0; it < d.max_iter && ok; ++it) {
    const double* poses = v.pose[sel] + (size_t)d.pose_off * 24;
    const double* vba = v.vba[sel] + (size_t)d.vel_off * 9;
    const double* pts = v.pts[sel] + (size_t)d.pt_off * 3;
    double currentChi = eval_chi2(v, d, sel, shw);
    const double iniChi = currentChi;
    // ---------------------------------------------------------------- linearise (buildSystem)
    for (int k = tid; k < n * n; k += kLT) H[k] = 0.0;
    for (int k = tid; k < n; k += kLT) b[k] = 0.0;
    __syncthreads();
    // landmark side: thread per landmark, its edges in order
    for (int j = tid; j < L; j += kLT) {
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
    // pose side: one wavefront per optimisable pose (rows of H are disjoint)
    for (int i = wave; i < N; i += kLT / 64) {
      double Hp[21], bp[6];
#pragma unroll
      for (int k = 0; k < 21; ++k) Hp[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) bp[k] = 0.0;
      const double* pose = poses + 24 * (size_t)i;
      for (int idx = po[i] + lane; idx < po[i + 1]; idx += 64) {
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
        int m = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
          for (int c = a; c < 6; ++c) { Hp[m] += (Jp[a] * ww) * Jp[c] + (Jp[6 + a] * ww) * Jp[6 + c] + (Jp[12 + a] * ww) * Jp[12 + c]; ++m; }
          bp[a] += Jp[a] * wr[0] + Jp[6 + a] * wr[1] + Jp[12 + a] * wr[2];
        }
      }
#pragma unroll
      for (int k = 0; k < 21; ++k) Hp[k] = dev::wave_sum(Hp[k]);
#pragma unroll
      for (int k = 0; k < 6; ++k) bp[k] = dev::wave_sum(bp[k]);
      if (lane == 0) {
        int m = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int c = a; c < 6; ++c) { H[(size_t)(6 * i + a) * n + 6 * i + c] = Hp[m]; H[(size_t)(6 * i + c) * n + 6 * i + a] = Hp[m]; ++m; }
#pragma unroll
        for (int k = 0; k < 6; ++k) b[6 * i + k] = bp[k];
      }
    }
    // inertial links: Jacobians by one thread per link, then the quadratic forms link by link (fixed order)
    for (int l = tid; l < d.NL; l += kLT) {
      const int gl = d.link_off + l;
      const int a = v.link_prev[gl], c = v.link_cur[gl];
      double* J = linkJ + (size_t)l * 306;
      double* Wm = J + 216; double* wr = Wm + 81;
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
      for (int i = 0; i < 81; ++i) Wm[i] = rho1 * Om[i];
      for (int i = 0; i < 9; ++i) { double t = 0; for (int j = 0; j < 9; ++j) t += Om[i * 9 + j] * r[j]; wr[i] = -t * rho1; }
    }
    __syncthreads();
    for (int l = 0; l < d.NL; ++l) {
      const int gl = d.link_off + l;
      const int a = v.link_prev[gl], c = v.link_cur[gl];
      const double* J = linkJ + (size_t)l * 306;
      const double* Wm = J + 216; const double* wr = Wm + 81;
      // vertex of a Jacobian column
      auto vert_of = [](int col) { return col < 6 ? 0 : col < 9 ? 1 : col < 12 ? 2 : col < 15 ? 3 : col < 21 ? 4 : 5; };
      const int vbase[6] = {0, 6, 9, 12, 15, 21};
      for (int idx = tid; idx < 24 * 24; idx += kLT) {
        const int ca = idx / 24, cb = idx - ca * 24;
        const int va = vert_of(ca), vb = vert_of(cb);
        if (vb < va) continue;   // upper blocks + mirrored below
        const int oa = link_vertex_offset(va, a, c, N), ob = link_vertex_offset(vb, a, c, N);
        if (oa < 0 || ob < 0) continue;
        double acc = 0.0;
        for (int k = 0; k < 9; ++k) {
          double t = 0.0;
          for (int m = 0; m < 9; ++m) t += Wm[k * 9 + m] * J[m * 24 + cb];
          acc += J[k * 24 + ca] * t;
        }
        const int ra = oa + (ca - vbase[va]), rb = ob + (cb - vbase[vb]);
        H[(size_t)ra * n + rb] += acc;
        if (va != vb) H[(size_t)rb * n + ra] += acc;
      }
      for (int ca = tid; ca < 24; ca += kLT) {
        const int va = vert_of(ca);
        const int oa = link_vertex_offset(va, a, c, N);
        if (oa >= 0) { double t = 0; for (int k = 0; k < 9; ++k) t += J[k * 24 + ca] * wr[k]; b[oa + (ca - vbase[va])] += t; }
      }
      __syncthreads();
      // EdgeGyroRW / EdgeAccRW: r = b2 - b1, J = [-I, I], plain information
      if (tid < 18) {
        const int which = tid / 9, i = (tid % 9) / 3, j = tid % 3;
        const double* Og = (which == 0 ? v.link_info_g : v.link_info_a) + (size_t)gl * 9;
        const int o1 = (a < N) ? 6 * N + 9 * a + 3 + 3 * which : -1;
        const int o2 = 6 * N + 9 * c + 3 + 3 * which;
        const double g = Og[i * 3 + j];
        if (o1 >= 0) {
          H[(size_t)(o1 + i) * n + o1 + j] += g;
          H[(size_t)(o1 + i) * n + o2 + j] += -g;
          H[(size_t)(o2 + j) * n + o1 + i] += -g;
        }
        H[(size_t)(o2 + i) * n + o2 + j] += g;
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
      if (d.lambda_init > 0) lambda = d.lambda_init;
      else {
        double m = 0.0;
        for (int k = tid; k < n; k += kLT) m = fmax(m, fabs(H[(size_t)k * n + k]));
        for (int j = tid; j < L; j += kLT) m = fmax(m, fmax(fabs(Hll[(size_t)j * 6]), fmax(fabs(Hll[(size_t)j * 6 + 3]), fabs(Hll[(size_t)j * 6 + 5]))));
        lambda = 1e-5 * blk_max(m, shw);
      }
      ni = 2.0; nBad = 0;
    }
    // ---------------------------------------------------------------- LM trials
    double rho = 0.0;
    int qmax = 0;
    do {
      const int trs = sel ^ 1;
      // Dinv and db per landmark (setLambda on Hll, block_solver.hpp:389,582-587)
      for (int j = tid; j < L; j += kLT) {
        const double* hl = Hll + (size_t)j * 6;
        double Di[9];
        dev::inv3_sym(hl[0] + lambda, hl[1], hl[2], hl[3] + lambda, hl[4], hl[5] + lambda, Di);
        const double b0 = bl[(size_t)j * 3], b1 = bl[(size_t)j * 3 + 1], b2 = bl[(size_t)j * 3 + 2];
        double* o = dinv + (size_t)j * 9;
        o[0] = Di[0]; o[1] = Di[1]; o[2] = Di[2]; o[3] = Di[4]; o[4] = Di[5]; o[5] = Di[8];
        o[6] = Di[0] * b0 + Di[1] * b1 + Di[2] * b2; o[7] = Di[3] * b0 + Di[4] * b1 + Di[5] * b2; o[8] = Di[6] * b0 + Di[7] * b1 + Di[8] * b2;
      }
      // S = H + lambda I (upper), rhs = b
      for (int k = tid; k < n * n; k += kLT) { const int r = k / n, c = k - r * n; S[k] = H[k] + ((r == c) ? lambda : 0.0); }
      for (int k = tid; k < n; k += kLT) bs[k] = b[k];
      __syncthreads();
      // Schur complement (block_solver.hpp:381-432), landmark-parallel.  (a) BD = B Dinv of every optimisable-pose edge, one thread
      // per edge; (b) one wavefront per pose PAIR (i <= i2): the lanes stride the landmarks, a landmark seen by both poses adds
      // BD_i B_i2^T to the lane's 6x6 partial, the 36 partials are summed by a fixed butterfly and subtracted from S(i, i2);
      // (c) the rhs terms B (Dinv b_l), one wavefront per pose over its edges.  (The first version walked the edges of one pose
      // row per wavefront and looked every partner block up on the way: 48 % of the kernel, with 8 of 10 rows on the first pass
      // and two wavefronts on the second.)
      for (int i = 0; i < N; ++i)
        for (int idx = po[i] + tid; idx < po[i + 1]; idx += kLT) {
          const int e = v.pel_edge[(size_t)d.pel_off + idx];
          const int j = v.e_point[(size_t)d.edge_off + e];
          const double* Dj = dinv + (size_t)j * 9;
          const double* Be = Hpl + (size_t)e * 18;
          double* o = BD + (size_t)e * 18;
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            const double x0 = Be[r * 3], x1 = Be[r * 3 + 1], x2 = Be[r * 3 + 2];
            o[r * 3 + 0] = x0 * Dj[0] + x1 * Dj[1] + x2 * Dj[2];
            o[r * 3 + 1] = x0 * Dj[1] + x1 * Dj[3] + x2 * Dj[4];
            o[r * 3 + 2] = x0 * Dj[2] + x1 * Dj[4] + x2 * Dj[5];
          }
        }
      __syncthreads();
      const int npairs = N * (N + 1) / 2;
      for (int pr = wave; pr < npairs; pr += kLT / 64) {
        int i = 0, rem = pr;
        while (rem >= N - i) { rem -= N - i; ++i; }
        const int i2 = i + rem;
        double acc[36];
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
      }
      for (int i = wave; i < N; i += kLT / 64) {
        double ci[6] = {0, 0, 0, 0, 0, 0};
        for (int idx = po[i] + lane; idx < po[i + 1]; idx += 64) {
          const int e = v.pel_edge[(size_t)d.pel_off + idx];
          const int j = v.e_point[(size_t)d.edge_off + e];
          // second edge of a (keyframe, landmark) pair: its block lives in the first edge's slot
          if (e > lmo[j] && v.e_pose[(size_t)d.edge_off + e - 1] == i) continue;
          const double* Dj = dinv + (size_t)j * 9;
          const double* Be = Hpl + (size_t)e * 18;
#pragma unroll
          for (int r = 0; r < 6; ++r) ci[r] += Be[r * 3] * Dj[6] + Be[r * 3 + 1] * Dj[7] + Be[r * 3 + 2] * Dj[8];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) ci[r] = dev::wave_sum(ci[r]);
        if (lane < 6) {
          double c = ci[0];
          if (lane == 1) c = ci[1]; else if (lane == 2) c = ci[2]; else if (lane == 3) c = ci[3];
          else if (lane == 4) c = ci[4]; else if (lane == 5) c = ci[5];
          bs[6 * i + lane] -= c;
        }
      }
      __syncthreads();
      double *xs, *shw2;
      const bool ok2 = ldlt_solve_block<kLNB, kLT>(S, bs, n, W, sh, xs, shw2);
      for (int k = tid; k < n; k += kLT) xg[k] = xs[k];
      __syncthreads();
      // landmark back-substitution, point update, landmark part of computeScale
      double sc = 0.0;
      double* pts_t = v.pts[trs] + (size_t)d.pt_off * 3;
      for (int j = tid; j < L; j += kLT) {
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
      // pose / velocity / bias update into the trial buffers (ImuCamPose::Update, src/G2oTypes.cc:187-220)
      double* poses_t = v.pose[trs] + (size_t)d.pose_off * 24;
      double* vba_t = v.vba[trs] + (size_t)d.vel_off * 9;
      for (int k = tid; k < N; k += kLT) {
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
      const double scale_sum = blk_sum(sc, shw);   // (also the barrier before the trial errors)
      double tempChi = eval_chi2(v, d, trs, shw);
      last_chi = tempChi;
      eval_sel = trs;
      if (!ok2) tempChi = DBL_MAX;
      // controller: identical decisions in every thread (all inputs are block-uniform)
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
      __syncthreads();
    } while (rho < 0 && qmax < 10);
    ++cj;
    if (tid == 0 && n_trace < OSH_LBA_MAX_TRACE) { out.chi2_trace[n_trace] = currentChi; out.lambda_trace[n_trace] = lambda; out.trials_trace[n_trace] = qmax; }
    if (n_trace < OSH_LBA_MAX_TRACE) ++n_trace;
    if (qmax == 10 || rho == 0) { ok = false; continue; }
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) { ok = false; continue; }
  }
  if (tid == 0) { out.iterations = cj; out.trials = trials_total; out.n_trace = n_trace; out.sel = sel; out.chi2_final = last_chi; }
  // e->chi2() of the errors computeActiveErrors saw last (buffer eval_sel: stale after a rejected final trial) and
  // isDepthPositive() of the final estimates (src/Optimizer.cc:2861-2888; ImuCamPose::isDepthPositive, G2oTypes.cc:185-188)
  {
    const double* pe = v.pose[eval_sel] + (size_t)d.pose_off * 24;
    const double* xe = v.pts[eval_sel] + (size_t)d.pt_off * 3;
    const double* pf = v.pose[sel] + (size_t)d.pose_off * 24;
    const double* xf = v.pts[sel] + (size_t)d.pt_off * 3;
    for (int e = tid; e < d.E; e += kLT) {
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
}

}  // namespace osh

// =============================================================================================
// Host driver: osh_liba_solve (upload + one launch + download)
// =============================================================================================
using namespace osh;

namespace {
struct LibaBuffers {
  DevBuf desc, out, pose[2], vba[2], pts[2], e_pose, e_point, e_kind, e_obs, e_info, e_orig, lm_off, pel_off, pel_edge, lmpe,
      l_prev, l_cur, l_pre, l_info, l_ig, l_ia, l_rob, Hpl, BD, Hll, bl, dinv, H, b, S, bs, x, linkJ, o_chi2, o_depth;
};
LibaBuffers& liba_buffers() { static thread_local LibaBuffers b; return b; }
template <class T>
int up(DevBuf& b, const std::vector<T>& v, hipStream_t s) {
  int rc = b.reserve(std::max<size_t>(v.size(), 1) * sizeof(T));
  if (rc != OSH_OK) return rc;
  if (!v.empty()) OSH_HIP(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return OSH_OK;
}
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
  // ---- pack
  std::vector<double> h_pose(K * 24), h_vba(NV * 9), h_pts(L * 3), h_obs(E * 3), h_info(E);
  std::vector<int> h_ep(E), h_el(E), h_eo(E), h_lmo(LO), h_po(PO), h_pel(EF), h_lmpe(LP, -1), h_lp(NL), h_lc(NL);
  std::vector<unsigned char> h_kind(E), h_rob(NL);
  std::vector<float> h_pre(NL * OSH_PREINT_FLOATS);
  std::vector<double> h_li(NL * 81), h_lg(NL * 9), h_la(NL * 9);
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
  LibaBuffers& B = liba_buffers();
  OSH_HIP(hipStreamSynchronize(s));
  OSH_TRY(up(B.desc, h_desc, s));
  OSH_TRY(up(B.pose[0], h_pose, s)); OSH_TRY(up(B.pose[1], h_pose, s));
  OSH_TRY(up(B.vba[0], h_vba, s)); OSH_TRY(up(B.vba[1], h_vba, s));
  OSH_TRY(up(B.pts[0], h_pts, s)); OSH_TRY(up(B.pts[1], h_pts, s));
  OSH_TRY(up(B.e_pose, h_ep, s)); OSH_TRY(up(B.e_point, h_el, s)); OSH_TRY(up(B.e_kind, h_kind, s)); OSH_TRY(up(B.e_obs, h_obs, s));
  OSH_TRY(up(B.e_info, h_info, s)); OSH_TRY(up(B.e_orig, h_eo, s)); OSH_TRY(up(B.lm_off, h_lmo, s)); OSH_TRY(up(B.pel_off, h_po, s));
  OSH_TRY(up(B.pel_edge, h_pel, s)); OSH_TRY(up(B.lmpe, h_lmpe, s)); OSH_TRY(up(B.l_prev, h_lp, s)); OSH_TRY(up(B.l_cur, h_lc, s));
  OSH_TRY(up(B.l_pre, h_pre, s)); OSH_TRY(up(B.l_info, h_li, s)); OSH_TRY(up(B.l_ig, h_lg, s)); OSH_TRY(up(B.l_ia, h_la, s));
  OSH_TRY(up(B.l_rob, h_rob, s));
  auto R = [](DevBuf& b, size_t bytes) { return b.reserve(std::max<size_t>(bytes, 8)); };
  OSH_TRY(R(B.out, nw * sizeof(LibaOut))); OSH_TRY(R(B.Hpl, E * 18 * 8)); OSH_TRY(R(B.BD, E * 18 * 8)); OSH_TRY(R(B.Hll, L * 6 * 8)); OSH_TRY(R(B.bl, L * 3 * 8));
  OSH_TRY(R(B.dinv, L * 9 * 8)); OSH_TRY(R(B.H, Htot * 8)); OSH_TRY(R(B.S, Htot * 8)); OSH_TRY(R(B.b, btot * 8)); OSH_TRY(R(B.bs, btot * 8));
  OSH_TRY(R(B.x, btot * 8)); OSH_TRY(R(B.linkJ, NL * 306 * 8)); OSH_TRY(R(B.o_chi2, E * 8)); OSH_TRY(R(B.o_depth, E));
  OSH_HIP(hipMemsetAsync(B.Hpl.p, 0, std::max<size_t>(E * 18 * 8, 8), s));
  LibaView v{};
  v.desc = B.desc.as<LibaDesc>(); v.out = B.out.as<LibaOut>();
  for (int k = 0; k < 2; ++k) { v.pose[k] = B.pose[k].as<double>(); v.vba[k] = B.vba[k].as<double>(); v.pts[k] = B.pts[k].as<double>(); }
  v.e_pose = B.e_pose.as<int>(); v.e_point = B.e_point.as<int>(); v.e_kind = B.e_kind.as<unsigned char>(); v.e_obs = B.e_obs.as<double>();
  v.e_info = B.e_info.as<double>(); v.e_orig = B.e_orig.as<int>(); v.lm_off = B.lm_off.as<int>(); v.pel_off = B.pel_off.as<int>();
  v.pel_edge = B.pel_edge.as<int>(); v.lm_pose_edge = B.lmpe.as<int>(); v.link_prev = B.l_prev.as<int>(); v.link_cur = B.l_cur.as<int>();
  v.link_preint = B.l_pre.as<float>(); v.link_info = B.l_info.as<double>(); v.link_info_g = B.l_ig.as<double>(); v.link_info_a = B.l_ia.as<double>();
  v.link_robust = B.l_rob.as<unsigned char>(); v.Hpl = B.Hpl.as<double>(); v.BD = B.BD.as<double>(); v.Hll = B.Hll.as<double>(); v.bl = B.bl.as<double>();
  v.dinv = B.dinv.as<double>(); v.H = B.H.as<double>(); v.b = B.b.as<double>(); v.S = B.S.as<double>(); v.bs = B.bs.as<double>();
  v.x = B.x.as<double>(); v.linkJ = B.linkJ.as<double>(); v.out_chi2 = B.o_chi2.as<double>(); v.out_depth = B.o_depth.as<unsigned char>();
  static bool attr_done = false;
  if (!attr_done) { OSH_HIP(hipFuncSetAttribute((const void*)k_liba, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64)); attr_done = true; }
  hipLaunchKernelGGL(k_liba, dim3((unsigned)nw), dim3(kLT), lds, s, v, W);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { set_error("k_liba launch failed: %s", hipGetErrorString(le)); return OSH_ERR_DEVICE; }
  std::vector<LibaOut> h_out(nw);
  OSH_HIP(hipMemcpyAsync(h_out.data(), B.out.p, nw * sizeof(LibaOut), hipMemcpyDeviceToHost, s));
  OSH_HIP(hipStreamSynchronize(s));
  std::vector<double> tmp;
  for (int w = 0; w < nw; ++w) {
    const LibaDesc& d = h_desc[w];
    const LibaOut& o = h_out[w];
    osh_liba_result& r = res[w];
    r.status = OSH_OK; r.iterations = o.iterations; r.trials = o.trials; r.n_trace = o.n_trace;
    r.chi2_initial = o.chi2_initial; r.chi2_final = o.chi2_final;
    for (int k = 0; k < o.n_trace; ++k) { r.chi2_trace[k] = o.chi2_trace[k]; r.lambda_trace[k] = o.lambda_trace[k]; r.trials_trace[k] = o.trials_trace[k]; }
    tmp.resize((size_t)d.N * 24);
    OSH_HIP(hipMemcpy(tmp.data(), B.pose[o.sel].as<double>() + (size_t)d.pose_off * 24, tmp.size() * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < d.N; ++k) {
      const double* q = &tmp[(size_t)k * 24];
      if (r.pose_Rcw) std::memcpy(r.pose_Rcw + 9 * k, q, 72);
      if (r.pose_tcw) std::memcpy(r.pose_tcw + 3 * k, q + 9, 24);
      if (r.pose_Rwb) std::memcpy(r.pose_Rwb + 9 * k, q + 12, 72);
      if (r.pose_twb) std::memcpy(r.pose_twb + 3 * k, q + 21, 24);
    }
    tmp.resize((size_t)d.N * 9);
    OSH_HIP(hipMemcpy(tmp.data(), B.vba[o.sel].as<double>() + (size_t)d.vel_off * 9, tmp.size() * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < d.N; ++k) {
      if (r.vel) std::memcpy(r.vel + 3 * k, &tmp[(size_t)k * 9], 24);
      if (r.bias_g) std::memcpy(r.bias_g + 3 * k, &tmp[(size_t)k * 9 + 3], 24);
      if (r.bias_a) std::memcpy(r.bias_a + 3 * k, &tmp[(size_t)k * 9 + 6], 24);
    }
    if (r.points && d.L) OSH_HIP(hipMemcpy(r.points, B.pts[o.sel].as<double>() + (size_t)d.pt_off * 3, (size_t)d.L * 24, hipMemcpyDeviceToHost));
    if (r.edge_chi2 && d.E) OSH_HIP(hipMemcpy(r.edge_chi2, B.o_chi2.as<double>() + d.edge_off, (size_t)d.E * 8, hipMemcpyDeviceToHost));
    if (r.edge_depth_pos && d.E) OSH_HIP(hipMemcpy(r.edge_depth_pos, B.o_depth.as<unsigned char>() + d.edge_off, (size_t)d.E, hipMemcpyDeviceToHost));
  }
  return OSH_OK;
}
