// pose_device.hip -- pose-only optimisation of tracked frames on MI355X (gfx950): HIP kernel + C-ABI driver.
//
// Replaces, for a batch of frames, the solver part of ORB_SLAM3::Optimizer::PoseOptimization
// (src/Optimizer.cc:815-1114): one VertexSE3Expmap, unary edges EdgeSE3ProjectXYZOnlyPose
// (src/OptimizableTypes.cpp:49-61) / g2o::EdgeStereoSE3ProjectXYZOnlyPose
// (Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:306-405), four rounds of
// optimizer.optimize(10) that restart from the frame's pose, re-classify every edge (chi2 as float against
// 5.991 / 7.815, :1035-1105) and drop the Huber kernel after the third round (:1054).
//
// The problem is tiny (6 unknowns, O(10^3) edges) and latency-critical, so the whole of it -- all four rounds with
// their Levenberg-Marquardt loops (g2o/core/sparse_optimizer.cpp:354-419, optimization_algorithm_levenberg.cpp:61-169)
// and the dense 6x6 solve (g2o/solvers/linear_solver_dense.h:97-105) -- runs inside ONE block per frame: no host round
// trip, thread per edge, fixed-order block reductions (bitwise reproducible).
#include "common.h"
#include "lba_math.h"
#include <cfloat>
#include <cstring>
#include <vector>

namespace osh {

constexpr int kPT = 256;   // threads per frame block

struct PoseDesc {
  int E, edge_off;
  double qt[7], cam[5], huber_mono, huber_stereo;
  double kb8[4];   // KannalaBrandt8 k1..k4 (osh_pose_problem.kb8)
  int kb8_on;      // 1: the frame's mono edges project through KannalaBrandt8
  double cam2[8], trl[7];   // fisheye stereo frame: right camera and Trl of the OSH_EDGE_BODY edges (EdgeSE3ProjectXYZOnlyPoseToBody)
  float chi2_mono[4], chi2_stereo[4];
  int iters[4];
};
struct PoseOut { double qt[7]; double chi2_final[4]; int iterations[4]; int n_bad, rounds; };
struct PoseView {
  const PoseDesc* desc;
  PoseOut* out;
  const double* X;            // [NE*3]
  const unsigned char* kind;  // [NE]
  const double* obs;          // [NE*3]
  const double* info;         // [NE]
  double* chi2;               // [NE] chi2 of the edge's _error as last computed
  unsigned char* level;       // [NE] 1: classified outlier, outside the active set
};

// deterministic block sum: butterfly inside each wavefront, wavefronts added in order
__device__ __forceinline__ double pose_block_sum(double v, double* sh) {
  v = dev::wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int k = 0; k < kPT / 64; ++k) t += sh[k];
  return t;
}

// edge residual / Jacobians through the frame's camera model; the pinhole instantiation carries no KannalaBrandt8 code
template <bool KB8>
__device__ __forceinline__ double pose_edge_residual(const PoseDesc& d, int kind, const double* qt, const double* X, const double* obs,
                                                     double info, double* r, double* Xc) {
  if (KB8 && kind == OSH_EDGE_BODY) { double Xe[3]; return dev::edge_residual_body(qt, d.cam2, d.trl, X, obs, info, r, Xc, Xe); }   // Xc: left-camera point
  if (KB8 && d.kb8_on && kind == OSH_EDGE_MONO) return dev::edge_residual_kb8(qt, d.cam, d.kb8, X, obs, info, r, Xc);
  return dev::edge_residual(kind, qt, d.cam, X, obs, info, r, Xc);
}
template <bool KB8>
__device__ __forceinline__ void pose_edge_jacobians(const PoseDesc& d, int kind, const double* R, const double* Xc, double* JX, double* Jp) {
  if (KB8 && kind == OSH_EDGE_BODY) { dev::edge_jacobian_pose_body(d.cam2, d.trl, Xc, Jp); return; }   // unary edge: JX unused
  if (KB8 && d.kb8_on && kind == OSH_EDGE_MONO) { dev::edge_jacobians_kb8(R, d.cam, d.kb8, Xc, JX, Jp); return; }
  dev::edge_jacobians(kind, R, d.cam, Xc, JX, Jp);
}

// The frame's edges never change during the optimisation: every thread keeps the inputs of its first kPC edges (edge tid + q kPT),
// their level and the chi2 last computed in registers -- 40 Levenberg-Marquardt iterations otherwise pay two memory round trips
// per pass over the edges.  Edges beyond kPC kPT of a frame (more than 1024 matches) go through global memory as before.
constexpr int kPC = 4;
struct EdgeCache {
  double X[kPC][3], obs[kPC][3], info[kPC], chi2[kPC];
  int kind[kPC];
  unsigned char level[kPC];
};
// f(kind, X, obs, info, level&, chi2&) on every edge of the frame owned by this thread
template <class F>
__device__ __forceinline__ void for_edges(const PoseView& v, const PoseDesc& d, EdgeCache& c, F&& f) {
#pragma unroll
  for (int q = 0; q < kPC; ++q) {
    const int e = threadIdx.x + q * kPT;
    if (e < d.E) f(c.kind[q], c.X[q], c.obs[q], c.info[q], c.level[q], c.chi2[q]);
  }
  for (int e = threadIdx.x + kPC * kPT; e < d.E; e += kPT) {
    const size_t ge = (size_t)d.edge_off + e;
    double X[3], obs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { X[k] = v.X[ge * 3 + k]; obs[k] = v.obs[ge * 3 + k]; }
    unsigned char lv = v.level[ge];
    double ch = v.chi2[ge];
    f((int)v.kind[ge], X, obs, v.info[ge], lv, ch);
    v.level[ge] = lv; v.chi2[ge] = ch;
  }
}

// computeActiveErrors + activeRobustChi2 at pose `qt` over the active edges; stores every edge's chi2
template <bool KB8>
__device__ double pose_eval(const PoseView& v, const PoseDesc& d, EdgeCache& ec, const double* qt, bool robust, double* sh) {
  double acc = 0.0;
  for_edges(v, d, ec, [&](int kind, const double* X, const double* obs, double info, unsigned char& level, double& chi2) {
    if (level) return;
    double r[3], Xc[3];
    const double c = pose_edge_residual<KB8>(d, kind, qt, X, obs, info, r, Xc);
    chi2 = c;
    if (robust) {
      double r0, r1;
      dev::huber(c, kind != OSH_EDGE_STEREO ? d.huber_mono : d.huber_stereo, r0, r1);
      acc += r0;
    } else acc += c;
  });
  return pose_block_sum(acc, sh);
}

template <bool KB8>
__global__ __launch_bounds__(kPT) void k_pose_opt(PoseView v) {
  __shared__ double sh[kPT / 64];
  __shared__ double shH[28];          // upper(Hpp) (21), b (6), spare
  __shared__ double shn[(kPT / 64) * 27];   // per-wavefront partials of the 27 sums
  __shared__ double sh_qt[2][7];      // current / trial estimate
  __shared__ double sh_x[6];
  __shared__ int sh_ok;
  const PoseDesc& d = v.desc[blockIdx.x];
  PoseOut& out = v.out[blockIdx.x];
  const int tid = threadIdx.x;
  for (int e = tid + kPC * kPT; e < d.E; e += kPT) { v.level[(size_t)d.edge_off + e] = 0; v.chi2[(size_t)d.edge_off + e] = 0.0; }
  EdgeCache ec;
#pragma unroll
  for (int q = 0; q < kPC; ++q) {
    const size_t ge = (size_t)d.edge_off + min(tid + q * kPT, max(d.E - 1, 0));
#pragma unroll
    for (int k = 0; k < 3; ++k) { ec.X[q][k] = v.X[ge * 3 + k]; ec.obs[q][k] = v.obs[ge * 3 + k]; }
    ec.info[q] = v.info[ge]; ec.kind[q] = v.kind[ge]; ec.level[q] = 0; ec.chi2[q] = 0.0;
  }
  bool robust = true;
  int n_bad = 0, rounds = 0;
  __syncthreads();
  for (int round = 0; round < 4; ++round) {
    // vSE3->setEstimate(g2o::SE3Quat(q, t)): the rotation is normalised (se3quat.h:61-63)
    if (tid == 0) {
      double q[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) q[k] = d.qt[k];
      dev::quat_normalize_rotation(q);
#pragma unroll
      for (int k = 0; k < 7; ++k) sh_qt[0][k] = q[k];
    }
    __syncthreads();
    int sel = 0, cj = 0;
    bool ok = true;
    double lambda = -1.0, ni = 2.0, last_chi = 0.0;
    int nBad = 0;
    // initializeOptimization(0): nothing to do without active edges
    int active = 0;
    for_edges(v, d, ec, [&](int, const double*, const double*, double, unsigned char& level, double&) { active += level ? 0 : 1; });
    const bool any = pose_block_sum((double)active, sh) > 0.0;
    for (int it = 0; it < d.iters[round] && ok && any; ++it) {
      double qt[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) qt[k] = sh_qt[sel][k];
      // activeRobustChi2 of the current estimate: after the first iteration of a round it is the accepted trial's value (the same
      // sum over the same edges of the same buffer; an iteration that accepts nothing ends the round)
      double currentChi = it == 0 ? pose_eval<KB8>(v, d, ec, qt, robust, sh) : last_chi;
      const double iniChi = currentChi;
      // ---- buildSystem: Hpp += Jp^T W Jp, b += Jp^T (-rho' Omega r) over the active edges
      double H[21], b[6];
#pragma unroll
      for (int k = 0; k < 21; ++k) H[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) b[k] = 0.0;
      double R[9];
      dev::quat_to_R(qt, R);
      for_edges(v, d, ec, [&](int kind, const double* X, const double* obs, double info, unsigned char& level, double&) {
        if (level) return;
        double r[3], Xc[3];
        const double c = pose_edge_residual<KB8>(d, kind, qt, X, obs, info, r, Xc);
        double r0 = c, r1 = 1.0;
        if (robust) dev::huber(c, kind != OSH_EDGE_STEREO ? d.huber_mono : d.huber_stereo, r0, r1);
        double JX[9], Jp[18];
        pose_edge_jacobians<KB8>(d, kind, R, Xc, JX, Jp);
        const double ww = r1 * info;
        const double wr[3] = {-(info * r[0]) * r1, -(info * r[1]) * r1, -(info * r[2]) * r1};
        int m = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double b0 = Jp[a] * ww, b1 = Jp[6 + a] * ww, b2 = Jp[12 + a] * ww;
#pragma unroll
          for (int c2 = a; c2 < 6; ++c2) { H[m] += b0 * Jp[c2] + b1 * Jp[6 + c2] + b2 * Jp[12 + c2]; ++m; }
          b[a] += Jp[a] * wr[0] + Jp[6 + a] * wr[1] + Jp[12 + a] * wr[2];
        }
      });
      {
        // 27 block sums with two barriers: wavefront butterflies, partials parked in LDS, added in wavefront order (deterministic);
        // one barrier pair per value cost 54 barriers per LM iteration
        double red[27];
#pragma unroll
        for (int k = 0; k < 21; ++k) red[k] = dev::wave_sum(H[k]);
#pragma unroll
        for (int k = 0; k < 6; ++k) red[21 + k] = dev::wave_sum(b[k]);
        __syncthreads();
        if ((tid & 63) == 0) {
#pragma unroll
          for (int k = 0; k < 27; ++k) shn[(tid >> 6) * 27 + k] = red[k];
        }
        __syncthreads();
        if (tid < 27) {
          double t = 0.0;
#pragma unroll
          for (int w = 0; w < kPT / 64; ++w) t += shn[w * 27 + tid];
          shH[tid] = t;
        }
      }
      __syncthreads();
      if (it == 0) {
        // computeLambdaInit: tau * max |H_dd| (optimization_algorithm_levenberg.cpp:171-185); diagonal = entries 0, 6, 11, 15, 18, 20
        lambda = 1e-5 * fmax(fmax(fmax(fabs(shH[0]), fabs(shH[6])), fmax(fabs(shH[11]), fabs(shH[15]))), fmax(fabs(shH[18]), fabs(shH[20])));
        ni = 2.0; nBad = 0;
      }
      double rho = 0.0;
      int qmax = 0;
      do {
        const int trs = sel ^ 1;
        if (tid == 0) {
          // (Hpp + lambda I) x = b by LDL^T; the solver reports failure unless every pivot is positive
          // (every loop unrolled: with run-time indices the 6x6 array lives in scratch memory and this one-thread section was
          // most of an iteration; a non-positive pivot no longer leaves the loop early, its results are simply not used)
          double A[36], x[6];
          {
            int m = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
              for (int c2 = a; c2 < 6; ++c2) { A[a * 6 + c2] = shH[m] + ((a == c2) ? lambda : 0.0); ++m; }
          }
          bool good = true;
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            const double dk = A[k * 6 + k];
            good = good && (dk > 0.0);
            double l[6];
#pragma unroll
            for (int i = k + 1; i < 6; ++i) l[i] = A[k * 6 + i] / dk;
#pragma unroll
            for (int i = k + 1; i < 6; ++i)
#pragma unroll
              for (int j = i; j < 6; ++j) A[i * 6 + j] -= l[i] * A[k * 6 + j];
#pragma unroll
            for (int i = k + 1; i < 6; ++i) A[k * 6 + i] = l[i];
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) x[k] = shH[21 + k];
#pragma unroll
          for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int i = k + 1; i < 6; ++i) x[i] -= A[k * 6 + i] * x[k];
#pragma unroll
          for (int k = 0; k < 6; ++k) x[k] /= A[k * 6 + k];
#pragma unroll
          for (int k = 5; k >= 0; --k) {
            double s2 = x[k];
#pragma unroll
            for (int i = k + 1; i < 6; ++i) s2 -= A[k * 6 + i] * x[i];
            x[k] = s2;
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) x[k] = good ? x[k] : 0.0;
          double qin[7], qout[7];
          for (int k = 0; k < 7; ++k) qin[k] = sh_qt[sel][k];
          dev::pose_oplus(x, qin, qout);
          for (int k = 0; k < 7; ++k) sh_qt[trs][k] = qout[k];
          for (int k = 0; k < 6; ++k) sh_x[k] = x[k];
          sh_ok = good ? 1 : 0;
        }
        __syncthreads();
        double qtr[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) qtr[k] = sh_qt[trs][k];
        double tempChi = pose_eval<KB8>(v, d, ec, qtr, robust, sh);
        if (!sh_ok) tempChi = DBL_MAX;
        rho = currentChi - tempChi;
        double scale = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) scale += sh_x[k] * (lambda * sh_x[k] + shH[21 + k]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && isfinite(tempChi)) {
          double alpha = 1. - pow((2 * rho - 1), 3);
          alpha = fmin(alpha, 2. / 3.);
          lambda *= fmax(1. / 3., alpha);
          ni = 2; currentChi = tempChi;
          sel = trs;
        } else {
          lambda *= ni; ni *= 2;
        }
        qmax++;
        __syncthreads();
      } while (rho < 0 && qmax < 10);
      ++cj;
      last_chi = currentChi;
      if (qmax == 10 || rho == 0) { ok = false; continue; }
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      if (nBad >= 3) { ok = false; continue; }
    }
    // ---- classification (:1030-1105).  chi2 of an inlier = its _error as last computed by the optimiser (after a rejected
    // final trial that is the trial's error, as in the reference); an outlier is re-evaluated at the final estimate.
    double qf[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) qf[k] = sh_qt[sel][k];
    int bad = 0;
    for_edges(v, d, ec, [&](int kind, const double* X, const double* obs, double info, unsigned char& level, double& chi2e) {
      if (level) {
        double r[3], Xc[3];
        chi2e = pose_edge_residual<KB8>(d, kind, qf, X, obs, info, r, Xc);
      }
      const float chi2 = (float)chi2e;
      const float th = kind != OSH_EDGE_STEREO ? d.chi2_mono[round] : d.chi2_stereo[round];
      if (chi2 > th) { level = 1; ++bad; } else level = 0;
    });
    n_bad = (int)pose_block_sum((double)bad, sh);
    rounds = round + 1;
    if (tid == 0) { out.iterations[round] = cj; out.chi2_final[round] = last_chi; }
    if (round == 2) robust = false;
    if (tid == 0 && (round == 3 || d.E < 10)) {
#pragma unroll
      for (int k = 0; k < 7; ++k) out.qt[k] = sh_qt[sel][k];
    }
    __syncthreads();
    if (d.E < 10) break;
  }
  if (tid == 0) { out.n_bad = n_bad; out.rounds = rounds; }
  // the cached edges' outlier flags and chi2 for the caller
#pragma unroll
  for (int q = 0; q < kPC; ++q) {
    const int e = tid + q * kPT;
    if (e < d.E) { v.level[(size_t)d.edge_off + e] = ec.level[q]; v.chi2[(size_t)d.edge_off + e] = ec.chi2[q]; }
  }
}

struct PosePinned {
  void* p = nullptr;
  size_t cap = 0;
  ~PosePinned() { if (p) (void)hipHostFree(p); }
  void* reserve(size_t bytes) {
    if (bytes <= cap) return p;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&p, want) != hipSuccess) { p = nullptr; return nullptr; }
    cap = want;
    return p;
  }
};
// staging and device arena of osh_pose_optimize: kept with the context (released by osh_lba_destroy)
struct PoseBuffers { PosePinned h_in, h_out; DevBuf arena; };

}  // namespace osh

using namespace osh;

#define OSH_TRY(expr) do { int _rc = (expr); if (_rc != OSH_OK) return _rc; } while (0)

extern "C" int osh_lba_stream(osh_lba_ctx* ctx, int* device, hipStream_t* stream);   // lba_device.hip
extern "C" void** osh_lba_attachment(osh_lba_ctx* ctx, int slot, void (*free_fn)(void*));   // lba_device.hip

extern "C" int osh_pose_optimize(osh_lba_ctx* ctx, int32_t n, const osh_pose_problem* pr, osh_pose_result* res) {
  if (!ctx || n <= 0 || !pr || !res) { set_error("osh_pose_optimize: bad arguments"); return OSH_ERR_INVALID; }
  int device = 0;
  hipStream_t s = nullptr;
  OSH_TRY(osh_lba_stream(ctx, &device, &s));
  OSH_HIP(hipSetDevice(device));
  std::vector<PoseDesc> h_desc(n);
  size_t NE = 0;
  bool any_kb8 = false;
  for (int f = 0; f < n; ++f) {
    const osh_pose_problem& p = pr[f];
    if (p.n_edges < 0 || !p.pose_qt || !p.cam || (p.n_edges > 0 && (!p.points || !p.edge_kind || !p.edge_obs || !p.edge_info))) {
      set_error("frame %d: negative size or NULL array", f); return OSH_ERR_INVALID;
    }
    PoseDesc& d = h_desc[f];
    d.E = p.n_edges; d.edge_off = (int)NE;
    for (int k = 0; k < 7; ++k) d.qt[k] = p.pose_qt[k];
    for (int k = 0; k < 5; ++k) d.cam[k] = p.cam[k];
    d.huber_mono = p.huber_mono; d.huber_stereo = p.huber_stereo;
    d.kb8_on = p.kb8 ? 1 : 0;
    for (int k = 0; k < 4; ++k) d.kb8[k] = p.kb8 ? p.kb8[k] : 0.0;
    const bool rig = p.kb8 && p.cam2 && p.trl;
    for (int k = 0; k < 8; ++k) d.cam2[k] = rig ? p.cam2[k] : 0.0;
    for (int k = 0; k < 7; ++k) d.trl[k] = rig ? p.trl[k] : (k == 3 ? 1.0 : 0.0);
    if (p.kb8) {
      any_kb8 = true;
      for (int e = 0; e < p.n_edges; ++e)
        if (p.edge_kind[e] == OSH_EDGE_STEREO) { set_error("frame %d: a KannalaBrandt8 frame takes monocular and right-camera edges only (edge %d)", f, e); return OSH_ERR_UNSUPPORTED; }
    }
    for (int e = 0; e < p.n_edges; ++e)
      if (p.edge_kind[e] == OSH_EDGE_BODY && !rig) { set_error("frame %d edge %d: a right-camera edge needs kb8, cam2 and trl", f, e); return OSH_ERR_INVALID; }
    for (int k = 0; k < 4; ++k) { d.chi2_mono[k] = p.chi2_mono[k]; d.chi2_stereo[k] = p.chi2_stereo[k]; d.iters[k] = p.iterations[k]; }
    for (int e = 0; e < p.n_edges; ++e) if (p.edge_kind[e] > OSH_EDGE_BODY) { set_error("frame %d edge %d: kind out of range", f, e); return OSH_ERR_INVALID; }
    NE += (size_t)p.n_edges;
  }
  if (NE > 0x7fffff00u) { set_error("batch too large for 32-bit offsets"); return OSH_ERR_UNSUPPORTED; }
  // one pinned staging buffer, one device arena, one copy each way (eight separate copies cost a third of a single frame's call)
  void** slot = osh_lba_attachment(ctx, 1, [](void* q) { delete static_cast<PoseBuffers*>(q); });
  if (!slot) { set_error("osh_pose_optimize: no context"); return OSH_ERR_INVALID; }
  if (!*slot) *slot = new PoseBuffers();
  PoseBuffers& B = *static_cast<PoseBuffers*>(*slot);
  size_t in_bytes = 0, out_bytes = 0;
  auto take = [](size_t& total, size_t bytes) { const size_t o = total; total = (total + std::max<size_t>(bytes, 8) + 255) & ~(size_t)255; return o; };
  const size_t i_desc = take(in_bytes, n * sizeof(PoseDesc)), i_X = take(in_bytes, NE * 24), i_obs = take(in_bytes, NE * 24), i_info = take(in_bytes, NE * 8),
               i_kind = take(in_bytes, NE);
  const size_t o_out = take(out_bytes, n * sizeof(PoseOut)), o_level = take(out_bytes, NE), o_chi2 = take(out_bytes, NE * 8);
  char* hs = static_cast<char*>(B.h_in.reserve(in_bytes));
  char* hr = static_cast<char*>(B.h_out.reserve(out_bytes));
  if (!hs || !hr) { set_error("osh_pose_optimize: pinned staging allocation failed"); return OSH_ERR_DEVICE; }
  std::memcpy(hs + i_desc, h_desc.data(), n * sizeof(PoseDesc));
  double* h_X = reinterpret_cast<double*>(hs + i_X); double* h_obs = reinterpret_cast<double*>(hs + i_obs); double* h_info = reinterpret_cast<double*>(hs + i_info);
  unsigned char* h_kind = reinterpret_cast<unsigned char*>(hs + i_kind);
  for (int f = 0; f < n; ++f) {
    const osh_pose_problem& p = pr[f];
    const size_t o = (size_t)h_desc[f].edge_off;
    if (p.n_edges > 0) {
      std::memcpy(h_X + o * 3, p.points, (size_t)p.n_edges * 24);
      std::memcpy(h_obs + o * 3, p.edge_obs, (size_t)p.n_edges * 24);
      std::memcpy(h_info + o, p.edge_info, (size_t)p.n_edges * 8);
      std::memcpy(h_kind + o, p.edge_kind, (size_t)p.n_edges);
    }
  }
  OSH_TRY(B.arena.reserve(in_bytes + out_bytes));
  char* din = B.arena.as<char>();
  char* dout = din + in_bytes;
  OSH_HIP(hipMemcpyAsync(din, hs, in_bytes, hipMemcpyHostToDevice, s));
  PoseView v;
  v.desc = reinterpret_cast<const PoseDesc*>(din + i_desc); v.out = reinterpret_cast<PoseOut*>(dout + o_out);
  v.X = reinterpret_cast<const double*>(din + i_X); v.kind = reinterpret_cast<const unsigned char*>(din + i_kind);
  v.obs = reinterpret_cast<const double*>(din + i_obs); v.info = reinterpret_cast<const double*>(din + i_info);
  v.chi2 = reinterpret_cast<double*>(dout + o_chi2); v.level = reinterpret_cast<unsigned char*>(dout + o_level);
  if (any_kb8) hipLaunchKernelGGL(k_pose_opt<true>, dim3((unsigned)n), dim3(kPT), 0, s, v);
  else hipLaunchKernelGGL(k_pose_opt<false>, dim3((unsigned)n), dim3(kPT), 0, s, v);
  { hipError_t e = hipGetLastError(); if (e != hipSuccess) { set_error("kernel launch k_pose_opt failed: %s", hipGetErrorString(e)); return OSH_ERR_DEVICE; } }
  OSH_HIP(hipMemcpyAsync(hr, dout, out_bytes, hipMemcpyDeviceToHost, s));
  OSH_HIP(hipStreamSynchronize(s));
  const PoseOut* h_out = reinterpret_cast<const PoseOut*>(hr + o_out);
  const unsigned char* h_level = reinterpret_cast<const unsigned char*>(hr + o_level);
  const double* h_chi2 = reinterpret_cast<const double*>(hr + o_chi2);
  for (int f = 0; f < n; ++f) {
    osh_pose_result& r = res[f];
    const PoseOut& o = h_out[f];
    for (int k = 0; k < 7; ++k) r.pose_qt[k] = o.qt[k];
    for (int k = 0; k < 4; ++k) { r.iterations[k] = k < o.rounds ? o.iterations[k] : 0; r.chi2_final[k] = k < o.rounds ? o.chi2_final[k] : 0.0; }
    r.n_bad = o.n_bad; r.rounds = o.rounds; r.status = OSH_OK;
    const size_t off = (size_t)h_desc[f].edge_off;
    if (r.outlier) for (int e = 0; e < h_desc[f].E; ++e) r.outlier[e] = h_level[off + e];
    if (r.edge_chi2) for (int e = 0; e < h_desc[f].E; ++e) r.edge_chi2[e] = h_chi2[off + e];
  }
  return OSH_OK;
}
