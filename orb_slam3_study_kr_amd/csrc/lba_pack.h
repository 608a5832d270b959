// lba_pack.h -- host-side packing of a batch of local-BA windows into the flat device layout (pure C++, no device code).
//
// This is the device path's counterpart of SparseOptimizer::initializeOptimization + BlockSolver::buildStructure
// (Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:199-267, block_solver.hpp:143-295): index mapping and the structure of
// the Schur complement, built once per upload.  What it produces, per window:
//   * edges sorted landmark-major, inside a landmark by pose index (optimisable poses first);
//   * the two edges a fisheye-stereo rig puts on one (keyframe, landmark) Hessian block (EdgeSE3ProjectXYZ +
//     EdgeSE3ProjectXYZToBody, src/Optimizer.cc:1305-1399) merged into ONE sorted edge with two observation records;
//   * the Schur work plan (schur_plan.h) and, following it, a RENUMBERING of the landmarks: landmark indices follow the
//     order of the plan's owner records, so the landmarks (and therefore the edges) an item walks are contiguous in memory;
//   * one 32-byte observation record per sorted edge: u, v, u_right, +-invSigma2 (sign bit set = monocular edge), so the
//     pinhole kernels need no per-edge kind byte.
// Everything lands in two host arenas (fixed-size sections, plan sections) that the caller supplies through an allocator
// (pinned memory in the device path) and copies to the device with one transfer each.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/orbslam3_hip.h"
#include "schur_plan.h"

namespace osh {

constexpr int kBlock = 256;            // threads per block of the edge/landmark kernels
constexpr int kChunkEdges = 256;       // edges handled per pass of a chunk (== kBlock)
constexpr int kChunkMaxEdges = 1024;   // edges of one chunk at most (= one block of the landmark-major kernels)
// Edges per chunk for a batch of nw windows: a call with one or two windows spreads its landmark-major kernels over four times as
// many blocks (a config-2 window: 74 blocks of 1024 edges take 11-12 us per launch, 300 of 256 edges ...), as item_max_lm() does for the
// Schur items.  A multiple of kChunkEdges.
inline int chunk_max_edges(int nw) {
  if (const char* e = std::getenv("OSH_LBA_CHUNK_EDGES")) { const int v = std::atoi(e); if (v >= kChunkEdges && v <= kChunkMaxEdges && v % kChunkEdges == 0) return v; }   // (tuning aid)
  return nw <= 2 ? 256 : (nw <= 8 ? 512 : kChunkMaxEdges);
}

// internal edge kinds of the sorted edge list (the C-ABI kinds OSH_EDGE_* plus the merged rig edge)
constexpr int kKindMono = 0, kKindStereo = 1, kKindBody = 2, kKindBoth = 3;

struct WinDesc {
  int P, F, L, E;      // E = sorted edges (a merged left+right pair counts once)
  int pose_off;        // first pose of the window in the pose arrays (P+F poses per window)
  int fpose_off;       // first optimisable pose in the per-free-pose arrays
  int pt_off;          // first landmark
  int edge_off;        // first (sorted) edge
  int lmoff_off;       // start of this window's L+1 landmark->edge offsets
  int chunk_off, n_chunks;
  int sitem_off, n_sitems;  // this window's symmetric items of the Schur plan
  int n;               // 6P
  int max_iter;
  int kb8_on;          // 1: the window's mono edges project through KannalaBrandt8
  int rig_on;          // 1: fisheye stereo rig: body edges through trl / cam2
  int in_edges;        // edges of the caller's problem (outputs are per caller edge)
  int out_off;         // first caller edge of the window in the per-caller-edge output arrays
  long long S_off;     // doubles
  double huber_mono, huber_stereo, lambda_init;
  double kb8[4];       // KannalaBrandt8 k1..k4 of the window's (left) camera
  double cam2[8];      // right camera of a fisheye rig: fx fy cx cy k1..k4
  double trl[7];       // Trl (left camera -> right camera) as qx qy qz qw tx ty tz
};

struct Chunk { int win, lm0, lm1; };
struct I2 { int x, y; };

struct PackedBatch {
  int nw = 0;
  std::vector<WinDesc> win;
  size_t NP = 0, NFP = 0, NL = 0, NE = 0, NLO = 0, NOUT = 0, S_total = 0;
  size_t n_chunks = 0, n_items = 0, n_sym = 0, n_recs = 0, n_rblk = 0, n_contrib = 0, n_ccontrib = 0;
  long long tile_steps = 0, pair_blocks = 0;
  int n_max = 0, np_max = 0;
  bool has_kb8 = false, has_rig = false;
  bool rec_f32 = false;   // every observation / information value of the batch is a float32 value (what the reference stores:
                          // cv::KeyPoint::pt, mvuRight, mvInvLevelSigma2): the records travel as 16 bytes and are widened on the device
  // arena 0: sections whose sizes follow from the problem sizes; arena 1: plan sections
  enum Sec { POSE, CAM, PT, EREC, EREC2, EPOSE, EPOINT, EORIG, EORIG2, LMOFF, LMPERM, FPW, EKIND, A0_COUNT,
             WIN = A0_COUNT, CHUNKS, ITEMS, RECS, SPAIR, SCSLOT, POSEX, POSEY, RBLK, CRANGE, SEC_COUNT };
  size_t off[SEC_COUNT] = {0}, bytes[SEC_COUNT] = {0};
  unsigned char* arena[2] = {nullptr, nullptr};
  size_t arena_bytes[2] = {0, 0};
  template <class T> T* sec(int s) const { return reinterpret_cast<T*>(arena[s < A0_COUNT ? 0 : 1] + off[s]); }
  size_t sec_off(int s) const { return off[s]; }
  int sec_arena(int s) const { return s < A0_COUNT ? 0 : 1; }
  int err = OSH_OK;
  char msg[400] = {0};
};

namespace pack_detail {

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct WinLocal {
  std::vector<Chunk> chunks;
  std::vector<plan_detail::Build> builds;
  std::vector<SRec> recs;            // records of all builds, Build::rec_off / n_rec
  SchurPlan plan;
  int n_sym = 0, n_cross = 0;
  size_t recs_sym = 0, recs_cross = 0;
  int err = OSH_OK;
  char msg[320];
};

struct Scratch { std::vector<int> cnt, fill, order, second, tmp_epose, nfree, old2new, old_lmo; plan_detail::PlanScratch plan; };

}  // namespace pack_detail

// Packs `nw` problems.  `alloc(which, bytes)` returns host memory for arena `which` (0, 1) that stays valid until the next
// call with the same `which`.  Returns pb.err (OSH_OK on success), message in pb.msg.
// Pass 1 of every packer (host and device): validates the sizes and pointers of the problems and fills the part of the window
// descriptors that follows from them (offsets of the fixed-size sections, camera models, controller parameters).
inline int pack_describe(int nw, const osh_lba_problem* pr, PackedBatch& pb) {
  pb = PackedBatch();
  pb.nw = nw;
  pb.win.assign(nw, WinDesc{});
  auto fail = [&](int code, const char* fmt, auto... a) {
    pb.err = code;
    if constexpr (sizeof...(a) == 0) std::snprintf(pb.msg, sizeof(pb.msg), "%s", fmt); else std::snprintf(pb.msg, sizeof(pb.msg), fmt, a...);
    return code;
  };
  size_t NP = 0, NFP = 0, NL = 0, NE = 0, NLO = 0, NOUT = 0, S_total = 0;
  for (int w = 0; w < nw; ++w) {
    const osh_lba_problem& p = pr[w];
    if (p.n_free < 0 || p.n_fixed < 0 || p.n_points < 0 || p.n_edges < 0 ||
        (p.n_edges > 0 && (!p.edge_pose || !p.edge_point || !p.edge_kind || !p.edge_obs || !p.edge_info)) ||
        ((p.n_free + p.n_fixed) > 0 && (!p.pose_qt || !p.pose_cam)) || (p.n_points > 0 && !p.points))
      return fail(OSH_ERR_INVALID, "window %d: negative size or NULL array", w);
    if (p.max_iterations > OSH_LBA_MAX_TRACE) return fail(OSH_ERR_INVALID, "window %d: max_iterations > %d", w, OSH_LBA_MAX_TRACE);
    WinDesc& d = pb.win[w];
    d.P = p.n_free; d.F = p.n_fixed; d.L = p.n_points; d.E = 0; d.in_edges = p.n_edges;
    d.pose_off = (int)NP; d.fpose_off = (int)NFP; d.pt_off = (int)NL; d.edge_off = (int)NE; d.lmoff_off = (int)NLO; d.out_off = (int)NOUT;
    d.n = 6 * p.n_free; d.max_iter = p.max_iterations; d.S_off = (long long)S_total;
    d.huber_mono = p.huber_mono; d.huber_stereo = p.huber_stereo; d.lambda_init = p.lambda_init;
    d.kb8_on = p.kb8 ? 1 : 0;
    for (int k = 0; k < 4; ++k) d.kb8[k] = p.kb8 ? p.kb8[k] : 0.0;
    d.rig_on = (p.kb8 && p.cam2 && p.trl) ? 1 : 0;
    for (int k = 0; k < 8; ++k) d.cam2[k] = d.rig_on ? p.cam2[k] : 0.0;
    for (int k = 0; k < 7; ++k) d.trl[k] = d.rig_on ? p.trl[k] : (k == 3 ? 1.0 : 0.0);
    if (p.kb8) pb.has_kb8 = true;
    if (d.rig_on) pb.has_rig = true;
    NP += (size_t)p.n_free + p.n_fixed; NFP += p.n_free; NL += p.n_points; NE += p.n_edges; NOUT += p.n_edges;
    NLO += (size_t)p.n_points + 1;
    S_total += (size_t)d.n * d.n;
    pb.n_max = std::max(pb.n_max, d.n);
    pb.np_max = std::max(pb.np_max, p.n_free + p.n_fixed);
  }
  if (NP > 0x7fffff00u || NL > 0x7fffff00u || NE > 0x7fffff00u) return fail(OSH_ERR_UNSUPPORTED, "batch too large for 32-bit offsets");
  pb.NP = NP; pb.NFP = NFP; pb.NL = NL; pb.NLO = NLO; pb.NOUT = NOUT; pb.S_total = S_total;
  // NE here is an upper bound (merged rig edges shrink it); the sorted edges of a window start at the caller's edge offset,
  // the tail of a window with merged edges stays unused.
  pb.NE = NE;
  return OSH_OK;
}

// Offsets of the sections of arena 0 (sizes follow from the problem sizes and pb.rec_f32 / pb.has_rig); returns the arena size.
inline size_t pack_layout0(PackedBatch& pb) {
  size_t o = 0;
  auto put = [&](int s, size_t b) { pb.off[s] = o; pb.bytes[s] = b; o += pack_detail::align_up(std::max<size_t>(b, 8)); };
  const size_t NP = pb.NP, NL = pb.NL, NE = pb.NE;
  put(PackedBatch::POSE, NP * 7 * 8); put(PackedBatch::CAM, NP * 5 * 8); put(PackedBatch::PT, NL * 3 * 8);
  put(PackedBatch::EREC, NE * (pb.rec_f32 ? 16 : 32)); put(PackedBatch::EREC2, pb.has_rig ? NE * 32 : 0);
  put(PackedBatch::EPOSE, NE * 4); put(PackedBatch::EPOINT, NE * 4); put(PackedBatch::EORIG, NE * 4);
  put(PackedBatch::EORIG2, pb.has_rig ? NE * 4 : 0);
  put(PackedBatch::LMOFF, pb.NLO * 4); put(PackedBatch::LMPERM, NL * 4); put(PackedBatch::FPW, pb.NFP * 4); put(PackedBatch::EKIND, NE);
  pb.arena_bytes[0] = o;
  return o;
}

// Offsets of the plan sections (arena 1) from the plan totals in pb; returns the arena size.
inline size_t pack_layout1(PackedBatch& pb) {
  size_t o = 0;
  auto put = [&](int s, size_t b) { pb.off[s] = o; pb.bytes[s] = b; o += pack_detail::align_up(std::max<size_t>(b, 8)); };
  put(PackedBatch::WIN, (size_t)pb.nw * sizeof(WinDesc)); put(PackedBatch::CHUNKS, pb.n_chunks * sizeof(Chunk));
  put(PackedBatch::ITEMS, pb.n_items * sizeof(SItem)); put(PackedBatch::RECS, pb.n_recs * sizeof(SRec));
  put(PackedBatch::SPAIR, pb.n_items * 64 * 4); put(PackedBatch::SCSLOT, pb.n_items * 8 * 4);
  put(PackedBatch::POSEX, pb.n_items * 8 * 4); put(PackedBatch::POSEY, pb.n_items * 8 * 4);
  put(PackedBatch::RBLK, pb.n_rblk * sizeof(RBlk)); put(PackedBatch::CRANGE, pb.NFP * sizeof(I2));
  pb.arena_bytes[1] = o;
  return o;
}

// g2o::SE3Quat(q,t) constructor: normalizeRotation (se3quat.h:61-63,280-285) -- the pose as every packer stores it
inline void pack_pose(const double* in7, double* out7) {
  double q[4] = {in7[0], in7[1], in7[2], in7[3]};
  if (q[3] < 0) { q[0] *= -1; q[1] *= -1; q[2] *= -1; q[3] *= -1; }
  const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) out7[k] = q[k] / nrm;
  for (int k = 0; k < 3; ++k) out7[4 + k] = in7[4 + k];
}

inline int pack_batch(int nw, const osh_lba_problem* pr, const std::function<void*(int, size_t)>& alloc, int n_threads, PackedBatch& pb,
                      bool allow_f32 = true) {
  using namespace pack_detail;
  if (pack_describe(nw, pr, pb) != OSH_OK) return pb.err;
  auto fail = [&](int code, const char* fmt, auto... a) {
    pb.err = code;
    if constexpr (sizeof...(a) == 0) std::snprintf(pb.msg, sizeof(pb.msg), "%s", fmt); else std::snprintf(pb.msg, sizeof(pb.msg), fmt, a...);
    return code;
  };
  n_threads = std::max(1, std::min(n_threads, nw));
  // Observation records travel as float32 when every value is one (what the reference stores: cv::KeyPoint::pt, mvuRight,
  // mvInvLevelSigma2).  Optimistic: the records are written as floats and checked on the way; a batch with a value that is not
  // a float32 (or a rig batch, which keeps doubles) is packed again with double records.
  pb.rec_f32 = allow_f32 && !pb.has_rig;
  std::atomic<int> inexact{0};
  {
    const size_t o = pack_layout0(pb);
    pb.arena[0] = static_cast<unsigned char*>(alloc(0, o));
    if (!pb.arena[0]) return fail(OSH_ERR_DEVICE, "cannot allocate %zu bytes of staging memory", o);
  }
  double* h_pose = pb.sec<double>(PackedBatch::POSE);
  double* h_cam = pb.sec<double>(PackedBatch::CAM);
  double* h_pt = pb.sec<double>(PackedBatch::PT);
  double* h_rec = pb.sec<double>(PackedBatch::EREC);
  float* h_recf = pb.sec<float>(PackedBatch::EREC);
  const bool rec_f32 = pb.rec_f32;
  double* h_rec2 = pb.sec<double>(PackedBatch::EREC2);
  int* h_epose = pb.sec<int>(PackedBatch::EPOSE);
  int* h_epoint = pb.sec<int>(PackedBatch::EPOINT);
  int* h_eorig = pb.sec<int>(PackedBatch::EORIG);
  int* h_eorig2 = pb.sec<int>(PackedBatch::EORIG2);
  int* h_lmoff = pb.sec<int>(PackedBatch::LMOFF);
  int* h_lmperm = pb.sec<int>(PackedBatch::LMPERM);
  int* h_fpw = pb.sec<int>(PackedBatch::FPW);
  unsigned char* h_kind = pb.sec<unsigned char>(PackedBatch::EKIND);

  std::vector<WinLocal> locals(nw);
  auto pack_window = [&](int w, Scratch& sc) {
    WinLocal& L = locals[w];
    const osh_lba_problem& p = pr[w];
    WinDesc& d = pb.win[w];
    auto lfail = [&](int code, const char* fmt, auto... a) { L.err = code; std::snprintf(L.msg, sizeof(L.msg), fmt, a...); };
    const int NPw = p.n_free + p.n_fixed;
    for (int i = 0; i < NPw; ++i) {
      pack_pose(p.pose_qt + 7 * (size_t)i, &h_pose[((size_t)d.pose_off + i) * 7]);
      for (int k = 0; k < 5; ++k) h_cam[((size_t)d.pose_off + i) * 5 + k] = p.pose_cam[5 * i + k];
    }
    for (int i = 0; i < p.n_free; ++i) h_fpw[(size_t)d.fpose_off + i] = w;
    // validation, landmark histogram and a sortedness check in one pass: the reference inserts its edges landmark by landmark
    // (src/Optimizer.cc:1295-1400, observers in the order of a std::map keyed by pointer), so a caller that keeps that order
    // needs no counting sort, only the short insertion sort of every landmark's observers
    std::vector<int>& cnt = sc.cnt; std::vector<int>& fill = sc.fill; std::vector<int>& order = sc.order;
    cnt.assign((size_t)p.n_points + 1, 0);
    bool presorted = true, lm_major = true;   // poses ascending inside every landmark as well / landmarks ascending
    for (int e = 0; e < p.n_edges; ++e) {
      const int ip = p.edge_pose[e], il = p.edge_point[e], kd = p.edge_kind[e];
      if (ip < 0 || ip >= NPw || il < 0 || il >= p.n_points || kd > OSH_EDGE_BODY) return lfail(OSH_ERR_INVALID, "window %d edge %d: index or kind out of range", w, e);
      if (p.kb8 && kd == OSH_EDGE_STEREO) return lfail(OSH_ERR_UNSUPPORTED, "window %d: a KannalaBrandt8 window takes monocular and body edges only (edge %d is a rectified-stereo edge)", w, e);
      if (kd == OSH_EDGE_BODY && !d.rig_on) return lfail(OSH_ERR_INVALID, "window %d edge %d: a body edge (EdgeSE3ProjectXYZToBody) needs kb8, cam2 and trl", w, e);
      cnt[il + 1]++;
      if (e > 0) {
        const int pl = p.edge_point[e - 1], pp = p.edge_pose[e - 1];
        lm_major &= il >= pl;
        presorted &= (il > pl) | ((il == pl) & ((ip > pp) | ((ip == pp) & (kd >= (int)p.edge_kind[e - 1]))));
      }
    }
    for (int j = 0; j < p.n_points; ++j) cnt[j + 1] += cnt[j];
    order.resize(p.n_edges);
    if (lm_major) {
      for (int e = 0; e < p.n_edges; ++e) order[e] = e;
    } else {
      // counting sort by landmark (stable), then order poses inside each landmark
      fill.assign(cnt.begin(), cnt.end() - 1);
      for (int e = 0; e < p.n_edges; ++e) order[fill[p.edge_point[e]]++] = e;
    }
    // per landmark: stable insertion sort by (pose, kind); a (pose, landmark) pair may carry a left (mono) and a right (body)
    // edge, which merge into one sorted edge; `second[x]` is the caller index of the merged right edge or -1
    std::vector<int>& second = sc.second; std::vector<int>& tmp_epose = sc.tmp_epose; std::vector<int>& nfree = sc.nfree;
    second.assign(p.n_edges, -1);
    tmp_epose.resize((size_t)p.n_edges + 1);
    nfree.assign(p.n_points, 0);
    std::vector<int>& old_lmo = sc.old_lmo;
    old_lmo.resize((size_t)p.n_points + 1);
    int n_sorted = 0;
    for (int j = 0; j < p.n_points; ++j) {
      const int lo = cnt[j], hi = cnt[j + 1];
      for (int x = lo + 1; x < hi && !presorted; ++x) {
        const int e = order[x], pe = p.edge_pose[e], ke = p.edge_kind[e];
        int y = x;
        for (; y > lo && (p.edge_pose[order[y - 1]] > pe || (p.edge_pose[order[y - 1]] == pe && p.edge_kind[order[y - 1]] > ke)); --y) order[y] = order[y - 1];
        order[y] = e;
      }
      old_lmo[j] = n_sorted;
      int nf = 0;
      for (int x = lo; x < hi; ++x) {
        const int e = order[x], pe = p.edge_pose[e];
        if (x > lo && pe == p.edge_pose[order[x - 1]]) {
          // second edge on the same Hessian block: only a body edge after a mono edge (fisheye rig) merges
          const int prev = order[x - 1];
          if (p.edge_kind[e] == OSH_EDGE_BODY && p.edge_kind[prev] == OSH_EDGE_MONO && second[n_sorted - 1] < 0) {
            second[n_sorted - 1] = e;
            continue;
          }
          return lfail(OSH_ERR_UNSUPPORTED, "window %d: landmark %d is observed twice by pose %d with edge kinds that do not form a "
                       "fisheye-rig pair (left EdgeSE3ProjectXYZ + right EdgeSE3ProjectXYZToBody)", w, j, pe);
        }
        order[n_sorted] = e;          // compaction in place (n_sorted <= x)
        second[n_sorted] = -1;
        tmp_epose[n_sorted] = pe;
        if (pe < p.n_free) ++nf;
        ++n_sorted;
      }
      nfree[j] = nf;
    }
    old_lmo[p.n_points] = n_sorted;
    tmp_epose[n_sorted] = 0;
    d.E = n_sorted;
    // Schur work plan on the old numbering (schur_plan.h)
    L.recs.reserve((size_t)p.n_points + p.n_points / 4);
    if (!plan_window(w, p.n_free, p.n_points, old_lmo.data(), nfree.data(), tmp_epose.data(), L.builds, L.recs, L.plan, sc.plan, item_max_lm(nw)))
      return lfail(OSH_ERR_UNSUPPORTED, "window %d: a landmark has more than 254 optimisable observers", w);
    // renumber the landmarks in the order of the plan's owner records (symmetric builds, in build order)
    std::vector<int>& old2new = sc.old2new;
    old2new.assign(p.n_points, -1);
    int* perm = &h_lmperm[d.pt_off];   // new -> old
    int next = 0;
    for (const plan_detail::Build& bd : L.builds) {
      if (!bd.sym) continue;
      for (int q = bd.rec_off; q < bd.rec_off + bd.n_rec; ++q) { const SRec& r = L.recs[q]; if (r.flags & 1) { old2new[r.lm] = next; perm[next] = r.lm; ++next; } }
    }
    if (next != p.n_points) return lfail(OSH_ERR_DEVICE, "window %d: plan owns %d of %d landmarks", w, next, p.n_points);
    int* lmo = &h_lmoff[d.lmoff_off];
    lmo[0] = 0;
    for (int jn = 0; jn < p.n_points; ++jn) { const int jo = perm[jn]; lmo[jn + 1] = lmo[jn] + (old_lmo[jo + 1] - old_lmo[jo]); }
    // The caller's arrays are walked in their own (old) order -- sequential reads -- and every landmark's run is written to
    // its renumbered place: scattered stores retire from the store buffer, scattered loads would each wait for memory.
    bool exact = true;
    for (int jo = 0; jo < p.n_points; ++jo) {
      const int jn = old2new[jo];
      for (int k = 0; k < 3; ++k) h_pt[((size_t)d.pt_off + jn) * 3 + k] = p.points[3 * (size_t)jo + k];
      int x = lmo[jn];
      for (int xo = old_lmo[jo]; xo < old_lmo[jo + 1]; ++xo, ++x) {
        const int e = order[xo], e2 = second[xo];
        const size_t g = (size_t)d.edge_off + x;
        const int kd = p.edge_kind[e];
        const int kind = e2 >= 0 ? kKindBoth : (kd == OSH_EDGE_BODY ? kKindBody : kd);
        h_epose[g] = p.edge_pose[e]; h_epoint[g] = jn; h_kind[g] = (unsigned char)kind; h_eorig[g] = e;
        // the sign of the information carries the edge kind for the pinhole kernels (negative = monocular)
        const double inf = (kd == OSH_EDGE_STEREO) ? p.edge_info[e] : -p.edge_info[e];
        if (rec_f32) {
          const double o0 = p.edge_obs[3 * (size_t)e], o1 = p.edge_obs[3 * (size_t)e + 1], o2 = p.edge_obs[3 * (size_t)e + 2];
          const float f0 = (float)o0, f1 = (float)o1, f2 = (float)o2, f3 = (float)inf;
          h_recf[g * 4] = f0; h_recf[g * 4 + 1] = f1; h_recf[g * 4 + 2] = f2; h_recf[g * 4 + 3] = f3;
          exact &= ((double)f0 == o0) & ((double)f1 == o1) & ((double)f2 == o2) & ((double)f3 == inf);
        } else {
          h_rec[g * 4] = p.edge_obs[3 * (size_t)e]; h_rec[g * 4 + 1] = p.edge_obs[3 * (size_t)e + 1]; h_rec[g * 4 + 2] = p.edge_obs[3 * (size_t)e + 2];
          h_rec[g * 4 + 3] = inf;
        }
        if (pb.has_rig) {
          h_eorig2[g] = e2;
          const int er = e2 >= 0 ? e2 : e;   // a lone body edge keeps its observation in the first record too
          h_rec2[g * 4] = p.edge_obs[3 * (size_t)er]; h_rec2[g * 4 + 1] = p.edge_obs[3 * (size_t)er + 1]; h_rec2[g * 4 + 2] = 0.0;
          h_rec2[g * 4 + 3] = p.edge_info[er];
        }
      }
    }
    if (!exact) inexact.store(1, std::memory_order_relaxed);
    for (SRec& r : L.recs) { r.lm = old2new[r.lm]; r.e_first = lmo[r.lm]; }
    for (const plan_detail::Build& bd : L.builds) {
      if (bd.sym) { L.n_sym++; L.recs_sym += bd.n_rec; } else { L.n_cross++; L.recs_cross += bd.n_rec; }
    }
    // chunks of the landmark-major kernels: consecutive landmarks, <= chunk_max_edges(nw) edges and <= kBlock landmarks (a single
    // landmark with more edges gets its own multi-pass chunk)
    const int chunk_edges = chunk_max_edges(nw);
    int j = 0;
    while (j < p.n_points) {
      int j1 = j + 1;
      while (j1 < p.n_points && (j1 - j) < kBlock && (lmo[j1 + 1] - lmo[j]) <= chunk_edges) ++j1;
      L.chunks.push_back(Chunk{w, j, j1});
      j = j1;
    }
  };
  {
    std::atomic<int> next{0};
    auto worker = [&]() {
      Scratch sc;
      for (int w = next.fetch_add(1); w < nw; w = next.fetch_add(1)) pack_window(w, sc);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (std::thread& t : pool) t.join();
  }
  for (int w = 0; w < nw; ++w) if (locals[w].err != OSH_OK) return fail(locals[w].err, "%s", locals[w].msg);
  if (pb.rec_f32 && inexact.load()) return pack_batch(nw, pr, alloc, n_threads, pb, false);

  // ---- phase 2: offsets of the plan sections, then a parallel merge
  struct WOff { size_t chunk, sym_item, cross_item, sym_rec, cross_rec, rblk, crange; int contrib, ccontrib; };
  std::vector<WOff> wo(nw + 1);
  {
    WOff a{0, 0, 0, 0, 0, 0, 0, 0, 0};
    size_t contrib = 0, ccontrib = 0;
    for (int w = 0; w < nw; ++w) {
      wo[w] = a; wo[w].contrib = (int)contrib; wo[w].ccontrib = (int)ccontrib;
      const WinLocal& L = locals[w];
      a.chunk += L.chunks.size(); a.sym_item += L.n_sym; a.cross_item += L.n_cross; a.sym_rec += L.recs_sym; a.cross_rec += L.recs_cross;
      a.rblk += L.plan.rblk.size(); a.crange += (size_t)pb.win[w].P;
      contrib += L.plan.n_contrib; ccontrib += L.plan.n_ccontrib;
      pb.tile_steps += L.plan.tile_steps; pb.pair_blocks += L.plan.pair_blocks;
    }
    wo[nw] = a;
    pb.n_chunks = a.chunk; pb.n_sym = a.sym_item; pb.n_items = a.sym_item + a.cross_item; pb.n_recs = a.sym_rec + a.cross_rec;
    pb.n_rblk = a.rblk; pb.n_contrib = contrib; pb.n_ccontrib = ccontrib;
    if (contrib > 0x7fffff00u / 36 * 16 || pb.n_recs > 0x7fffff00u) return fail(OSH_ERR_UNSUPPORTED, "batch too large for 32-bit contribution offsets");
    const size_t o = pack_layout1(pb);
    pb.arena[1] = static_cast<unsigned char*>(alloc(1, o));
    if (!pb.arena[1]) return fail(OSH_ERR_DEVICE, "cannot allocate %zu bytes of staging memory", o);
  }
  Chunk* h_chunks = pb.sec<Chunk>(PackedBatch::CHUNKS);
  SItem* h_items = pb.sec<SItem>(PackedBatch::ITEMS);
  SRec* h_recs = pb.sec<SRec>(PackedBatch::RECS);
  int* h_spair = pb.sec<int>(PackedBatch::SPAIR);
  int* h_scslot = pb.sec<int>(PackedBatch::SCSLOT);
  int* h_posex = pb.sec<int>(PackedBatch::POSEX);
  int* h_posey = pb.sec<int>(PackedBatch::POSEY);
  RBlk* h_rblk = pb.sec<RBlk>(PackedBatch::RBLK);
  I2* h_crange = pb.sec<I2>(PackedBatch::CRANGE);
  const size_t n_sym_total = pb.n_sym, sym_recs_total = wo[nw].sym_rec;
  auto merge_window = [&](int w) {
    WinLocal& L = locals[w];
    WinDesc& d = pb.win[w];
    const WOff& o = wo[w];
    d.chunk_off = (int)o.chunk; d.n_chunks = (int)L.chunks.size();
    if (!L.chunks.empty()) std::memcpy(h_chunks + o.chunk, L.chunks.data(), L.chunks.size() * sizeof(Chunk));
    d.sitem_off = (int)o.sym_item; d.n_sitems = L.n_sym;
    size_t is = o.sym_item, ic = n_sym_total + o.cross_item, rs = o.sym_rec, rc = sym_recs_total + o.cross_rec;
    for (const plan_detail::Build& bd : L.builds) {
      const size_t it = bd.sym ? is++ : ic++;
      size_t& r = bd.sym ? rs : rc;
      SItem I;
      I.win = w; I.rec_off = (int)r; I.n_lm = bd.n_rec;
      I.shape = bd.nx | (bd.ny << 8) | ((bd.sym ? 1 : 0) << 16);
      h_items[it] = I;
      std::memcpy(h_recs + r, L.recs.data() + bd.rec_off, (size_t)bd.n_rec * sizeof(SRec));
      r += bd.n_rec;
      for (int k = 0; k < 64; ++k) h_spair[it * 64 + k] = bd.pair_slot[k] >= 0 ? bd.pair_slot[k] + o.contrib : -1;
      for (int k = 0; k < 8; ++k) {
        h_scslot[it * 8 + k] = bd.c_slot[k] >= 0 ? bd.c_slot[k] + o.ccontrib : -1;
        h_posex[it * 8 + k] = bd.X[k];
        h_posey[it * 8 + k] = bd.Y[k];
      }
    }
    size_t rb_i = o.rblk, cr_i = o.crange;
    for (RBlk rb : L.plan.rblk) {
      const bool rhs = ((rb.ij >> 16) & 0xffff) == 0xffff;
      rb.start += rhs ? o.ccontrib : o.contrib;
      h_rblk[rb_i++] = rb;
      if (rhs) h_crange[cr_i++] = I2{rb.start, rb.count};
    }
    L = WinLocal();
  };
  {
    std::atomic<int> next{0};
    auto worker = [&]() { for (int w = next.fetch_add(1); w < nw; w = next.fetch_add(1)) merge_window(w); };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (std::thread& t : pool) t.join();
  }
  std::memcpy(pb.sec<WinDesc>(PackedBatch::WIN), pb.win.data(), (size_t)nw * sizeof(WinDesc));
  return OSH_OK;
}

inline int default_pack_threads(int nw) {
  if (nw <= 1) return 1;
  const char* env = std::getenv("ORBSLAM3_HIP_UPLOAD_THREADS");
  const unsigned hw = std::thread::hardware_concurrency();
  int n = env ? std::atoi(env) : (int)std::min<unsigned>(hw ? hw : 1u, 16u);
  return std::max(1, std::min(n, nw));
}

}  // namespace osh
