// lba_pack_check.cpp -- host-only self check + timing of the batch packer (lba_pack.h).  Needs no GPU.
// Packs the problems into malloc'ed arenas exactly as osh_lba_upload packs them into pinned staging and verifies the
// layout the kernels rely on: the landmark renumbering is a permutation that follows the plan's owner records, every
// sorted edge is the caller's edge it claims to be (pose, renumbered landmark, observation record, sign-coded kind, merged
// fisheye-rig pairs), every record's slots point at edges of the item's poses, and contribution slots stay inside the
// ranges of their blocks after the per-window rebasing.
#include "common.h"
#include "lba_pack.h"
#include <chrono>
#include <cstdlib>

using namespace osh;

extern "C" int osh_lba_pack_check(int32_t nw, const osh_lba_problem* pr, int32_t n_threads, int64_t stats[8], double* pack_ms) {
  if (nw <= 0 || !pr || !stats) { set_error("osh_lba_pack_check: bad arguments"); return OSH_ERR_INVALID; }
  void* mem[2] = {nullptr, nullptr};
  size_t cap[2] = {0, 0};
  PackedBatch pb;
  // grow-only staging like the upload's pinned arenas; the timed pass is the second one (staging already touched)
  auto alloc = [&](int which, size_t bytes) { if (bytes > cap[which]) { std::free(mem[which]); mem[which] = std::malloc(bytes); cap[which] = bytes; } return mem[which]; };
  const int nt = n_threads > 0 ? n_threads : default_pack_threads(nw);
  int rc = pack_batch(nw, pr, alloc, nt, pb);
  const auto t0 = std::chrono::steady_clock::now();
  if (rc == OSH_OK && pack_ms) rc = pack_batch(nw, pr, alloc, nt, pb);
  const auto t1 = std::chrono::steady_clock::now();
  if (pack_ms) *pack_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  auto done = [&](int code) { std::free(mem[0]); std::free(mem[1]); return code; };
  if (rc != OSH_OK) { set_error("%s", pb.msg); return done(rc); }
#define CHECK(cond, ...) do { if (!(cond)) { set_error(__VA_ARGS__); return done(OSH_ERR_DEVICE); } } while (0)
  // the checks below read the records as doubles: widen a float32 batch first
  std::vector<double> rec_wide;
  if (pb.rec_f32) {
    const float* rf = pb.sec<float>(PackedBatch::EREC);
    rec_wide.assign(rf, rf + pb.NE * 4);
  }
  const double* rec = pb.rec_f32 ? rec_wide.data() : pb.sec<double>(PackedBatch::EREC);
  const double* rec2 = pb.sec<double>(PackedBatch::EREC2);
  const int* epose = pb.sec<int>(PackedBatch::EPOSE);
  const int* epoint = pb.sec<int>(PackedBatch::EPOINT);
  const int* eorig = pb.sec<int>(PackedBatch::EORIG);
  const int* eorig2 = pb.sec<int>(PackedBatch::EORIG2);
  const unsigned char* ekind = pb.sec<unsigned char>(PackedBatch::EKIND);
  const int* lmoff = pb.sec<int>(PackedBatch::LMOFF);
  const int* perm = pb.sec<int>(PackedBatch::LMPERM);
  const double* pt = pb.sec<double>(PackedBatch::PT);
  const SItem* items = pb.sec<SItem>(PackedBatch::ITEMS);
  const SRec* recs = pb.sec<SRec>(PackedBatch::RECS);
  const int* spair = pb.sec<int>(PackedBatch::SPAIR);
  const int* scslot = pb.sec<int>(PackedBatch::SCSLOT);
  const int* posex = pb.sec<int>(PackedBatch::POSEX);
  const int* posey = pb.sec<int>(PackedBatch::POSEY);
  const RBlk* rblk = pb.sec<RBlk>(PackedBatch::RBLK);
  const Chunk* chunks = pb.sec<Chunk>(PackedBatch::CHUNKS);
  long long merged = 0;
  for (int w = 0; w < nw; ++w) {
    const osh_lba_problem& p = pr[w];
    const WinDesc& d = pb.win[w];
    CHECK(d.P == p.n_free && d.F == p.n_fixed && d.L == p.n_points && d.in_edges == p.n_edges, "window %d: sizes", w);
    // permutation + landmark data
    std::vector<char> seen(p.n_points, 0);
    for (int jn = 0; jn < p.n_points; ++jn) {
      const int jo = perm[d.pt_off + jn];
      CHECK(jo >= 0 && jo < p.n_points && !seen[jo], "window %d: landmark renumbering is not a permutation", w);
      seen[jo] = 1;
      for (int k = 0; k < 3; ++k) CHECK(pt[((size_t)d.pt_off + jn) * 3 + k] == p.points[3 * (size_t)jo + k], "window %d: landmark %d moved without its position", w, jn);
    }
    const int* lmo = lmoff + d.lmoff_off;
    CHECK(lmo[0] == 0 && lmo[p.n_points] == d.E, "window %d: landmark offsets", w);
    std::vector<char> used(p.n_edges, 0);
    for (int jn = 0; jn < p.n_points; ++jn) {
      int prev_pose = -1;
      bool seen_fixed = false;
      for (int x = lmo[jn]; x < lmo[jn + 1]; ++x) {
        const size_t g = (size_t)d.edge_off + x;
        const int e = eorig[g];
        CHECK(e >= 0 && e < p.n_edges && !used[e], "window %d: sorted edge %d has a bad caller index", w, x);
        used[e] = 1;
        CHECK(epoint[g] == jn && perm[d.pt_off + jn] == p.edge_point[e] && epose[g] == p.edge_pose[e], "window %d: sorted edge %d is not its caller edge", w, x);
        CHECK(epose[g] > prev_pose, "window %d: landmark %d: poses not strictly ascending", w, jn);
        prev_pose = epose[g];
        if (epose[g] >= p.n_free) seen_fixed = true; else CHECK(!seen_fixed, "window %d: optimisable pose after a fixed one", w);
        const int kd = p.edge_kind[e];
        int kind = kd == OSH_EDGE_BODY ? kKindBody : kd;
        if (pb.has_rig && eorig2[g] >= 0) {
          const int e2 = eorig2[g];
          CHECK(e2 < p.n_edges && !used[e2] && p.edge_kind[e2] == OSH_EDGE_BODY && kd == OSH_EDGE_MONO && p.edge_pose[e2] == p.edge_pose[e] && p.edge_point[e2] == p.edge_point[e],
                "window %d: merged edge %d is not a left/right pair", w, x);
          used[e2] = 1;
          kind = kKindBoth;
          ++merged;
          CHECK(rec2[g * 4] == p.edge_obs[3 * (size_t)e2] && rec2[g * 4 + 1] == p.edge_obs[3 * (size_t)e2 + 1] && rec2[g * 4 + 3] == p.edge_info[e2], "window %d: right record of edge %d", w, x);
        }
        CHECK(ekind[g] == kind, "window %d: kind of sorted edge %d", w, x);
        CHECK(rec[g * 4] == p.edge_obs[3 * (size_t)e] && rec[g * 4 + 1] == p.edge_obs[3 * (size_t)e + 1] && rec[g * 4 + 2] == p.edge_obs[3 * (size_t)e + 2], "window %d: observation of edge %d", w, x);
        CHECK(std::fabs(rec[g * 4 + 3]) == p.edge_info[e] && (rec[g * 4 + 3] > 0) == (kd == OSH_EDGE_STEREO), "window %d: information / kind sign of edge %d", w, x);
      }
    }
    for (int e = 0; e < p.n_edges; ++e) CHECK(used[e], "window %d: caller edge %d lost", w, e);
    // chunks tile the landmarks
    int next_lm = 0;
    for (int c = 0; c < d.n_chunks; ++c) {
      const Chunk& ch = chunks[d.chunk_off + c];
      CHECK(ch.win == w && ch.lm0 == next_lm && ch.lm1 > ch.lm0 && (ch.lm1 - ch.lm0) <= kBlock, "window %d: chunk %d", w, c);
      CHECK(ch.lm1 - ch.lm0 == 1 || lmo[ch.lm1] - lmo[ch.lm0] <= kChunkMaxEdges, "window %d: chunk %d too long", w, c);
      next_lm = ch.lm1;
    }
    CHECK(next_lm == p.n_points, "window %d: chunks do not cover the landmarks", w);
  }
  // plan: records against the renumbered structure, slots against the block ranges
  std::vector<int> blk_lo(pb.n_contrib ? pb.n_contrib : 1, 0);
  std::vector<char> cused(pb.n_contrib, 0), ccused(pb.n_ccontrib, 0);
  long long next_owner_w = -1, next_owner_lm = 0;
  for (size_t it = 0; it < pb.n_items; ++it) {
    const SItem& I = items[it];
    const bool sym = (I.shape >> 16) & 1;
    CHECK(sym == (it < pb.n_sym) && I.win >= 0 && I.win < nw, "item %zu: symmetry / window", it);
    const WinDesc& d = pb.win[I.win];
    const int* lmo = lmoff + d.lmoff_off;
    if (sym && I.win != next_owner_w) { next_owner_w = I.win; next_owner_lm = 0; }
    for (int r = 0; r < I.n_lm; ++r) {
      const SRec& R = recs[(size_t)I.rec_off + r];
      CHECK(R.lm >= 0 && R.lm < d.L && R.e_first == lmo[R.lm] && R.pad == lmo[R.lm + 1] - lmo[R.lm], "item %zu record %d: landmark", it, r);
      if (R.flags & 1) { CHECK(sym && R.lm == next_owner_lm, "item %zu record %d: owner records must number the landmarks consecutively", it, r); ++next_owner_lm; }
      const unsigned long long xs = R.x_lo | ((unsigned long long)R.x_hi << 32), ys = R.y_lo | ((unsigned long long)R.y_hi << 32);
      for (int s = 0; s < 8; ++s) {
        const unsigned ra = (unsigned)(xs >> (8 * s)) & 0xff, rb = (unsigned)(ys >> (8 * s)) & 0xff;
        if (ra != kAbsent) CHECK((int)ra < R.pad && epose[(size_t)d.edge_off + R.e_first + ra] == posex[it * 8 + s], "item %zu record %d: row slot %d", it, r, s);
        if (rb != kAbsent) CHECK((int)rb < R.pad && epose[(size_t)d.edge_off + R.e_first + rb] == posey[it * 8 + s], "item %zu record %d: column slot %d", it, r, s);
      }
    }
    for (int k = 0; k < 64; ++k) { const int sl = spair[it * 64 + k]; if (sl >= 0) { CHECK((size_t)sl < pb.n_contrib && !cused[sl], "item %zu: contribution slot", it); cused[sl] = 1; } }
    for (int k = 0; k < 8; ++k) { const int sl = scslot[it * 8 + k]; if (sl >= 0) { CHECK((size_t)sl < pb.n_ccontrib && !ccused[sl], "item %zu: rhs slot", it); ccused[sl] = 1; } }
  }
  for (size_t k = 0; k < pb.n_contrib; ++k) CHECK(cused[k], "contribution slot %zu unused", k);
  for (size_t k = 0; k < pb.n_ccontrib; ++k) CHECK(ccused[k], "rhs slot %zu unused", k);
  size_t csum = 0, ccsum = 0;
  for (size_t b = 0; b < pb.n_rblk; ++b) {
    const bool rhs = ((rblk[b].ij >> 16) & 0xffff) == 0xffff;
    if (rhs) { CHECK((size_t)rblk[b].start == ccsum, "rhs range %zu not contiguous", b); ccsum += rblk[b].count; }
    else { CHECK((size_t)rblk[b].start == csum, "block range %zu not contiguous", b); csum += rblk[b].count; }
  }
  CHECK(csum == pb.n_contrib && ccsum == pb.n_ccontrib, "block ranges do not cover the contributions");
#undef CHECK
  stats[0] = (int64_t)pb.n_items; stats[1] = (int64_t)pb.n_sym; stats[2] = (int64_t)pb.n_recs; stats[3] = (int64_t)pb.n_contrib;
  stats[4] = (int64_t)pb.n_chunks; stats[5] = (int64_t)(pb.arena_bytes[0] + pb.arena_bytes[1]); stats[6] = merged; stats[7] = (int64_t)pb.n_rblk;
  return done(OSH_OK);
}
