// schur_plan.h -- host-side work plan of the Schur complement (pure C++, no device code).
//
// The reference forms S = Hpp - sum_l  B_l Dinv_l B_l^T one landmark at a time
// (Thirdparty/g2o/g2o/core/block_solver.hpp:381-432).  On the device the landmarks of a window
// are grouped into ITEMS of landmarks that share (nearly) the same set of optimisable observers:
// inside an item the products BD_a W_b^T of all observer pairs are one small dense GEMM
//   D[6 nx x 6 ny] = BD[6 nx x 3 n_lm] * W^T[3 n_lm x 6 ny]
// which k_schur_items runs on the FP64 matrix cores, one wavefront per item.  An item has at
// most 8 row poses X and 8 column poses Y (three 16x16 MFMA tiles per side).  A landmark with
// more than 8 optimisable observers is cut into parts of <= 8 consecutive observers; part pair
// (a,a) goes to a symmetric item (X == Y, upper tiles only), part pair (a<b) to a cross item.
// Every item writes one 6x6 contribution per live pose pair; k_schur_reduce sums the
// contributions of every block of S in plan order, so the result does not depend on scheduling.
//
// The plan depends only on the graph structure: it is built once per upload and reused by every
// Levenberg-Marquardt trial.
#pragma once
#include <algorithm>
#include <array>
#include <climits>
#include <cstdint>
#include <iterator>
#include <utility>
#include <cstdlib>
#include <vector>

namespace osh {

constexpr int kItemPoses = 8;     // row / column poses of one item
constexpr int kItemMaxLm = 64;    // landmarks per item at most (load balance; partial sums are per item)
// Landmarks per item for a batch of nw windows.  One wavefront walks an item's chunks one after the other, so in a call with one or two
// windows -- the live-SLAM pattern, where every item has a CU to itself anyway -- an item of 64 landmarks is the latency of the Schur
// launches (28 us per launch for the cross items of a config-2 window): such calls cut their items at 24 landmarks, up to eight
// windows at 32.  More items mean more 6x6 contributions for k_schur_reduce and k_pose_reduce, which is why a large batch keeps 64
// (one config-2 window, optimize(): 2.55 ms at 64, 2.40 at 32, 2.34 at 24, 2.37 at 16, 2.41 at 8).
inline int item_max_lm(int nw) {
  if (const char* e = std::getenv("OSH_LBA_ITEM_MAX")) { const int v = std::atoi(e); if (v >= 8 && v <= kItemMaxLm) return v; }   // (tuning aid)
  return nw <= 2 ? 24 : (nw <= 8 ? 32 : kItemMaxLm);
}
constexpr unsigned kAbsent = 0xffu;

struct SItem { int win, rec_off, n_lm, shape; };  // shape = nx | ny << 8 | sym << 16
struct SRec {
  int lm, e_first;                // window-local landmark, its first sorted edge (window-local)
  unsigned x_lo, x_hi;            // byte s = rank of the edge of row pose X[s] among the landmark's edges, 0xff absent
  unsigned y_lo, y_hi;            // same for the column poses (== x for symmetric items)
  int flags, pad;                 // flags bit 0: this record owns the landmark (Hll, b_l, chi2, dinv), bits 8-15: its
                                  // observers in part 0, bits 16-23 / 24-31: first rank / number of row observers of
                                  // this record (consecutive ranks); pad: all edges of the landmark (optimisable + fixed)
};
struct RBlk { int win, ij, start, count; };  // block (i,j) of S: ij = i | j << 16; j == 0xffff: the rhs segment of pose i

struct SchurPlan {
  std::vector<SItem> items;       // symmetric items first
  int n_sym = 0;
  std::vector<SRec> recs;
  std::vector<int> pair_slot;     // [items][64]: contribution index of pose pair (sa, sb) or -1
  std::vector<int> c_slot;        // [items][8]: rhs contribution index of row pose sa or -1 (symmetric items)
  std::vector<int> pose_x, pose_y;  // [items][8] window-local pose of each slot (-1 unused): checks / debugging
  std::vector<RBlk> rblk;         // every block of every window + one rhs segment per pose
  size_t n_contrib = 0, n_ccontrib = 0;
  // statistics
  long long tile_steps = 0;       // MFMA instructions of one pass
  long long pair_blocks = 0;      // useful 6x6 products (upper triangle, per landmark)
};

namespace plan_detail {

inline int tiles_of(int n_poses) { return (6 * n_poses + 15) / 16; }

// Sort key of one (landmark, part pair): sym flag, X poses, Y poses, compared lexicographically with absent slots last.
// Poses are packed 16 bits each, most significant first, so four 64-bit compares order a key (pose indices < 65535).
struct Unit {
  uint64_t k[5];   // k[0]: 0 symmetric / 1 cross; k[1..2]: X poses; k[3..4]: Y poses (0xffff padded)
  int lm, a, b;
  bool same_key(const Unit& o) const { return k[0] == o.k[0] && k[1] == o.k[1] && k[2] == o.k[2] && k[3] == o.k[3] && k[4] == o.k[4]; }
};
inline bool unit_less(const Unit& p, const Unit& q) {
  for (int i = 0; i < 5; ++i) if (p.k[i] != q.k[i]) return p.k[i] < q.k[i];
  if (p.lm != q.lm) return p.lm < q.lm;      // creation order (what a stable sort on the key alone would keep)
  if (p.a != q.a) return p.a < q.a;
  return p.b < q.b;
}
inline void pack_poses(const int* obs, int r0, int r1, uint64_t out[2]) {
  out[0] = out[1] = ~0ull;
  for (int r = r0; r < r1; ++r) {
    const int s = r - r0, w = s >> 2, sh = 48 - 16 * (s & 3);
    out[w] = (out[w] & ~(0xffffull << sh)) | ((uint64_t)(unsigned)obs[r] << sh);
  }
}

// One item being built.  The records of all builds of a window live in one flat array (PlanScratch::recs / the caller's),
// [rec_off, rec_off + n_rec): no per-build allocations (a window has thousands of builds and windows are planned in parallel).
struct Build {
  bool sym = true;
  int nx = 0, ny = 0;
  int X[kItemPoses], Y[kItemPoses];
  int rec_off = 0, n_rec = 0;
  int pair_slot[64];
  int c_slot[8];
};

// Reusable buffers of plan_window (one per packing thread)
struct PlanScratch {
  std::vector<Unit> units;
  std::vector<int> table, rep, count, gid, gorder, uorder, cnt, ccnt, fill, cfill;
  std::vector<size_t> start;
};

// union of two ascending pose lists (<= 8 entries each) into out (<= 16); returns its length
inline int set_union(const int* a, int na, const int* b, int nb, int* out) {
  int i = 0, j = 0, n = 0;
  while (i < na && j < nb) { if (a[i] < b[j]) out[n++] = a[i++]; else if (b[j] < a[i]) out[n++] = b[j++]; else { out[n++] = a[i++]; ++j; } }
  while (i < na) out[n++] = a[i++];
  while (j < nb) out[n++] = b[j++];
  return n;
}
inline int unpack_poses(const uint64_t in[2], int* out) {
  int n = 0;
  for (int s = 0; s < kItemPoses; ++s) {
    const unsigned v = (unsigned)((in[s >> 2] >> (48 - 16 * (s & 3))) & 0xffff);
    if (v != 0xffff) out[n++] = (int)v;
  }
  return n;
}

inline unsigned long long pack_slots(const int* S, int ns, const int* obs, int r0, int r1) {
  unsigned long long v = ~0ull;
  for (int r = r0; r < r1; ++r) {
    const int slot = (int)(std::lower_bound(S, S + ns, obs[r]) - S);
    v &= ~(0xffull << (8 * slot));
    v |= (unsigned long long)(unsigned)r << (8 * slot);
  }
  return v;
}

}  // namespace plan_detail

// Adds the items of window `w` to `plan` (items of all windows are re-ordered by finish_plan).
//   P: optimisable poses; L: landmarks; lmo[L+1]: sorted-edge offsets; nfree[L]: optimisable-pose
//   edges of each landmark (they come first, poses ascending); epose: pose of every sorted edge.
//   out_builds / out_recs: the window's items and their records (appended); sc: reusable buffers.
// Returns false when a landmark has more optimisable observers than a record can index.
inline bool plan_window(int w, int P, int L, const int* lmo, const int* nfree, const int* epose,
                        std::vector<plan_detail::Build>& out_builds, std::vector<SRec>& out_recs, SchurPlan& plan,
                        plan_detail::PlanScratch& sc, const int item_max = kItemMaxLm) {
  using namespace plan_detail;
  std::vector<Unit>& units = sc.units;
  units.clear();
  if (units.capacity() < (size_t)L + L / 2) units.reserve((size_t)L + L / 2);
  if (P >= 0xffff) return false;
  for (int j = 0; j < L; ++j) {
    const int k = nfree[j];
    if (k > 254) return false;
    const int* obs = epose + lmo[j];
    if (k <= kItemPoses) {            // one symmetric unit (the common case)
      Unit u;
      u.k[0] = 0;
      pack_poses(obs, 0, k, &u.k[1]);
      u.k[3] = u.k[1]; u.k[4] = u.k[2];
      u.lm = j; u.a = 0; u.b = 0;
      units.push_back(u);
    } else {
      const int nparts = (k + kItemPoses - 1) / kItemPoses;
      for (int a = 0; a < nparts; ++a)
        for (int b = a; b < nparts; ++b) {
          Unit u;
          u.k[0] = (a == b) ? 0 : 1;
          const int a0 = a * kItemPoses, a1 = std::min(k, a0 + kItemPoses);
          const int b0 = b * kItemPoses, b1 = std::min(k, b0 + kItemPoses);
          pack_poses(obs, a0, a1, &u.k[1]);
          pack_poses(obs, b0, b1, &u.k[3]);
          u.lm = j; u.a = a; u.b = b;
          units.push_back(u);
        }
    }
    plan.pair_blocks += (long long)k * (k + 1) / 2;
  }
  // Order by unit_less.  A window has few distinct keys (hundreds, against tens of thousands of units), so the units are
  // bucketed by key with a small open-addressing table, only the distinct keys are sorted, and the units of a key keep their
  // creation order (lm, a, b ascending), which is unit_less's tie-break.  `uorder` is the resulting permutation.
  const size_t nu = units.size();
  std::vector<int>& uorder = sc.uorder;
  {
    size_t cap = 64;
    while (cap < 2 * nu) cap <<= 1;
    sc.table.assign(cap, -1);          // slot -> group
    sc.rep.clear(); sc.count.clear();  // group -> first unit with that key, number of units
    sc.gid.resize(nu);
    for (size_t i = 0; i < nu; ++i) {
      const Unit& u = units[i];
      uint64_t h = u.k[0] * 0x9e3779b97f4a7c15ull;
      for (int q = 1; q < 5; ++q) { h ^= u.k[q]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
      size_t slot = (size_t)h & (cap - 1);
      while (sc.table[slot] >= 0 && !units[sc.rep[sc.table[slot]]].same_key(u)) slot = (slot + 1) & (cap - 1);
      if (sc.table[slot] < 0) { sc.table[slot] = (int)sc.rep.size(); sc.rep.push_back((int)i); sc.count.push_back(0); }
      sc.gid[i] = sc.table[slot];
      sc.count[sc.gid[i]]++;
    }
    const size_t ng = sc.rep.size();
    sc.gorder.resize(ng);
    for (size_t g = 0; g < ng; ++g) sc.gorder[g] = (int)g;
    std::sort(sc.gorder.begin(), sc.gorder.end(), [&](int p, int q) { return unit_less(units[sc.rep[p]], units[sc.rep[q]]); });
    sc.start.resize(ng);
    size_t acc = 0;
    for (int g : sc.gorder) { sc.start[g] = acc; acc += (size_t)sc.count[g]; }
    uorder.resize(nu);
    for (size_t i = 0; i < nu; ++i) uorder[sc.start[sc.gid[i]]++] = (int)i;
  }

  const size_t first_build = out_builds.size();
  int curX[2 * kItemPoses], curY[2 * kItemPoses], ncx = 0, ncy = 0;
  size_t cur0 = 0, cur1 = 0;      // [cur0, cur1) of uorder: the units of the item(s) being collected
  bool cur_sym = true;
  auto flush = [&]() {
    for (size_t base = cur0; base < cur1; base += (size_t)item_max) {
      out_builds.emplace_back();
      Build& bd = out_builds.back();
      bd.sym = cur_sym; bd.nx = ncx; bd.ny = ncy;
      for (int s2 = 0; s2 < kItemPoses; ++s2) { bd.X[s2] = s2 < ncx ? curX[s2] : -1; bd.Y[s2] = s2 < ncy ? curY[s2] : -1; }
      std::fill(bd.pair_slot, bd.pair_slot + 64, -1);
      std::fill(bd.c_slot, bd.c_slot + 8, -1);
      const size_t end = std::min(cur1, base + (size_t)item_max);
      bd.rec_off = (int)out_recs.size(); bd.n_rec = (int)(end - base);
      const Unit* prev = nullptr;
      unsigned long long xs = 0, ys = 0;
      for (size_t x = base; x < end; ++x) {
        const Unit& u = units[uorder[x]];
        const int k = nfree[u.lm];
        const int* obs = epose + lmo[u.lm];
        const int a0 = u.a * kItemPoses, a1 = std::min(k, a0 + kItemPoses);
        const int b0 = u.b * kItemPoses, b1 = std::min(k, b0 + kItemPoses);
        // units with the same key and part pair have the same slot assignment and mark the same live pairs
        const bool same = prev && prev->same_key(u) && prev->a == u.a && prev->b == u.b;
        if (!same) { xs = pack_slots(curX, ncx, obs, a0, a1); ys = pack_slots(curY, ncy, obs, b0, b1); }
        prev = &u;
        SRec r;
        r.lm = u.lm; r.e_first = lmo[u.lm];
        r.x_lo = (unsigned)xs; r.x_hi = (unsigned)(xs >> 32); r.y_lo = (unsigned)ys; r.y_hi = (unsigned)(ys >> 32);
        r.flags = ((u.a == 0 && u.b == 0) ? (1 | (std::min(k, kItemPoses) << 8)) : 0) | (a0 << 16) | ((a1 - a0) << 24);
        r.pad = lmo[u.lm + 1] - lmo[u.lm];
        out_recs.push_back(r);
        // live pairs (marked with -2, numbered later)
        if (same) continue;
        for (int sa = 0; sa < kItemPoses; ++sa) {
          if (((xs >> (8 * sa)) & 0xff) == kAbsent) continue;
          if (cur_sym) bd.c_slot[sa] = -2;
          for (int sb = cur_sym ? sa : 0; sb < kItemPoses; ++sb)
            if (((ys >> (8 * sb)) & 0xff) != kAbsent) bd.pair_slot[sa * 8 + sb] = -2;
        }
      }
      const int tx = tiles_of(ncx), ty = tiles_of(ncy);
      const long long tiles = cur_sym ? (long long)tx * (tx + 1) / 2 : (long long)tx * ty;
      plan.tile_steps += tiles * 6 * (long long)((end - base + 7) / 8);
    }
    cur0 = cur1;
  };
  int gX[kItemPoses], gY[kItemPoses], ux[2 * kItemPoses], uy[2 * kItemPoses];
  size_t x = 0;
  while (x < nu) {
    const Unit& u0 = units[uorder[x]];
    size_t x1 = x + 1;
    while (x1 < nu && units[uorder[x1]].same_key(u0)) ++x1;
    const int ngx = unpack_poses(&u0.k[1], gX), ngy = unpack_poses(&u0.k[3], gY);
    const bool g_sym = u0.k[0] == 0;
    bool merged = false;
    if (cur1 > cur0 && cur_sym == g_sym && (cur1 - cur0) < (size_t)item_max) {
      const int nux = set_union(curX, ncx, gX, ngx, ux), nuy = set_union(curY, ncy, gY, ngy, uy);
      if (nux <= kItemPoses && nuy <= kItemPoses && tiles_of(nux) == tiles_of(ncx) && tiles_of(nux) == tiles_of(ngx) &&
          tiles_of(nuy) == tiles_of(ncy) && tiles_of(nuy) == tiles_of(ngy)) {
        std::copy(ux, ux + nux, curX); ncx = nux;
        std::copy(uy, uy + nuy, curY); ncy = nuy;
        merged = true;
      }
    }
    if (!merged) {
      if (cur1 > cur0) flush();
      std::copy(gX, gX + ngx, curX); ncx = ngx;
      std::copy(gY, gY + ngy, curY); ncy = ngy;
      cur_sym = g_sym;
    }
    cur1 = x1;
    x = x1;
  }
  if (cur1 > cur0) flush();

  // contribution slots: the contributions of one block of S are contiguous, in item order
  const int nblk = P * (P + 1) / 2;
  auto blk = [P](int i, int j) { return i * P - i * (i - 1) / 2 + (j - i); };
  std::vector<int>& cnt = sc.cnt; std::vector<int>& ccnt = sc.ccnt;
  cnt.assign((size_t)nblk + 1, 0); ccnt.assign((size_t)P + 1, 0);
  for (size_t b = first_build; b < out_builds.size(); ++b) {
    const Build& bd = out_builds[b];
    for (int sa = 0; sa < 8; ++sa) {
      if (bd.c_slot[sa] == -2) ccnt[bd.X[sa] + 1]++;
      for (int sb = 0; sb < 8; ++sb) if (bd.pair_slot[sa * 8 + sb] == -2) cnt[blk(bd.X[sa], bd.Y[sb]) + 1]++;
    }
  }
  for (int k = 0; k < nblk; ++k) cnt[k + 1] += cnt[k];
  for (int k = 0; k < P; ++k) ccnt[k + 1] += ccnt[k];
  const size_t c0 = plan.n_contrib, cc0 = plan.n_ccontrib;
  plan.rblk.reserve(plan.rblk.size() + (size_t)nblk + P);
  for (int i = 0; i < P; ++i)
    for (int j = i; j < P; ++j) {
      const int k = blk(i, j);
      plan.rblk.push_back(RBlk{w, i | (j << 16), (int)(c0 + cnt[k]), cnt[k + 1] - cnt[k]});
    }
  for (int i = 0; i < P; ++i) plan.rblk.push_back(RBlk{w, i | (0xffff << 16), (int)(cc0 + ccnt[i]), ccnt[i + 1] - ccnt[i]});
  sc.fill.assign(cnt.begin(), cnt.end() - 1); sc.cfill.assign(ccnt.begin(), ccnt.end() - 1);
  for (size_t b = first_build; b < out_builds.size(); ++b) {
    Build& bd = out_builds[b];
    for (int sa = 0; sa < 8; ++sa) {
      if (bd.c_slot[sa] == -2) bd.c_slot[sa] = (int)(cc0 + sc.cfill[bd.X[sa]]++);
      for (int sb = 0; sb < 8; ++sb)
        if (bd.pair_slot[sa * 8 + sb] == -2) bd.pair_slot[sa * 8 + sb] = (int)(c0 + sc.fill[blk(bd.X[sa], bd.Y[sb])]++);
    }
  }
  plan.n_contrib += (size_t)cnt[nblk];
  plan.n_ccontrib += (size_t)ccnt[P];
  return true;
}

// Flattens the per-window builds into the device arrays: symmetric items first (they and the
// cross items are launched as two kernels with different register budgets).
inline void finish_plan(const std::vector<int>& build_win, const std::vector<plan_detail::Build>& builds, const std::vector<SRec>& recs, SchurPlan& plan) {
  std::vector<size_t> order;
  for (size_t b = 0; b < builds.size(); ++b) if (builds[b].sym) order.push_back(b);
  plan.n_sym = (int)order.size();
  for (size_t b = 0; b < builds.size(); ++b) if (!builds[b].sym) order.push_back(b);
  plan.items.reserve(order.size());
  for (size_t b : order) {
    const plan_detail::Build& bd = builds[b];
    SItem it;
    it.win = build_win[b]; it.rec_off = (int)plan.recs.size(); it.n_lm = bd.n_rec;
    it.shape = bd.nx | (bd.ny << 8) | ((bd.sym ? 1 : 0) << 16);
    plan.items.push_back(it);
    plan.recs.insert(plan.recs.end(), recs.begin() + bd.rec_off, recs.begin() + bd.rec_off + bd.n_rec);
    plan.pair_slot.insert(plan.pair_slot.end(), bd.pair_slot, bd.pair_slot + 64);
    plan.c_slot.insert(plan.c_slot.end(), bd.c_slot, bd.c_slot + 8);
    for (int s = 0; s < 8; ++s) { plan.pose_x.push_back(bd.X[s]); plan.pose_y.push_back(bd.Y[s]); }
  }
}

}  // namespace osh
