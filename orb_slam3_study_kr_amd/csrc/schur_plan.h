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
#include <vector>

namespace osh {

constexpr int kItemPoses = 8;     // row / column poses of one item
constexpr int kItemMaxLm = 64;    // landmarks per item (load balance; partial sums are per item)
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
inline void unpack_poses(const uint64_t in[2], std::vector<int>& out) {
  out.clear();
  for (int s = 0; s < kItemPoses; ++s) {
    const unsigned v = (unsigned)((in[s >> 2] >> (48 - 16 * (s & 3))) & 0xffff);
    if (v != 0xffff) out.push_back((int)v);
  }
}

struct Build {
  bool sym = true;
  std::vector<int> X, Y;
  std::vector<SRec> recs;
  int pair_slot[64];
  int c_slot[8];
};

inline std::vector<int> set_union(const std::vector<int>& a, const std::vector<int>& b) {
  std::vector<int> u;
  std::set_union(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(u));
  return u;
}

inline unsigned long long pack_slots(const std::vector<int>& S, const int* obs, int r0, int r1) {
  unsigned long long v = ~0ull;
  for (int r = r0; r < r1; ++r) {
    const int slot = (int)(std::lower_bound(S.begin(), S.end(), obs[r]) - S.begin());
    v &= ~(0xffull << (8 * slot));
    v |= (unsigned long long)(unsigned)r << (8 * slot);
  }
  return v;
}

}  // namespace plan_detail

// Adds the items of window `w` to `plan` (items of all windows are re-ordered by finish_plan).
//   P: optimisable poses; L: landmarks; lmo[L+1]: sorted-edge offsets; nfree[L]: optimisable-pose
//   edges of each landmark (they come first, poses ascending); epose: pose of every sorted edge.
// Returns false when a landmark has more optimisable observers than a record can index.
inline bool plan_window(int w, int P, int L, const int* lmo, const int* nfree, const int* epose,
                        std::vector<plan_detail::Build>& out_builds, SchurPlan& plan) {
  using namespace plan_detail;
  std::vector<Unit> units;
  units.reserve((size_t)L + L / 2);
  for (int j = 0; j < L; ++j) {
    const int k = nfree[j];
    if (k > 254 || P >= 0xffff) return false;
    const int* obs = epose + lmo[j];
    const int nparts = std::max(1, (k + kItemPoses - 1) / kItemPoses);
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
    plan.pair_blocks += (long long)k * (k + 1) / 2;
  }
  // Order by unit_less.  A window has few distinct keys (hundreds, against tens of thousands of units), so the units are
  // bucketed by key with a small open-addressing table, only the distinct keys are sorted, and the units of a key keep their
  // creation order (lm, a, b ascending), which is unit_less's tie-break.
  {
    size_t cap = 64;
    while (cap < 2 * units.size()) cap <<= 1;
    std::vector<int> table(cap, -1);          // slot -> group
    std::vector<int> rep, count, gid(units.size());   // group -> first unit with that key, number of units
    for (size_t i = 0; i < units.size(); ++i) {
      const Unit& u = units[i];
      uint64_t h = u.k[0] * 0x9e3779b97f4a7c15ull;
      for (int q = 1; q < 5; ++q) { h ^= u.k[q]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
      size_t slot = (size_t)h & (cap - 1);
      while (table[slot] >= 0 && !units[rep[table[slot]]].same_key(u)) slot = (slot + 1) & (cap - 1);
      if (table[slot] < 0) { table[slot] = (int)rep.size(); rep.push_back((int)i); count.push_back(0); }
      gid[i] = table[slot];
      count[gid[i]]++;
    }
    std::vector<int> gorder(rep.size());
    for (size_t g = 0; g < rep.size(); ++g) gorder[g] = (int)g;
    std::sort(gorder.begin(), gorder.end(), [&](int p, int q) { return unit_less(units[rep[p]], units[rep[q]]); });
    std::vector<size_t> start(rep.size());
    size_t acc = 0;
    for (int g : gorder) { start[g] = acc; acc += (size_t)count[g]; }
    std::vector<Unit> sorted(units.size());
    for (size_t i = 0; i < units.size(); ++i) sorted[start[gid[i]]++] = units[i];
    units.swap(sorted);
  }

  const size_t first_build = out_builds.size();
  std::vector<int> curX, curY;
  std::vector<const Unit*> cur;
  bool cur_sym = true;
  auto key_sets = [](const Unit& u, std::vector<int>& X, std::vector<int>& Y) {
    unpack_poses(&u.k[1], X);
    unpack_poses(&u.k[3], Y);
  };
  auto flush = [&]() {
    for (size_t base = 0; base < cur.size(); base += kItemMaxLm) {
      Build bd;
      bd.sym = cur_sym; bd.X = curX; bd.Y = curY;
      std::fill(bd.pair_slot, bd.pair_slot + 64, -1);
      std::fill(bd.c_slot, bd.c_slot + 8, -1);
      const size_t end = std::min(cur.size(), base + kItemMaxLm);
      bd.recs.reserve(end - base);
      const Unit* prev = nullptr;
      unsigned long long xs = 0, ys = 0;
      for (size_t x = base; x < end; ++x) {
        const Unit& u = *cur[x];
        const int k = nfree[u.lm];
        const int* obs = epose + lmo[u.lm];
        const int a0 = u.a * kItemPoses, a1 = std::min(k, a0 + kItemPoses);
        const int b0 = u.b * kItemPoses, b1 = std::min(k, b0 + kItemPoses);
        // units with the same key and part pair have the same slot assignment and mark the same live pairs
        const bool same = prev && prev->same_key(u) && prev->a == u.a && prev->b == u.b;
        if (!same) { xs = pack_slots(curX, obs, a0, a1); ys = pack_slots(curY, obs, b0, b1); }
        prev = &u;
        SRec r;
        r.lm = u.lm; r.e_first = lmo[u.lm];
        r.x_lo = (unsigned)xs; r.x_hi = (unsigned)(xs >> 32); r.y_lo = (unsigned)ys; r.y_hi = (unsigned)(ys >> 32);
        r.flags = ((u.a == 0 && u.b == 0) ? (1 | (std::min(k, kItemPoses) << 8)) : 0) | (a0 << 16) | ((a1 - a0) << 24);
        r.pad = lmo[u.lm + 1] - lmo[u.lm];
        bd.recs.push_back(r);
        // live pairs (marked with -2, numbered later)
        if (same) continue;
        for (int sa = 0; sa < kItemPoses; ++sa) {
          if (((xs >> (8 * sa)) & 0xff) == kAbsent) continue;
          if (cur_sym) bd.c_slot[sa] = -2;
          for (int sb = cur_sym ? sa : 0; sb < kItemPoses; ++sb)
            if (((ys >> (8 * sb)) & 0xff) != kAbsent) bd.pair_slot[sa * 8 + sb] = -2;
        }
      }
      const int tx = tiles_of((int)curX.size()), ty = tiles_of((int)curY.size());
      const long long tiles = cur_sym ? (long long)tx * (tx + 1) / 2 : (long long)tx * ty;
      plan.tile_steps += tiles * 6 * (long long)((end - base + 7) / 8);
      out_builds.push_back(std::move(bd));
    }
    cur.clear();
  };
  std::vector<int> gX, gY;
  size_t x = 0;
  while (x < units.size()) {
    size_t x1 = x + 1;
    while (x1 < units.size() && units[x1].same_key(units[x])) ++x1;
    key_sets(units[x], gX, gY);
    const bool g_sym = units[x].k[0] == 0;
    bool merged = false;
    if (!cur.empty() && cur_sym == g_sym && cur.size() < (size_t)kItemMaxLm) {
      std::vector<int> ux = set_union(curX, gX), uy = set_union(curY, gY);
      if ((int)ux.size() <= kItemPoses && (int)uy.size() <= kItemPoses &&
          tiles_of((int)ux.size()) == tiles_of((int)curX.size()) && tiles_of((int)ux.size()) == tiles_of((int)gX.size()) &&
          tiles_of((int)uy.size()) == tiles_of((int)curY.size()) && tiles_of((int)uy.size()) == tiles_of((int)gY.size())) {
        curX.swap(ux); curY.swap(uy);
        merged = true;
      }
    }
    if (!merged) {
      if (!cur.empty()) flush();
      curX = gX; curY = gY; cur_sym = g_sym;
    }
    for (size_t u = x; u < x1; ++u) cur.push_back(&units[u]);
    x = x1;
  }
  if (!cur.empty()) flush();

  // contribution slots: the contributions of one block of S are contiguous, in item order
  const int nblk = P * (P + 1) / 2;
  auto blk = [P](int i, int j) { return i * P - i * (i - 1) / 2 + (j - i); };
  std::vector<int> cnt((size_t)nblk + 1, 0), ccnt((size_t)P + 1, 0);
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
  for (int i = 0; i < P; ++i)
    for (int j = i; j < P; ++j) {
      const int k = blk(i, j);
      plan.rblk.push_back(RBlk{w, i | (j << 16), (int)(c0 + cnt[k]), cnt[k + 1] - cnt[k]});
    }
  for (int i = 0; i < P; ++i) plan.rblk.push_back(RBlk{w, i | (0xffff << 16), (int)(cc0 + ccnt[i]), ccnt[i + 1] - ccnt[i]});
  std::vector<int> fill(cnt.begin(), cnt.end() - 1), cfill(ccnt.begin(), ccnt.end() - 1);
  for (size_t b = first_build; b < out_builds.size(); ++b) {
    Build& bd = out_builds[b];
    for (int sa = 0; sa < 8; ++sa) {
      if (bd.c_slot[sa] == -2) bd.c_slot[sa] = (int)(cc0 + cfill[bd.X[sa]]++);
      for (int sb = 0; sb < 8; ++sb)
        if (bd.pair_slot[sa * 8 + sb] == -2) bd.pair_slot[sa * 8 + sb] = (int)(c0 + fill[blk(bd.X[sa], bd.Y[sb])]++);
    }
  }
  plan.n_contrib += (size_t)cnt[nblk];
  plan.n_ccontrib += (size_t)ccnt[P];
  return true;
}

// Flattens the per-window builds into the device arrays: symmetric items first (they and the
// cross items are launched as two kernels with different register budgets).
inline void finish_plan(const std::vector<int>& build_win, std::vector<plan_detail::Build>& builds, SchurPlan& plan) {
  std::vector<size_t> order;
  for (size_t b = 0; b < builds.size(); ++b) if (builds[b].sym) order.push_back(b);
  plan.n_sym = (int)order.size();
  for (size_t b = 0; b < builds.size(); ++b) if (!builds[b].sym) order.push_back(b);
  plan.items.reserve(order.size());
  for (size_t b : order) {
    const plan_detail::Build& bd = builds[b];
    SItem it;
    it.win = build_win[b]; it.rec_off = (int)plan.recs.size(); it.n_lm = (int)bd.recs.size();
    it.shape = (int)bd.X.size() | ((int)bd.Y.size() << 8) | ((bd.sym ? 1 : 0) << 16);
    plan.items.push_back(it);
    plan.recs.insert(plan.recs.end(), bd.recs.begin(), bd.recs.end());
    plan.pair_slot.insert(plan.pair_slot.end(), bd.pair_slot, bd.pair_slot + 64);
    plan.c_slot.insert(plan.c_slot.end(), bd.c_slot, bd.c_slot + 8);
    for (int s = 0; s < 8; ++s) {
      plan.pose_x.push_back(s < (int)bd.X.size() ? bd.X[s] : -1);
      plan.pose_y.push_back(s < (int)bd.Y.size() ? bd.Y[s] : -1);
    }
  }
}

}  // namespace osh
