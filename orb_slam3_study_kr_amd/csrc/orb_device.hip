// orb_device.hip -- 256-bit Hamming nearest / second-nearest search on MI355X (gfx950).
//
// Replaces the candidate loops of ORB_SLAM3::ORBmatcher::SearchByProjection
// (src/ORBmatcher.cc:84-120, 1743-1768, 1949-1964) and ORBmatcher::DescriptorDistance
// (src/ORBmatcher.cc:2058-2074).  Bit-exact contract: the reference scans the candidates left
// to right with a strict '<', i.e. it keeps the two smallest (distance, position) pairs in
// lexicographic order; a candidate at distance 256 can never win (bestDist starts at 256).
// Both are reproduced with one packed integer key  (distance << 22) | position  and min/max
// selection, which is associative and therefore order independent.
//
// Integer-VALU bound: per descriptor pair 8 v_xor_b32 + 8 v_bcnt_u32_b32 and
// 4 select ops; the 32-byte train rows are staged in LDS and read as wave-wide broadcasts.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace osh {

constexpr int kPosBits = 22;
constexpr unsigned kPosMask = (1u << kPosBits) - 1;
constexpr unsigned kKeyNone = 0xFFFFFFFFu;
constexpr int kQBlock = 256;     // queries per block (one per thread)
constexpr int kTrainTile = 1024; // train descriptors staged in LDS per tile (32 KiB)

struct OrbView {
  int n_pairs, n_query, n_train, n_split;
  const uint4* query;        // [n_pairs*n_query][2]
  const uint4* train;        // [n_pairs*n_train][2]
  const int* train_level;    // [n_pairs*n_train] or null
  const int* cand_off;       // [n_pairs*(n_query+1)] or null
  const int* cand_idx;
  const long long* pair_cand_base;
  unsigned* part_keys;       // [n_split][n_pairs*n_query][2] (brute force partials)
  // grid mode (osh_orb_upload_grid)
  const float2* train_xy; const float* train_uright; const unsigned char* train_skip;
  const int* cell_off;       // [n_pairs*(cols*rows+1)]
  const int* cell_idx;       // [n_pairs*n_train] train indices by cell, insertion order inside a cell
  const float* qwin; const int2* qlev; const float2* qur;
  float min_x, min_y, winv, hinv; int cols, rows;
  int* best_idx; int* best_dist; int* second_dist; int* best_level; int* second_level; int* second_idx;
};

// popcount(x) + acc in ONE instruction (v_bcnt_u32_b32 adds its second operand).  Written as `__builtin_popcount(x) + acc` the
// compiler re-associates the eight terms of a distance into separate counts and a tree of v_add3_u32: 3 extra lane-ops per pair
// (22.8 measured against the 16 of eight xor + eight chained counts).
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) {
  unsigned r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}
__device__ __forceinline__ unsigned hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
  unsigned d = __builtin_popcount(a0.x ^ b0.x);
  d = bcnt_acc(a0.y ^ b0.y, d);
  d = bcnt_acc(a0.z ^ b0.z, d);
  d = bcnt_acc(a0.w ^ b0.w, d);
  d = bcnt_acc(a1.x ^ b1.x, d);
  d = bcnt_acc(a1.y ^ b1.y, d);
  d = bcnt_acc(a1.z ^ b1.z, d);
  d = bcnt_acc(a1.w ^ b1.w, d);
  return d;
}

// keep the two smallest keys: with best <= second the new second is the MEDIAN of (best, second, key) -- one v_med3_u32 instead of
// a max and a min
__device__ __forceinline__ void top2_insert(unsigned key, unsigned& best, unsigned& second) {
  unsigned m;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(best), "v"(second), "v"(key));
  second = m;
  best = min(best, key);
}
__device__ __forceinline__ void top2_merge(unsigned b2, unsigned s2, unsigned& best, unsigned& second) {
  const unsigned nb = min(best, b2);
  const unsigned ns = min(max(best, b2), min(second, s2));
  best = nb; second = ns;
}

__device__ __forceinline__ void emit(const OrbView& v, size_t gq, size_t train_base, unsigned best, unsigned second,
                                     const int* cand /* null: position == train index */) {
  const unsigned bd = best >> kPosBits, sd = second >> kPosBits;
  int bi = -1, bl = -1, sl = -1, si_out = -1;
  int bdist = 256, sdist = 256;
  if (best != kKeyNone && bd < 256) {
    const int pos = (int)(best & kPosMask);
    bi = cand ? cand[pos] : pos;
    bdist = (int)bd;
    bl = v.train_level ? v.train_level[train_base + bi] : 0;
    if (second != kKeyNone && sd < 256) {
      const int pos2 = (int)(second & kPosMask);
      const int si = cand ? cand[pos2] : pos2;
      sdist = (int)sd;
      si_out = si;
      sl = v.train_level ? v.train_level[train_base + si] : 0;
    }
  }
  v.best_idx[gq] = bi; v.best_dist[gq] = bdist; v.second_dist[gq] = sdist;
  v.best_level[gq] = bl; v.second_level[gq] = sl; v.second_idx[gq] = si_out;
}

// Brute force: block = 256 queries of one pair x one slice of the train set.
// grid.x = n_pairs * ceil(n_query/256), grid.y = n_split.
__global__ __launch_bounds__(kQBlock) void k_orb_bruteforce(OrbView v) {
  __shared__ uint4 sh_train[kTrainTile * 2];
  const int qblocks = (v.n_query + kQBlock - 1) / kQBlock;
  const int pair = blockIdx.x / qblocks;
  const int qb = blockIdx.x - pair * qblocks;
  const int q = qb * kQBlock + threadIdx.x;
  const bool valid = q < v.n_query;
  const size_t gq = (size_t)pair * v.n_query + (valid ? q : 0);
  const uint4 a0 = v.query[gq * 2], a1 = v.query[gq * 2 + 1];
  // slice of the train set handled by this block
  const int per = (v.n_train + v.n_split - 1) / v.n_split;
  const int t_begin = blockIdx.y * per;
  const int t_end = min(v.n_train, t_begin + per);
  const uint4* tr = v.train + (size_t)pair * v.n_train * 2;
  unsigned best = kKeyNone, second = kKeyNone;
  for (int t0 = t_begin; t0 < t_end; t0 += kTrainTile) {
    const int nt = min(kTrainTile, t_end - t0);
    __syncthreads();
    for (int k = threadIdx.x; k < nt * 2; k += kQBlock) sh_train[k] = tr[(size_t)t0 * 2 + k];
    __syncthreads();
    int t = 0;
    for (; t + 4 <= nt; t += 4) {   // four train descriptors per step, unrolled by hand (the inline-asm counts keep `#pragma unroll` from applying)
      uint4 b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = sh_train[2 * t + u];   // wave-wide broadcast reads
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned d = hamming256(a0, a1, b[2 * u], b[2 * u + 1]);
        top2_insert((d << kPosBits) | (unsigned)(t0 + t + u), best, second);
      }
    }
    for (; t < nt; ++t) {
      const uint4 b0 = sh_train[2 * t], b1 = sh_train[2 * t + 1];
      const unsigned d = hamming256(a0, a1, b0, b1);
      top2_insert((d << kPosBits) | (unsigned)(t0 + t), best, second);
    }
  }
  if (!valid) return;
  if (v.n_split == 1) {
    emit(v, gq, (size_t)pair * v.n_train, best, second, nullptr);
  } else {
    const size_t nq_total = (size_t)v.n_pairs * v.n_query;
    unsigned* o = v.part_keys + ((size_t)blockIdx.y * nq_total + gq) * 2;
    o[0] = best; o[1] = second;
  }
}

__global__ void k_orb_merge(OrbView v) {
  const size_t nq_total = (size_t)v.n_pairs * v.n_query;
  const size_t gq = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gq >= nq_total) return;
  unsigned best = kKeyNone, second = kKeyNone;
  for (int s = 0; s < v.n_split; ++s) {
    const unsigned* o = v.part_keys + ((size_t)s * nq_total + gq) * 2;
    top2_merge(o[0], o[1], best, second);
  }
  const int pair = (int)(gq / v.n_query);
  emit(v, gq, (size_t)pair * v.n_train, best, second, nullptr);
}

// Windowed search: one wavefront per query, lanes stride the candidate list
// (Frame::GetFeaturesInArea order, src/Frame.cc:658-722); 64-lane butterfly top-2 merge.
__global__ __launch_bounds__(kQBlock) void k_orb_windowed(OrbView v) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t gq = (size_t)blockIdx.x * (kQBlock / 64) + wave;
  const size_t nq_total = (size_t)v.n_pairs * v.n_query;
  if (gq >= nq_total) return;
  const int pair = (int)(gq / v.n_query);
  const int q = (int)(gq - (size_t)pair * v.n_query);
  const int* off = v.cand_off + (size_t)pair * (v.n_query + 1);
  const int* cand = v.cand_idx + v.pair_cand_base[pair] + off[q];
  const int nc = off[q + 1] - off[q];
  const uint4 a0 = v.query[gq * 2], a1 = v.query[gq * 2 + 1];
  const uint4* tr = v.train + (size_t)pair * v.n_train * 2;
  unsigned best = kKeyNone, second = kKeyNone;
  for (int c = lane; c < nc; c += 64) {
    const int idx = cand[c];
    const uint4 b0 = tr[(size_t)idx * 2], b1 = tr[(size_t)idx * 2 + 1];
    const unsigned d = hamming256(a0, a1, b0, b1);
    top2_insert((d << kPosBits) | (unsigned)c, best, second);
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const unsigned ob = __shfl_xor(best, m, 64), os = __shfl_xor(second, m, 64);
    top2_merge(ob, os, best, second);
  }
  if (lane == 0) emit(v, gq, (size_t)pair * v.n_train, best, second, cand);
}

// Grid search: one wavefront per query, lane c walks cell c of the query's window (ix-major, then iy: the order of
// Frame::GetFeaturesInArea) and its entries in insertion order; position key = (cell number << 8) | entry, so equal
// distances resolve to the earliest candidate of the reference's scan.
__global__ __launch_bounds__(kQBlock) void k_orb_grid(OrbView v) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t gq = (size_t)blockIdx.x * (kQBlock / 64) + wave;
  const size_t nq_total = (size_t)v.n_pairs * v.n_query;
  if (gq >= nq_total) return;
  const int pair = (int)(gq / v.n_query);
  const size_t tb = (size_t)pair * v.n_train;
  const int* coff = v.cell_off + (size_t)pair * (v.cols * v.rows + 1);
  const int* cidx = v.cell_idx + tb;
  const float x = v.qwin[gq * 3], y = v.qwin[gq * 3 + 1], r = v.qwin[gq * 3 + 2];
  const int2 lev = v.qlev[gq];
  unsigned best = kKeyNone, second = kKeyNone;
  int c0x = 0, c0y = 0, ncx = 0, ncy = 0;
  if (r > 0.0f) {
    // src/Frame.cc:666-684: float arithmetic, left to right
    const int nMinCellX = max(0, (int)floorf((x - v.min_x - r) * v.winv));
    const int nMaxCellX = min(v.cols - 1, (int)ceilf((x - v.min_x + r) * v.winv));
    const int nMinCellY = max(0, (int)floorf((y - v.min_y - r) * v.hinv));
    const int nMaxCellY = min(v.rows - 1, (int)ceilf((y - v.min_y + r) * v.hinv));
    if (nMinCellX < v.cols && nMaxCellX >= 0 && nMinCellY < v.rows && nMaxCellY >= 0 && nMaxCellX >= nMinCellX && nMaxCellY >= nMinCellY) {
      c0x = nMinCellX; c0y = nMinCellY; ncx = nMaxCellX - nMinCellX + 1; ncy = nMaxCellY - nMinCellY + 1;
    }
  }
  const int ncell = ncx * ncy;
  const uint4 a0 = v.query[gq * 2], a1 = v.query[gq * 2 + 1];
  const uint4* tr = v.train + tb * 2;
  const bool check_ur = v.qur != nullptr && v.train_uright != nullptr;
  float q_ur = 0.f, q_tol = 0.f;
  if (check_ur) { const float2 u = v.qur[gq]; q_ur = u.x; q_tol = u.y; }
  for (int c = lane; c < ncell; c += 64) {
    const int cx = c / ncy, cy = c - cx * ncy;
    const int cell = (c0x + cx) * v.rows + (c0y + cy);
    const int k0 = coff[cell], k1 = coff[cell + 1];
    for (int k = k0; k < k1; ++k) {
      const int idx = cidx[k];
      if (v.train_skip && v.train_skip[tb + idx]) continue;
      const int oct = v.train_level ? v.train_level[tb + idx] : 0;
      if (oct < lev.x) continue;
      if (lev.y >= 0 && oct > lev.y) continue;
      const float2 p = v.train_xy[tb + idx];
      const float distx = p.x - x, disty = p.y - y;
      if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;
      if (check_ur) {
        const float tur = v.train_uright[tb + idx];
        if (tur > 0.f) { const float er = fabsf(q_ur - tur); if (er > q_tol) continue; }
      }
      const uint4 b0 = tr[(size_t)idx * 2], b1 = tr[(size_t)idx * 2 + 1];
      const unsigned d = hamming256(a0, a1, b0, b1);
      top2_insert((d << kPosBits) | ((unsigned)c << 8) | (unsigned)(k - k0), best, second);
    }
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const unsigned ob = __shfl_xor(best, m, 64), os = __shfl_xor(second, m, 64);
    top2_merge(ob, os, best, second);
  }
  if (lane == 0) {
    auto decode = [&](unsigned key) {
      const unsigned pos = key & kPosMask;
      const int c = (int)(pos >> 8), k = (int)(pos & 0xff);
      const int cx = c / ncy, cy = c - cx * ncy;
      return cidx[coff[(c0x + cx) * v.rows + (c0y + cy)] + k];
    };
    const unsigned bd = best >> kPosBits, sd = second >> kPosBits;
    int bi = -1, bl = -1, sl = -1, si = -1, bdist = 256, sdist = 256;
    if (best != kKeyNone && bd < 256) {
      bi = decode(best); bdist = (int)bd;
      bl = v.train_level ? v.train_level[tb + bi] : 0;
      if (second != kKeyNone && sd < 256) { si = decode(second); sdist = (int)sd; sl = v.train_level ? v.train_level[tb + si] : 0; }
    }
    v.best_idx[gq] = bi; v.best_dist[gq] = bdist; v.second_dist[gq] = sdist;
    v.best_level[gq] = bl; v.second_level[gq] = sl; v.second_idx[gq] = si;
  }
}

// --------------------------------------------------------------------------------------------
// Sequential slot occupancy of SearchByProjection(Frame&, vector<MapPoint*>&) resolved on the device
// (src/ORBmatcher.cc:84-139).  The reference walks the map points in order; a keypoint slot that already holds a map point
// with Observations() > 0 is skipped (:88-90), and an accepted match stores its map point into the slot (:131-136).  So the
// result of query q is the search over the candidates minus the slots claimed by accepted, blocking queries q' < q.
// Fixed-point rounds: with `owner[s]` = lowest blocking claimant of slot s under the claims of the previous round, every
// query is searched again with the slots { s : owner[s] < q } closed, and claims anew.  Round k makes the claims of the
// first k queries final (query q only depends on queries before it), so the rounds reach a state that repeats, and that
// state is the sequential result; dependency chains are short, a handful of rounds in practice.  A query whose unrestricted
// best and second-best slots are both open keeps its unrestricted result (closing slots only removes candidates).
// --------------------------------------------------------------------------------------------
struct ResolveView {
  int* claim;                    // [n_pairs*n_query] slot claimed in the last round, -1 none
  int* owner[2];                 // [n_pairs*n_train] lowest blocking claimant, INT_MAX none (double buffered by round parity)
  const unsigned char* pre;      // [n_pairs*n_train] slot occupied at call entry, or null
  const unsigned char* blocks;   // [n_pairs*n_query] 1: an accepted match of this query closes its slot, or null (all do)
  int* cur[5];                   // best_idx, best_dist, second_dist, best_level, second_level under the current occupancy
  int* state;                    // [0] claims changed this round, [1] done, [2] rounds run
  float nn_ratio; int th_high;
};

__global__ void k_orb_claim(OrbView v, ResolveView r, int round) {
  if (r.state[1]) return;
  const size_t nq_total = (size_t)v.n_pairs * v.n_query;
  const size_t gq = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gq >= nq_total) return;
  const int pair = (int)(gq / v.n_query), q = (int)(gq - (size_t)pair * v.n_query);
  const int bi = round ? r.cur[0][gq] : v.best_idx[gq], bd = round ? r.cur[1][gq] : v.best_dist[gq];
  const int sd = round ? r.cur[2][gq] : v.second_dist[gq];
  const int bl = round ? r.cur[3][gq] : v.best_level[gq], sl = round ? r.cur[4][gq] : v.second_level[gq];
  // src/ORBmatcher.cc:123-128: TH_HIGH, then the ratio test between two candidates of the same pyramid level (float compare)
  const bool accept = bi >= 0 && bd <= r.th_high && !(bl == sl && (float)bd > r.nn_ratio * (float)sd);
  const int c = accept ? bi : -1;
  if (c != r.claim[gq]) { atomicAdd(&r.state[0], 1); r.claim[gq] = c; }
  if (c >= 0 && (!r.blocks || r.blocks[gq])) atomicMin(&r.owner[round & 1][(size_t)pair * v.n_train + c], q);
}

__global__ void k_orb_round_end(OrbView v, ResolveView r, int round) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)v.n_pairs * v.n_train) r.owner[(round + 1) & 1][i] = INT_MAX;
  if (i == 0 && !r.state[1]) {
    r.state[2] = round + 1;
    if (r.state[0] == 0) r.state[1] = 1;
    r.state[0] = 0;
  }
}

// One wavefront per query: the search of k_orb_bruteforce / k_orb_windowed / k_orb_grid with the closed slots removed.
template <int MODE>  // 0 brute force, 1 candidate lists, 2 grid
__global__ __launch_bounds__(kQBlock) void k_orb_research(OrbView v, ResolveView r, int round) {
  if (r.state[1]) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t gq = (size_t)blockIdx.x * (kQBlock / 64) + wave;
  const size_t nq_total = (size_t)v.n_pairs * v.n_query;
  if (gq >= nq_total) return;
  const int pair = (int)(gq / v.n_query), q = (int)(gq - (size_t)pair * v.n_query);
  const size_t tb = (size_t)pair * v.n_train;
  const int* owner = r.owner[round & 1] + tb;
  const unsigned char* pre = r.pre ? r.pre + tb : nullptr;
  auto closed = [&](int t) { return (pre && pre[t]) || owner[t] < q; };
  const int b0i = v.best_idx[gq], s0i = v.second_idx[gq];
  const bool contested = (b0i >= 0 && closed(b0i)) || (s0i >= 0 && closed(s0i));
  if (!contested) {
    if (lane == 0) {
      r.cur[0][gq] = b0i; r.cur[1][gq] = v.best_dist[gq]; r.cur[2][gq] = v.second_dist[gq];
      r.cur[3][gq] = v.best_level[gq]; r.cur[4][gq] = v.second_level[gq];
    }
    return;
  }
  const uint4 a0 = v.query[gq * 2], a1 = v.query[gq * 2 + 1];
  const uint4* tr = v.train + tb * 2;
  unsigned best = kKeyNone, second = kKeyNone;
  int bi = -1, si = -1;
  if (MODE == 0) {
    for (int t = lane; t < v.n_train; t += 64) {
      if (closed(t)) continue;
      const unsigned d = hamming256(a0, a1, tr[(size_t)t * 2], tr[(size_t)t * 2 + 1]);
      top2_insert((d << kPosBits) | (unsigned)t, best, second);
    }
  } else if (MODE == 1) {
    const int* off = v.cand_off + (size_t)pair * (v.n_query + 1);
    const int* cand = v.cand_idx + v.pair_cand_base[pair] + off[q];
    const int nc = off[q + 1] - off[q];
    for (int c = lane; c < nc; c += 64) {
      const int idx = cand[c];
      if (closed(idx)) continue;
      const unsigned d = hamming256(a0, a1, tr[(size_t)idx * 2], tr[(size_t)idx * 2 + 1]);
      top2_insert((d << kPosBits) | (unsigned)c, best, second);
    }
  } else {
    // the candidate walk of k_orb_grid
    const int* coff = v.cell_off + (size_t)pair * (v.cols * v.rows + 1);
    const int* cidx = v.cell_idx + tb;
    const float x = v.qwin[gq * 3], y = v.qwin[gq * 3 + 1], rad = v.qwin[gq * 3 + 2];
    const int2 lev = v.qlev[gq];
    int c0x = 0, c0y = 0, ncx = 0, ncy = 0;
    if (rad > 0.0f) {
      const int nMinCellX = max(0, (int)floorf((x - v.min_x - rad) * v.winv));
      const int nMaxCellX = min(v.cols - 1, (int)ceilf((x - v.min_x + rad) * v.winv));
      const int nMinCellY = max(0, (int)floorf((y - v.min_y - rad) * v.hinv));
      const int nMaxCellY = min(v.rows - 1, (int)ceilf((y - v.min_y + rad) * v.hinv));
      if (nMinCellX < v.cols && nMaxCellX >= 0 && nMinCellY < v.rows && nMaxCellY >= 0 && nMaxCellX >= nMinCellX && nMaxCellY >= nMinCellY) {
        c0x = nMinCellX; c0y = nMinCellY; ncx = nMaxCellX - nMinCellX + 1; ncy = nMaxCellY - nMinCellY + 1;
      }
    }
    const bool check_ur = v.qur != nullptr && v.train_uright != nullptr;
    float q_ur = 0.f, q_tol = 0.f;
    if (check_ur) { const float2 u = v.qur[gq]; q_ur = u.x; q_tol = u.y; }
    for (int c = lane; c < ncx * ncy; c += 64) {
      const int cx = c / ncy, cy = c - cx * ncy;
      const int cell = (c0x + cx) * v.rows + (c0y + cy);
      const int k0 = coff[cell], k1 = coff[cell + 1];
      for (int k = k0; k < k1; ++k) {
        const int idx = cidx[k];
        if ((v.train_skip && v.train_skip[tb + idx]) || closed(idx)) continue;
        const int oct = v.train_level ? v.train_level[tb + idx] : 0;
        if (oct < lev.x || (lev.y >= 0 && oct > lev.y)) continue;
        const float2 p = v.train_xy[tb + idx];
        if (!(fabsf(p.x - x) < rad && fabsf(p.y - y) < rad)) continue;
        if (check_ur) { const float tur = v.train_uright[tb + idx]; if (tur > 0.f && fabsf(q_ur - tur) > q_tol) continue; }
        const unsigned d = hamming256(a0, a1, tr[(size_t)idx * 2], tr[(size_t)idx * 2 + 1]);
        top2_insert((d << kPosBits) | ((unsigned)c << 8) | (unsigned)(k - k0), best, second);
      }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) top2_merge(__shfl_xor(best, m, 64), __shfl_xor(second, m, 64), best, second);
    if (lane == 0) {
      auto decode = [&](unsigned key) {
        const unsigned pos = key & kPosMask;
        const int c = (int)(pos >> 8), k = (int)(pos & 0xff);
        const int cx = c / ncy, cy = c - cx * ncy;
        return cidx[coff[(c0x + cx) * v.rows + (c0y + cy)] + k];
      };
      if (best != kKeyNone && (best >> kPosBits) < 256) {
        bi = decode(best);
        if (second != kKeyNone && (second >> kPosBits) < 256) si = decode(second);
      }
    }
  }
  if (MODE != 2) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) top2_merge(__shfl_xor(best, m, 64), __shfl_xor(second, m, 64), best, second);
    if (lane == 0 && best != kKeyNone && (best >> kPosBits) < 256) {
      const int* cand = MODE == 1 ? v.cand_idx + v.pair_cand_base[pair] + v.cand_off[(size_t)pair * (v.n_query + 1) + q] : nullptr;
      const int pos = (int)(best & kPosMask);
      bi = cand ? cand[pos] : pos;
      if (second != kKeyNone && (second >> kPosBits) < 256) { const int p2 = (int)(second & kPosMask); si = cand ? cand[p2] : p2; }
    }
  }
  if (lane == 0) {
    r.cur[0][gq] = bi;
    r.cur[1][gq] = bi >= 0 ? (int)(best >> kPosBits) : 256;
    r.cur[2][gq] = si >= 0 ? (int)(second >> kPosBits) : 256;
    r.cur[3][gq] = bi >= 0 ? (v.train_level ? v.train_level[tb + bi] : 0) : -1;
    r.cur[4][gq] = si >= 0 ? (v.train_level ? v.train_level[tb + si] : 0) : -1;
  }
}

// One block per pair: the slot table by atomicMax (few writers per slot), the match count by a block reduction (a counter
// shared by the 2000 queries of a pair made this kernel the slowest of the resolution: 0.44 ms against 0.06 ms per search round).
__global__ __launch_bounds__(256) void k_orb_assign(OrbView v, ResolveView r, int* assignment, int* n_matches) {
  __shared__ int sh_cnt[4];
  const int pair = blockIdx.x;
  int cnt = 0;
  for (int q = threadIdx.x; q < v.n_query; q += 256) {
    const int c = r.claim[(size_t)pair * v.n_query + q];
    if (c < 0) continue;
    // several non-blocking claimants may store into one slot in turn (:131-136): the last writer stays
    atomicMax(&assignment[(size_t)pair * v.n_train + c], q);
    ++cnt;
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) cnt += __shfl_xor(cnt, m, 64);
  if ((threadIdx.x & 63) == 0) sh_cnt[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) n_matches[pair] = sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
}

__global__ void k_orb_distance_matrix(int n, int m, const uint4* a, const uint4* b, int* out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n * m) return;
  const int i = (int)(idx / m), j = (int)(idx - (size_t)i * m);
  out[idx] = (int)hamming256(a[2 * (size_t)i], a[2 * (size_t)i + 1], b[2 * (size_t)j], b[2 * (size_t)j + 1]);
}

}  // namespace osh

using namespace osh;

// --------------------------------------------------------------------------------------------
// k_frustum: Frame::isInFrustum (src/Frame.cc:513-587, Nleft == -1 branch) for a batch of map
// points, one point per thread.  Float32 throughout like the reference (mRcw, mtcw, mOw and the
// MapPoint getters are float), no fused multiply-add, and the three-term sums in the order of
// Eigen's fixed-size reduction, a0 + (a1 + a2) (recalled, not readable here).
// in : 8 floats per point  (world position, normal, mfMinDistance, mfMaxDistance)
// out: 8 words per point   (stage, u, v, u - bf/z, |Pc|, viewCos, level, 0); stage 0: rejected before
//      the projection was stored, 1: mTrackProjX/Y stored then rejected, 2: in view.
// --------------------------------------------------------------------------------------------
struct FrustumFrame { float R[9], t[3], O[3], fx, fy, cx, cy, bf, min_x, max_x, min_y, max_y, log_sf, cos_limit; int levels; int fisheye; float kb[4]; };

// plain operators with contraction switched off per function (HIP's __fmul_rn / __fadd_rn are inline functions compiled with
// the default contract flag: after inlining their results still fuse into FMAs)
__device__ __forceinline__ float sum3(float a0, float a1, float a2) {
#pragma clang fp contract(off)
  return a0 + (a1 + a2);
}
// correctly rounded float32 square root (v_sqrt_f32 alone is good to 1 ulp): the FP64 root rounds to the same float as the
// exact one because 53 >= 2 * 24 + 2
__device__ __forceinline__ float sqrt_rn(float x) { return (float)sqrt((double)x); }

__global__ __launch_bounds__(256) void k_frustum(FrustumFrame f, int n, const float4* __restrict__ in, float4* __restrict__ out) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 a = in[2 * i], b = in[2 * i + 1];
  const float P[3] = {a.x, a.y, a.z}, Pn[3] = {a.w, b.x, b.y};
  // MapPoint::GetMinDistanceInvariance / GetMaxDistanceInvariance (src/MapPoint.cc:502-512)
  const float min_d = (0.8f * b.z), max_d = (1.2f * b.w);
  float Pc[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    Pc[r] = sum3(f.R[3 * r] * P[0], f.R[3 * r + 1] * P[1], f.R[3 * r + 2] * P[2]) + f.t[r];
  const float pc_dist = sqrt_rn(sum3((Pc[0] * Pc[0]), (Pc[1] * Pc[1]), (Pc[2] * Pc[2])));
  int stage = 0, level = -1;
  float u = -1.f, v = -1.f, ur = 0.f, vcos = 0.f, vcos_out = 0.f;
  if (!(Pc[2] < 0.0f)) {
    const float invz = (1.0f / Pc[2]);
    // Pinhole::project(Vector3f): fx * x / z + cx  (src/CameraModels/Pinhole.cpp:43-49)
    float pu = f.fx * Pc[0] / Pc[2] + f.cx;
    float pv = f.fy * Pc[1] / Pc[2] + f.cy;
    if (f.fisheye) {
      // KannalaBrandt8::project(Vector3f) (src/CameraModels/KannalaBrandt8.cpp:66-84): float32 throughout; atan2f / cosf / sinf
      // taken as their correctly rounded values (FP64 function rounded once) so the result does not depend on the libm
      const float x2y2 = Pc[0] * Pc[0] + Pc[1] * Pc[1];
      const float theta = (float)atan2((double)sqrt_rn(x2y2), (double)Pc[2]);
      const float psi = (float)atan2((double)Pc[1], (double)Pc[0]);
      const float t2 = theta * theta, t3 = theta * t2, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
      const float rr = theta + f.kb[0] * t3 + f.kb[1] * t5 + f.kb[2] * t7 + f.kb[3] * t9;
      pu = f.fx * rr * (float)cos((double)psi) + f.cx;
      pv = f.fy * rr * (float)sin((double)psi) + f.cy;
    }
    if (!(pu < f.min_x || pu > f.max_x) && !(pv < f.min_y || pv > f.max_y)) {
      stage = 1; u = pu; v = pv;
      const float PO[3] = {(P[0] - f.O[0]), (P[1] - f.O[1]), (P[2] - f.O[2])};
      const float dist = sqrt_rn(sum3((PO[0] * PO[0]), (PO[1] * PO[1]), (PO[2] * PO[2])));
      if (!(dist < min_d || dist > max_d)) {
        vcos = sum3(PO[0] * Pn[0], PO[1] * Pn[1], PO[2] * Pn[2]) / dist;
        if (!(vcos < f.cos_limit)) {
          // MapPoint::PredictScale (src/MapPoint.cc:531-546)
          const float ratio = (b.w / dist);
          // logf as the correctly rounded value (FP64 log rounded once): libm-independent, so the level of a point that sits
          // exactly on a boundary (ratio == scaleFactor^k) is reproducible; glibc's logf agrees except for rare 1-ulp cases
          const float lg = (float)log((double)ratio);
          int ns = (int)ceilf(lg / f.log_sf);
          if (ns < 0) ns = 0; else if (ns >= f.levels) ns = f.levels - 1;
          level = ns; stage = 2;
          ur = pu - f.bf * invz;
          vcos_out = vcos;
        }
      }
    }
  }
  out[2 * i] = make_float4(__int_as_float(stage), u, v, ur);
  out[2 * i + 1] = make_float4(pc_dist, vcos_out, __int_as_float(level), 0.f);
}

// Distance of every (query, candidate) entry of the uploaded candidate lists: out[pair_cand_base[p] + e] for entry e of pair p.
// For callers whose choice among the candidates needs a per-pair test the search kernels cannot make (the epipolar constraint of
// ORBmatcher::SearchForTriangulation is a virtual call on the reference's camera object): the host walks only the entries
// within its distance threshold.
__global__ __launch_bounds__(256) void k_orb_list_dist(OrbView v, int* __restrict__ out) {
  const int p = blockIdx.y;
  const int* off = v.cand_off + (size_t)p * (v.n_query + 1);
  const int total = off[v.n_query];
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  int lo = 0, hi = v.n_query;            // the query whose list holds entry e: largest q with off[q] <= e
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (off[mid] <= e) lo = mid; else hi = mid; }
  const long long base = v.pair_cand_base[p];
  const int t = v.cand_idx[base + e];
  const uint4* q = v.query + ((size_t)p * v.n_query + lo) * 2;
  const uint4* tr = v.train + ((size_t)p * v.n_train + t) * 2;
  out[base + e] = (int)hamming256(q[0], q[1], tr[0], tr[1]);
}

struct osh_orb_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  KernelTimer timer;
  DevBuf d_query, d_train, d_level, d_off, d_idx, d_base, d_part, d_out[6], d_a, d_b, d_dm;
  DevBuf d_txy, d_tur, d_tskip, d_coff, d_cidx, d_qwin, d_qlev, d_qur;
  DevBuf d_fin, d_fout, d_ldist;
  long long list_total = 0;
  int list_longest_pair = 0;
  DevBuf d_claim, d_owner[2], d_pre, d_blocks, d_cur[5], d_state, d_assign, d_nmatch;   // osh_orb_match_local_points
  std::vector<float> h_fin, h_fout;
  OrbView v{};
  bool uploaded = false, matched = false, windowed = false, grid = false;
};

#define OSH_TRY(expr) do { int _rc = (expr); if (_rc != OSH_OK) return _rc; } while (0)

extern "C" int osh_orb_create(int device, osh_orb_ctx** out) {
  if (!out) { set_error("osh_orb_create: out is NULL"); return OSH_ERR_INVALID; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible"); return OSH_ERR_NO_DEVICE; }
  if (device < 0 || device >= n) { set_error("device %d out of range (have %d)", device, n); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(device));
  osh_orb_ctx* c = new osh_orb_ctx();
  c->device = device;
  OSH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  *out = c;
  return OSH_OK;
}

extern "C" void osh_orb_destroy(osh_orb_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  DevBuf* bufs[] = {&c->d_query, &c->d_train, &c->d_level, &c->d_off, &c->d_idx, &c->d_base, &c->d_part,
                    &c->d_out[0], &c->d_out[1], &c->d_out[2], &c->d_out[3], &c->d_out[4], &c->d_out[5], &c->d_a, &c->d_b, &c->d_dm,
                    &c->d_txy, &c->d_tur, &c->d_tskip, &c->d_coff, &c->d_cidx, &c->d_qwin, &c->d_qlev, &c->d_qur,
                    &c->d_claim, &c->d_owner[0], &c->d_owner[1], &c->d_pre, &c->d_blocks, &c->d_cur[0], &c->d_cur[1], &c->d_cur[2], &c->d_cur[3],
                    &c->d_cur[4], &c->d_state, &c->d_assign, &c->d_nmatch, &c->d_fin, &c->d_fout};
  for (DevBuf* b : bufs) b->release();
  c->timer.destroy();
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int osh_orb_upload(osh_orb_ctx* c, const osh_orb_batch* b) {
  if (!c || !b || b->n_pairs <= 0 || b->n_query < 0 || b->n_train < 0 || !b->query_desc || !b->train_desc) {
    set_error("osh_orb_upload: bad arguments");
    return OSH_ERR_INVALID;
  }
  if (b->n_train > (int)kPosMask) { set_error("n_train exceeds %u", kPosMask); return OSH_ERR_UNSUPPORTED; }
  if (b->cand_off && (!b->cand_idx || !b->pair_cand_base)) { set_error("cand_off without cand_idx/pair_cand_base"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  OSH_HIP(hipStreamSynchronize(s));
  const size_t nq = (size_t)b->n_pairs * b->n_query, nt = (size_t)b->n_pairs * b->n_train;
  OSH_TRY(c->d_query.reserve(std::max<size_t>(nq * 32, 32)));
  OSH_TRY(c->d_train.reserve(std::max<size_t>(nt * 32, 32)));
  if (nq) OSH_HIP(hipMemcpyAsync(c->d_query.p, b->query_desc, nq * 32, hipMemcpyHostToDevice, s));
  if (nt) OSH_HIP(hipMemcpyAsync(c->d_train.p, b->train_desc, nt * 32, hipMemcpyHostToDevice, s));
  OrbView& v = c->v;
  v = OrbView{};
  v.n_pairs = b->n_pairs; v.n_query = b->n_query; v.n_train = b->n_train;
  v.query = c->d_query.as<uint4>(); v.train = c->d_train.as<uint4>();
  if (b->train_level) {
    OSH_TRY(c->d_level.reserve(std::max<size_t>(nt * 4, 4)));
    if (nt) OSH_HIP(hipMemcpyAsync(c->d_level.p, b->train_level, nt * 4, hipMemcpyHostToDevice, s));
    v.train_level = c->d_level.as<int>();
  }
  c->windowed = b->cand_off != nullptr;
  c->grid = false;
  if (c->windowed) {
    const size_t noff = (size_t)b->n_pairs * (b->n_query + 1);
    // validate candidate lists on the host: an out-of-range index would fault the GPU
    long long total = 0;
    for (int p = 0; p < b->n_pairs; ++p) {
      const int32_t* off = b->cand_off + (size_t)p * (b->n_query + 1);
      if (off[0] != 0) { set_error("pair %d: cand_off[0] != 0", p); return OSH_ERR_INVALID; }
      for (int q = 0; q < b->n_query; ++q)
        if (off[q + 1] < off[q]) { set_error("pair %d: cand_off not monotone at %d", p, q); return OSH_ERR_INVALID; }
      if (b->pair_cand_base[p] != total) { set_error("pair %d: pair_cand_base must be the running total of list lengths", p); return OSH_ERR_INVALID; }
      const int32_t* idx = b->cand_idx + total;
      for (int k = 0; k < off[b->n_query]; ++k)
        if (idx[k] < 0 || idx[k] >= b->n_train) { set_error("pair %d: candidate index %d out of range", p, idx[k]); return OSH_ERR_INVALID; }
      if (off[b->n_query] > 0) {
        int longest = 0;
        for (int q = 0; q < b->n_query; ++q) longest = std::max(longest, off[q + 1] - off[q]);
        if ((unsigned)longest > kPosMask) { set_error("candidate list too long"); return OSH_ERR_UNSUPPORTED; }
      }
      total += off[b->n_query];
    }
    OSH_TRY(c->d_off.reserve(noff * 4));
    OSH_TRY(c->d_idx.reserve(std::max<size_t>((size_t)total * 4, 4)));
    OSH_TRY(c->d_base.reserve((size_t)b->n_pairs * 8));
    OSH_HIP(hipMemcpyAsync(c->d_off.p, b->cand_off, noff * 4, hipMemcpyHostToDevice, s));
    if (total) OSH_HIP(hipMemcpyAsync(c->d_idx.p, b->cand_idx, (size_t)total * 4, hipMemcpyHostToDevice, s));
    OSH_HIP(hipMemcpyAsync(c->d_base.p, b->pair_cand_base, (size_t)b->n_pairs * 8, hipMemcpyHostToDevice, s));
    v.cand_off = c->d_off.as<int>(); v.cand_idx = c->d_idx.as<int>(); v.pair_cand_base = c->d_base.as<long long>();
    v.n_split = 1;
    c->list_total = total;
    c->list_longest_pair = 0;
    for (int p = 0; p < b->n_pairs; ++p) c->list_longest_pair = std::max(c->list_longest_pair, (int)b->cand_off[(size_t)p * (b->n_query + 1) + b->n_query]);
  } else {
    const int qblocks = (b->n_query + kQBlock - 1) / kQBlock;
    const long blocks = (long)b->n_pairs * std::max(qblocks, 1);
    int split = (int)std::min<long>(8, std::max<long>(1, (1024 + blocks - 1) / blocks));
    split = std::min(split, std::max(1, (b->n_train + 255) / 256));
    v.n_split = split;
    if (split > 1) OSH_TRY(c->d_part.reserve((size_t)split * std::max<size_t>(nq, 1) * 8));
    v.part_keys = c->d_part.as<unsigned>();
  }
  for (int k = 0; k < 6; ++k) OSH_TRY(c->d_out[k].reserve(std::max<size_t>(nq * 4, 4)));
  v.best_idx = c->d_out[0].as<int>(); v.best_dist = c->d_out[1].as<int>(); v.second_dist = c->d_out[2].as<int>();
  v.best_level = c->d_out[3].as<int>(); v.second_level = c->d_out[4].as<int>(); v.second_idx = c->d_out[5].as<int>();
  OSH_HIP(hipStreamSynchronize(s));
  c->uploaded = true; c->matched = false;
  return OSH_OK;
}

extern "C" int osh_orb_upload_grid(osh_orb_ctx* c, const osh_orb_batch* b, const osh_orb_grid* g) {
  if (!g || !b || b->cand_off) { set_error("osh_orb_upload_grid: needs a grid and a batch without candidate lists"); return OSH_ERR_INVALID; }
  if (!g->train_xy || !g->query_window || !g->query_levels || g->cols <= 0 || g->rows <= 0 || (long)g->cols * g->rows > (1 << 14)) {
    set_error("osh_orb_upload_grid: bad grid description"); return OSH_ERR_INVALID;
  }
  OSH_TRY(osh_orb_upload(c, b));
  hipStream_t s = c->stream;
  const size_t nq = (size_t)b->n_pairs * b->n_query, nt = (size_t)b->n_pairs * b->n_train;
  const int ncell = g->cols * g->rows;
  // Frame::AssignFeaturesToGrid (src/Frame.cc:397-417): counting sort by cell, keypoint order inside a cell
  std::vector<int> coff((size_t)b->n_pairs * (ncell + 1), 0), cidx(std::max<size_t>(nt, 1), 0), cell_of(std::max<size_t>(b->n_train, 1));
  for (int p = 0; p < b->n_pairs; ++p) {
    int* off = &coff[(size_t)p * (ncell + 1)];
    const float* xy = g->train_xy + (size_t)p * b->n_train * 2;
    for (int i = 0; i < b->n_train; ++i) {
      const int posX = (int)std::round((xy[2 * i] - g->min_x) * g->cell_w_inv);       // PosInGrid, src/Frame.cc:726-736
      const int posY = (int)std::round((xy[2 * i + 1] - g->min_y) * g->cell_h_inv);
      cell_of[i] = (posX < 0 || posX >= g->cols || posY < 0 || posY >= g->rows) ? -1 : posX * g->rows + posY;
      if (cell_of[i] >= 0) off[cell_of[i] + 1]++;
    }
    for (int k = 0; k < ncell; ++k) {
      if (off[k + 1] > 255) { set_error("pair %d: more than 255 keypoints in one grid cell", p); return OSH_ERR_UNSUPPORTED; }
      off[k + 1] += off[k];
    }
    std::vector<int> fill(off, off + ncell);
    int* ci = &cidx[(size_t)p * b->n_train];
    for (int i = 0; i < b->n_train; ++i) if (cell_of[i] >= 0) ci[fill[cell_of[i]]++] = i;
  }
  auto up = [&](DevBuf& d, const void* src, size_t bytes) -> int {
    OSH_TRY(d.reserve(std::max<size_t>(bytes, 8)));
    if (bytes) OSH_HIP(hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, s));
    return OSH_OK;
  };
  OSH_TRY(up(c->d_coff, coff.data(), coff.size() * 4)); OSH_TRY(up(c->d_cidx, cidx.data(), nt * 4));
  OSH_TRY(up(c->d_txy, g->train_xy, nt * 8));
  OSH_TRY(up(c->d_qwin, g->query_window, nq * 12)); OSH_TRY(up(c->d_qlev, g->query_levels, nq * 8));
  OrbView& v = c->v;
  v.train_xy = c->d_txy.as<float2>(); v.cell_off = c->d_coff.as<int>(); v.cell_idx = c->d_cidx.as<int>();
  v.qwin = c->d_qwin.as<float>(); v.qlev = c->d_qlev.as<int2>();
  v.train_uright = nullptr; v.train_skip = nullptr; v.qur = nullptr;
  if (g->train_uright) { OSH_TRY(up(c->d_tur, g->train_uright, nt * 4)); v.train_uright = c->d_tur.as<float>(); }
  if (g->train_skip) { OSH_TRY(up(c->d_tskip, g->train_skip, nt)); v.train_skip = c->d_tskip.as<unsigned char>(); }
  if (g->query_uright) { OSH_TRY(up(c->d_qur, g->query_uright, nq * 8)); v.qur = c->d_qur.as<float2>(); }
  v.min_x = g->min_x; v.min_y = g->min_y; v.winv = g->cell_w_inv; v.hinv = g->cell_h_inv; v.cols = g->cols; v.rows = g->rows;
  OSH_HIP(hipStreamSynchronize(s));
  c->grid = true;
  return OSH_OK;
}

extern "C" int osh_orb_match(osh_orb_ctx* c) {
  if (!c || !c->uploaded) { set_error("osh_orb_match: nothing uploaded"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const OrbView& v = c->v;
  const size_t nq = (size_t)v.n_pairs * v.n_query;
  if (nq == 0) { c->matched = true; return OSH_OK; }
  if (c->timer.enabled) OSH_TRY(c->timer.init());
  const bool t = c->timer.begin(0, s);
  if (c->grid) {
    const unsigned grid = (unsigned)((nq + (kQBlock / 64) - 1) / (kQBlock / 64));
    hipLaunchKernelGGL(k_orb_grid, dim3(grid), dim3(kQBlock), 0, s, v);
  } else if (c->windowed) {
    const unsigned grid = (unsigned)((nq + (kQBlock / 64) - 1) / (kQBlock / 64));
    hipLaunchKernelGGL(k_orb_windowed, dim3(grid), dim3(kQBlock), 0, s, v);
  } else {
    const int qblocks = (v.n_query + kQBlock - 1) / kQBlock;
    hipLaunchKernelGGL(k_orb_bruteforce, dim3((unsigned)(v.n_pairs * qblocks), (unsigned)v.n_split), dim3(kQBlock), 0, s, v);
    if (v.n_split > 1) hipLaunchKernelGGL(k_orb_merge, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, v);
  }
  if (t) c->timer.end(s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("orb kernel launch failed: %s", hipGetErrorString(e)); return OSH_ERR_DEVICE; }
  OSH_HIP(hipStreamSynchronize(s));
  if (c->timer.enabled) c->timer.collect();
  c->matched = true;
  return OSH_OK;
}

static int launch_search(osh_orb_ctx* c, hipStream_t s) {
  const OrbView& v = c->v;
  const size_t nq = (size_t)v.n_pairs * v.n_query;
  if (c->grid) {
    const unsigned grid = (unsigned)((nq + (kQBlock / 64) - 1) / (kQBlock / 64));
    hipLaunchKernelGGL(k_orb_grid, dim3(grid), dim3(kQBlock), 0, s, v);
  } else if (c->windowed) {
    const unsigned grid = (unsigned)((nq + (kQBlock / 64) - 1) / (kQBlock / 64));
    hipLaunchKernelGGL(k_orb_windowed, dim3(grid), dim3(kQBlock), 0, s, v);
  } else {
    const int qblocks = (v.n_query + kQBlock - 1) / kQBlock;
    hipLaunchKernelGGL(k_orb_bruteforce, dim3((unsigned)(v.n_pairs * qblocks), (unsigned)v.n_split), dim3(kQBlock), 0, s, v);
    if (v.n_split > 1) hipLaunchKernelGGL(k_orb_merge, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, v);
  }
  return OSH_OK;
}

extern "C" int osh_orb_match_local_points(osh_orb_ctx* c, float nn_ratio, int32_t th_high, const uint8_t* occupied, const uint8_t* query_blocks,
                                          int32_t* assignment, int32_t* n_matches, int32_t* query_slot, int32_t* rounds) {
  if (!c || !c->uploaded || !assignment || !n_matches) { set_error("osh_orb_match_local_points: nothing uploaded or NULL outputs"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const OrbView& v = c->v;
  const size_t nq = (size_t)v.n_pairs * v.n_query, nt = (size_t)v.n_pairs * v.n_train;
  if (rounds) *rounds = 0;
  for (int p = 0; p < v.n_pairs; ++p) n_matches[p] = 0;
  for (size_t i = 0; i < nt; ++i) assignment[i] = -1;
  if (query_slot) for (size_t i = 0; i < nq; ++i) query_slot[i] = -1;
  if (nq == 0 || nt == 0) { c->matched = true; return OSH_OK; }
  OSH_TRY(c->d_claim.reserve(nq * 4));
  for (int k = 0; k < 2; ++k) OSH_TRY(c->d_owner[k].reserve(nt * 4));
  for (int k = 0; k < 5; ++k) OSH_TRY(c->d_cur[k].reserve(nq * 4));
  OSH_TRY(c->d_state.reserve(16)); OSH_TRY(c->d_assign.reserve(nt * 4)); OSH_TRY(c->d_nmatch.reserve((size_t)v.n_pairs * 4));
  ResolveView r{};
  r.claim = c->d_claim.as<int>();
  r.owner[0] = c->d_owner[0].as<int>(); r.owner[1] = c->d_owner[1].as<int>();
  for (int k = 0; k < 5; ++k) r.cur[k] = c->d_cur[k].as<int>();
  r.state = c->d_state.as<int>();
  r.nn_ratio = nn_ratio; r.th_high = th_high;
  if (occupied) {
    OSH_TRY(c->d_pre.reserve(nt));
    OSH_HIP(hipMemcpyAsync(c->d_pre.p, occupied, nt, hipMemcpyHostToDevice, s));
    r.pre = c->d_pre.as<unsigned char>();
  }
  if (query_blocks) {
    OSH_TRY(c->d_blocks.reserve(nq));
    OSH_HIP(hipMemcpyAsync(c->d_blocks.p, query_blocks, nq, hipMemcpyHostToDevice, s));
    r.blocks = c->d_blocks.as<unsigned char>();
  }
  if (c->timer.enabled) OSH_TRY(c->timer.init());
  const bool t = c->timer.begin(0, s);
  OSH_TRY(launch_search(c, s));
  if (t) c->timer.end(s);
  const bool t1 = c->timer.begin(1, s);
  OSH_HIP(hipMemsetAsync(c->d_claim.p, 0xFE, nq * 4, s));            // no claim equals this value: round 0 always counts as a change
  OSH_HIP(hipMemsetAsync(c->d_owner[0].p, 0x7F, nt * 4, s));         // 0x7f7f7f7f: above every query index
  OSH_HIP(hipMemsetAsync(c->d_owner[1].p, 0x7F, nt * 4, s));
  OSH_HIP(hipMemsetAsync(c->d_state.p, 0, 16, s));
  OSH_HIP(hipMemsetAsync(c->d_assign.p, 0xFF, nt * 4, s));
  OSH_HIP(hipMemsetAsync(c->d_nmatch.p, 0, (size_t)v.n_pairs * 4, s));
  const unsigned gq256 = (unsigned)((nq + 255) / 256), gt256 = (unsigned)((nt + 255) / 256);
  const unsigned gwave = (unsigned)((nq + (kQBlock / 64) - 1) / (kQBlock / 64));
  int state[3] = {0, 0, 0};
  int round = 0;
  // a pre-occupied slot closes candidates for every query from the start: round 0 searches again where the unrestricted result touches one
  if (r.pre) {
    if (c->grid) hipLaunchKernelGGL(k_orb_research<2>, dim3(gwave), dim3(kQBlock), 0, s, v, r, 1);   // owner[1]: all open
    else if (c->windowed) hipLaunchKernelGGL(k_orb_research<1>, dim3(gwave), dim3(kQBlock), 0, s, v, r, 1);
    else hipLaunchKernelGGL(k_orb_research<0>, dim3(gwave), dim3(kQBlock), 0, s, v, r, 1);
  }
  const int max_rounds = v.n_query + 2;
  while (!state[1]) {
    if (round >= max_rounds) { set_error("osh_orb_match_local_points: occupancy rounds did not settle"); return OSH_ERR_DEVICE; }
    for (int k = 0; k < 4; ++k, ++round) {
      hipLaunchKernelGGL(k_orb_claim, dim3(gq256), dim3(256), 0, s, v, r, (round == 0 && !r.pre) ? 0 : round + 2);
      hipLaunchKernelGGL(k_orb_round_end, dim3(gt256), dim3(256), 0, s, v, r, round);
      if (c->grid) hipLaunchKernelGGL(k_orb_research<2>, dim3(gwave), dim3(kQBlock), 0, s, v, r, round);
      else if (c->windowed) hipLaunchKernelGGL(k_orb_research<1>, dim3(gwave), dim3(kQBlock), 0, s, v, r, round);
      else hipLaunchKernelGGL(k_orb_research<0>, dim3(gwave), dim3(kQBlock), 0, s, v, r, round);
    }
    OSH_HIP(hipMemcpyAsync(state, c->d_state.p, 12, hipMemcpyDeviceToHost, s));
    OSH_HIP(hipStreamSynchronize(s));
  }
  hipLaunchKernelGGL(k_orb_assign, dim3((unsigned)v.n_pairs), dim3(256), 0, s, v, r, c->d_assign.as<int>(), c->d_nmatch.as<int>());
  if (t1) c->timer.end(s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("orb kernel launch failed: %s", hipGetErrorString(e)); return OSH_ERR_DEVICE; }
  OSH_HIP(hipMemcpyAsync(assignment, c->d_assign.p, nt * 4, hipMemcpyDeviceToHost, s));
  OSH_HIP(hipMemcpyAsync(n_matches, c->d_nmatch.p, (size_t)v.n_pairs * 4, hipMemcpyDeviceToHost, s));
  if (query_slot) OSH_HIP(hipMemcpyAsync(query_slot, c->d_claim.p, nq * 4, hipMemcpyDeviceToHost, s));
  OSH_HIP(hipStreamSynchronize(s));
  if (c->timer.enabled) c->timer.collect();
  if (rounds) *rounds = state[2];
  c->matched = true;
  return OSH_OK;
}

extern "C" int osh_orb_list_distances(osh_orb_ctx* c, int32_t* dist_out) {
  if (!c || !c->uploaded || !c->windowed || c->grid || !dist_out) { set_error("osh_orb_list_distances: upload candidate lists first (osh_orb_upload with cand_off)"); return OSH_ERR_INVALID; }
  if (c->list_total == 0) return OSH_OK;
  OSH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  OSH_TRY(c->d_ldist.reserve((size_t)c->list_total * 4));
  hipLaunchKernelGGL(k_orb_list_dist, dim3((unsigned)((c->list_longest_pair + 255) / 256), (unsigned)c->v.n_pairs), dim3(256), 0, s, c->v, c->d_ldist.as<int>());
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { set_error("k_orb_list_dist launch failed: %s", hipGetErrorString(le)); return OSH_ERR_DEVICE; }
  OSH_HIP(hipMemcpyAsync(dist_out, c->d_ldist.p, (size_t)c->list_total * 4, hipMemcpyDeviceToHost, s));
  OSH_HIP(hipStreamSynchronize(s));
  return OSH_OK;
}

extern "C" int osh_orb_download(osh_orb_ctx* c, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist,
                                int32_t* best_level, int32_t* second_level, int32_t* second_idx) {
  if (!c || !c->matched) { set_error("osh_orb_download: call osh_orb_match first"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  const size_t nq = (size_t)c->v.n_pairs * c->v.n_query;
  int32_t* outs[6] = {best_idx, best_dist, second_dist, best_level, second_level, second_idx};
  for (int k = 0; k < 6; ++k)
    if (outs[k] && nq) OSH_HIP(hipMemcpyAsync(outs[k], c->d_out[k].p, nq * 4, hipMemcpyDeviceToHost, c->stream));
  OSH_HIP(hipStreamSynchronize(c->stream));
  return OSH_OK;
}

extern "C" int osh_orb_set_profiling(osh_orb_ctx* c, int enable) {
  if (!c) return OSH_ERR_INVALID;
  OSH_HIP(hipSetDevice(c->device));
  if (enable) OSH_TRY(c->timer.init());
  c->timer.enabled = enable != 0;
  c->timer.reset();
  return OSH_OK;
}

extern "C" int osh_orb_get_profile(osh_orb_ctx* c, int64_t* launches, double* total_ms) {
  if (!c || !launches || !total_ms) return OSH_ERR_INVALID;
  *launches = c->timer.launches[0];
  *total_ms = c->timer.total_ms[0];
  return OSH_OK;
}

extern "C" int osh_orb_get_resolve_profile(osh_orb_ctx* c, int64_t* launches, double* total_ms) {
  if (!c || !launches || !total_ms) return OSH_ERR_INVALID;
  *launches = c->timer.launches[1];
  *total_ms = c->timer.total_ms[1];
  return OSH_OK;
}

extern "C" int osh_orb_frustum(osh_orb_ctx* c, const osh_frustum_frame* fr, const osh_frustum_points* pts, osh_frustum_result* res) {
  if (!c || !fr || !pts || !res || pts->n < 0) { set_error("osh_orb_frustum: bad arguments"); return OSH_ERR_INVALID; }
  const int n = pts->n;
  if (n == 0) return OSH_OK;
  if (!pts->pos || !pts->normal || !pts->min_dist || !pts->max_dist || !res->stage || !res->proj_x || !res->proj_y || !res->proj_xr ||
      !res->depth || !res->view_cos || !res->level) { set_error("osh_orb_frustum: NULL array"); return OSH_ERR_INVALID; }
  if (fr->n_scale_levels <= 0 || !(fr->log_scale_factor > 0.f)) { set_error("osh_orb_frustum: bad scale pyramid"); return OSH_ERR_INVALID; }
  OSH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  c->h_fin.resize((size_t)n * 8); c->h_fout.resize((size_t)n * 8);
  for (int i = 0; i < n; ++i) {
    float* d = &c->h_fin[(size_t)i * 8];
    d[0] = pts->pos[3 * i]; d[1] = pts->pos[3 * i + 1]; d[2] = pts->pos[3 * i + 2];
    d[3] = pts->normal[3 * i]; d[4] = pts->normal[3 * i + 1]; d[5] = pts->normal[3 * i + 2];
    d[6] = pts->min_dist[i]; d[7] = pts->max_dist[i];
  }
  OSH_TRY(c->d_fin.reserve((size_t)n * 32));
  OSH_TRY(c->d_fout.reserve((size_t)n * 32));
  OSH_HIP(hipMemcpyAsync(c->d_fin.p, c->h_fin.data(), (size_t)n * 32, hipMemcpyHostToDevice, s));
  FrustumFrame f;
  for (int k = 0; k < 9; ++k) f.R[k] = fr->Rcw[k];
  for (int k = 0; k < 3; ++k) { f.t[k] = fr->tcw[k]; f.O[k] = fr->Ow[k]; }
  f.fx = fr->fx; f.fy = fr->fy; f.cx = fr->cx; f.cy = fr->cy; f.bf = fr->bf;
  f.min_x = fr->min_x; f.max_x = fr->max_x; f.min_y = fr->min_y; f.max_y = fr->max_y;
  f.log_sf = fr->log_scale_factor; f.cos_limit = fr->viewing_cos_limit; f.levels = fr->n_scale_levels;
  f.fisheye = fr->fisheye ? 1 : 0;
  for (int k = 0; k < 4; ++k) f.kb[k] = fr->fisheye ? fr->kb8[k] : 0.f;
  hipLaunchKernelGGL(k_frustum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, f, n, c->d_fin.as<float4>(), c->d_fout.as<float4>());
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("k_frustum launch failed: %s", hipGetErrorString(e)); return OSH_ERR_DEVICE; }
  OSH_HIP(hipMemcpyAsync(c->h_fout.data(), c->d_fout.p, (size_t)n * 32, hipMemcpyDeviceToHost, s));
  OSH_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i) {
    const float* o = &c->h_fout[(size_t)i * 8];
    int32_t st, lv;
    std::memcpy(&st, &o[0], 4); std::memcpy(&lv, &o[6], 4);
    res->stage[i] = (uint8_t)st; res->proj_x[i] = o[1]; res->proj_y[i] = o[2]; res->proj_xr[i] = o[3];
    res->depth[i] = o[4]; res->view_cos[i] = o[5]; res->level[i] = lv;
  }
  return OSH_OK;
}

extern "C" int osh_orb_distance_matrix(osh_orb_ctx* c, int32_t n, int32_t m, const uint8_t* a, const uint8_t* b, int32_t* out) {
  if (!c || n < 0 || m < 0 || !a || !b || !out) { set_error("osh_orb_distance_matrix: bad arguments"); return OSH_ERR_INVALID; }
  if (n == 0 || m == 0) return OSH_OK;
  OSH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  OSH_TRY(c->d_a.reserve((size_t)n * 32));
  OSH_TRY(c->d_b.reserve((size_t)m * 32));
  OSH_TRY(c->d_dm.reserve((size_t)n * m * 4));
  OSH_HIP(hipMemcpyAsync(c->d_a.p, a, (size_t)n * 32, hipMemcpyHostToDevice, s));
  OSH_HIP(hipMemcpyAsync(c->d_b.p, b, (size_t)m * 32, hipMemcpyHostToDevice, s));
  const size_t tot = (size_t)n * m;
  hipLaunchKernelGGL(k_orb_distance_matrix, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, n, m, c->d_a.as<uint4>(), c->d_b.as<uint4>(), c->d_dm.as<int>());
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("k_orb_distance_matrix launch failed: %s", hipGetErrorString(e)); return OSH_ERR_DEVICE; }
  OSH_HIP(hipMemcpyAsync(out, c->d_dm.p, tot * 4, hipMemcpyDeviceToHost, s));
  OSH_HIP(hipStreamSynchronize(s));
  return OSH_OK;
}
