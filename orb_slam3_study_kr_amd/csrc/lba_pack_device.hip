// lba_pack_device.hip -- the batch packer of osh_lba_upload as HIP kernels (gfx950).
//
// What SparseOptimizer::initializeOptimization + BlockSolver::buildStructure do inside the call this library replaces
// (Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:199-267, block_solver.hpp:143-295): index mapping, landmark-major edge order
// and the structure of the Schur complement.  lba_pack.h does it with host threads (1.6 ms per 50-keyframe window and thread:
// the bottleneck of an upload + optimize + download pipeline); here the caller's arrays go to the device in the caller's order
// and three kernels, ONE THREAD BLOCK PER WINDOW, build exactly the layout lba_pack.h builds (tests compare them byte by byte):
//
//   k_pack_pre1   histogram of the edges by landmark (workgroup-scope atomics), offsets, scatter, order inside a landmark by
//                 rank counting on the unique key (pose, kind), optimisable observers per landmark, number of plan units
//   k_pack_pre2   units (landmark x part pair) and their keys (the observer sets), grouping of equal keys in an open-addressing
//                 table, bitonic sort of the distinct keys, stable placement of the units behind their key (tiles of 1024 sorted
//                 in LDS), landmark renumbering along the owner units, chunks of the landmark-major kernels, the greedy merge of
//                 consecutive keys into items (sequential by nature: one thread, keys staged in LDS), slot bytes and live pose
//                 pairs of every item
//   k_pack_post   the arenas: renumbered landmarks and edges, observation records, items, records, contribution slots in plan
//                 order (stable placement again), block ranges of k_schur_reduce
//
// Between the kernels the host reads one small summary per window (sizes of what comes next) and lays out the next buffers:
// two round trips per upload.  Every reduction is an integer count or a sort on unique keys, so the result does not depend on
// scheduling.  The two edges a fisheye stereo rig puts on one (keyframe, landmark) block are found by the same rank counting and
// leave the sorted lists through a prefix count (k_pack_pre1).
#include "lba_pack_device.h"

#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

namespace osh {
namespace dpack {

typedef unsigned long long u64;
constexpr int NT = 1024;       // threads of a window's block
constexpr int NWV = NT / 64;
constexpr int GL = 1024;       // distinct keys whose sort and merge walk run in LDS
constexpr int LCH = 10 * GL;   // landmarks whose chunk chain is walked in LDS (the key area reused as ints)

struct PWin {                  // host -> device, per window
  int P, F, L, E;
  int flags;                   // bit 0: KannalaBrandt8 window, bit 1: rig
  int pose_off, fpose_off, pt_off, edge_off, lmoff_off;
  int nu, tcap, E2, pad_e;     // after round trip 1: plan units, table capacity (power of two >= 2 nu), sorted edges (merged rig pairs count once)
  long long s1, s2, s3;        // scratch offsets: ints, bytes, ints
  // after round trip 2
  int chunk_off, sym_item_off, cross_item_off, sym_rec_off, cross_rec_off, rblk_off, contrib_off, ccontrib_off;
};
struct PSum {                  // device -> host, per window
  unsigned err_a;              // first bad edge << 2 | class (0 index / kind, 1 stereo edge of a fisheye window, 2 body edge without a rig)
  int err_k;                   // a landmark with more than 254 optimisable observers (or P >= 0xffff)
  u64 err_d;                   // landmark << 32 | pose of the first duplicated (pose, landmark) pair
  int nu, ng, n_builds, n_sym, n_cross, recs_sym, recs_cross, n_contrib, n_ccontrib, n_chunks, internal, E2;
  long long tile_steps, pair_blocks;
  long long cyc[24];           // shader-clock cycles per phase (thread 0): [0..7] k_pack_pre1, [8..17] k_pack_pre2, [18..23] k_pack_post
};
struct DBuild {                // one item being built (64 bytes)
  int base, n, cls_idx, rec_rel;
  unsigned short X[8], Y[8];   // 0xffff unused
  int shape;                   // nx | ny << 8 | sym << 16
  int clive;
  u64 live;
};
static_assert(sizeof(DBuild) == 64, "DBuild layout");

struct PackArgs {
  const PWin* win;
  PSum* sum;
  const double *r_pose, *r_cam, *r_pt;
  const int *r_epose, *r_epoint;
  const unsigned char* r_kind;
  const void* r_rec;
  int rec_f32;
  int* s1;
  unsigned char* s2;
  int* s3;
  double *a_pose, *a_cam, *a_pt;
  void* a_rec;
  int *a_epose, *a_epoint, *a_eorig, *a_eorig2, *a_lmoff, *a_lmperm, *a_fpw;
  double* a_rec2;
  int has_rig;
  int item_max;   // landmarks per item at most (schur_plan.h:item_max_lm)
  int chunk_edges;   // edges per chunk at most (lba_pack.h:chunk_max_edges)
  unsigned char* a_ekind;
  Chunk* a_chunks;
  SItem* a_items;
  SRec* a_recs;
  int *a_spair, *a_scslot, *a_posex, *a_posey;
  RBlk* a_rblk;
  I2* a_crange;
  int* ptwin;
};

// ---- scratch layouts (shared by host and device)
// order / sepose / slm / lmo: the sorted edges.  A fisheye-rig window merges the (left EdgeSE3ProjectXYZ, right EdgeSE3ProjectXYZToBody) edges of
// one (keyframe, landmark) pair into ONE sorted edge: its sorted lists are compacted into the *2 arrays (second2: caller index of the merged
// right edge or -1, kind2: sorted-edge kind) and s1_final() makes the plain names point at them for the kernels that follow k_pack_pre1.
struct S1 { int *lmo, *fill, *t_e, *t_key, *t_lm, *order, *sepose, *slm, *nfree, *unit_off, *sec, *lmo2, *order2, *sepose2, *slm2, *second2, *kind2; };
__host__ __device__ inline size_t s1_ints(int L, int E, bool rig) { return 4 * (size_t)L + 6 * (size_t)E + 8 + (rig ? (size_t)L + 6 * (size_t)E + 8 : 0); }
__host__ __device__ inline S1 s1_of(int* b, int L, int E, bool rig) {
  S1 s;
  s.lmo = b; s.fill = s.lmo + (L + 1); s.t_e = s.fill + (L + 1); s.t_key = s.t_e + E; s.t_lm = s.t_key + E; s.order = s.t_lm + E; s.sepose = s.order + E;
  s.slm = s.sepose + (E + 1); s.nfree = s.slm + E; s.unit_off = s.nfree + L;
  int* r = s.unit_off + (L + 1);
  s.sec = r; s.lmo2 = s.sec + (E + 1); s.order2 = s.lmo2 + (L + 1); s.sepose2 = s.order2 + E; s.slm2 = s.sepose2 + (E + 1); s.second2 = s.slm2 + E; s.kind2 = s.second2 + E;
  if (!rig) s.sec = s.lmo2 = s.order2 = s.sepose2 = s.slm2 = s.second2 = s.kind2 = nullptr;
  return s;
}
__host__ __device__ inline void s1_final(S1& s) { if (s.order2) { s.lmo = s.lmo2; s.order = s.order2; s.sepose = s.sepose2; s.slm = s.slm2; } }
struct S2 {
  u64 *uk[5], *uxs, *uys, *gk[5];
  DBuild* builds;
  int *ulm, *uab, *ugid, *uorder, *plm, *pab, *gcnt, *gslot, *gfill, *table, *gcount, *grank, *glist, *old2new, *perm, *lmo_new, *nextc, *delta;
  Chunk* chunks;
};
__host__ __device__ inline size_t s2_bytes(int nu, int tcap, int L) {
  return (size_t)nu * (12 * 8 + 64 + 9 * 4) + (size_t)tcap * 4 * 4 + ((size_t)L * 5 + 2) * 4 + (size_t)L * 12 + 64;
}
__host__ __device__ inline S2 s2_of(unsigned char* b, int nu, int tcap, int L) {
  S2 s;
  u64* q = reinterpret_cast<u64*>(b);
  for (int k = 0; k < 5; ++k) { s.uk[k] = q; q += nu; }
  s.uxs = q; q += nu; s.uys = q; q += nu;
  for (int k = 0; k < 5; ++k) { s.gk[k] = q; q += nu; }
  s.builds = reinterpret_cast<DBuild*>(q); q += (size_t)nu * 8;
  int* p = reinterpret_cast<int*>(q);
  s.ulm = p; p += nu; s.uab = p; p += nu; s.ugid = p; p += nu; s.uorder = p; p += nu; s.plm = p; p += nu; s.pab = p; p += nu; s.gcnt = p; p += nu; s.gslot = p; p += nu; s.gfill = p; p += nu;
  s.table = p; p += tcap; s.gcount = p; p += tcap; s.grank = p; p += tcap; s.glist = p; p += tcap;
  s.old2new = p; p += L; s.perm = p; p += L; s.lmo_new = p; p += L + 1; s.nextc = p; p += L + 1; s.delta = p; p += L;
  s.chunks = reinterpret_cast<Chunk*>(p);
  return s;
}
struct S3 { int *cnt, *ccnt, *bpre, *cpre, *ent, *cent; };
__host__ __device__ inline size_t s3_ints(int P, int nb, int nc, int ncc) { return (size_t)P * (P + 1) / 2 + 1 + (P + 1) + 2 * ((size_t)nb + 1) + nc + ncc + 8; }
__host__ __device__ inline S3 s3_of(int* b, int P, int nb, int nc, int ncc) {
  S3 s;
  const size_t nblk = (size_t)P * (P + 1) / 2;
  s.cnt = b; s.ccnt = s.cnt + nblk + 1; s.bpre = s.ccnt + (P + 1); s.cpre = s.bpre + (nb + 1); s.ent = s.cpre + (nb + 1); s.cent = s.ent + nc;
  return s;
}

// ---- device helpers
#define OSH_WG __HIP_MEMORY_SCOPE_WORKGROUP
__device__ __forceinline__ int wg_add(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, OSH_WG); }
// a value other threads of the block produced with atomics: read it where the atomics were performed (L2), not from the L1
__device__ __forceinline__ int ld_l2(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Runs of equal keys in consecutive lanes of a wavefront (edges arrive landmark by landmark): `head` lane of the caller's run and the
// run's length, so that one lane per run touches the landmark's counter (64 same-address atomics of one instruction serialise in L2).
__device__ __forceinline__ void wave_runs(int key, int& head, int& len) {
  const int lane = threadIdx.x & 63;
  const int prev = __shfl_up(key, 1, 64);
  const u64 hm = __ballot(lane == 0 || prev != key);
  const u64 below = hm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
  head = 63 - __clzll((long long)below);
  const u64 above = (lane == 63) ? 0ull : (hm >> (lane + 1));
  len = (above ? (lane + 1 + (__ffsll((long long)above) - 1)) : 64) - head;
}

__device__ __forceinline__ int block_excl_scan(int v, int* sh, int& total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
  __syncthreads();
  if (lane == 63) sh[wv] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < NWV; ++k) { const int x = sh[k]; if (k < wv) base += x; tot += x; }
  total = tot;
  return base + inc - v;
}
__device__ __forceinline__ int block_incl_max_scan(int v, int* sh) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int m = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(m, o, 64); if (lane >= o) m = max(m, t); }
  __syncthreads();
  if (lane == 63) sh[wv] = m;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int k = 0; k < NWV; ++k) { const int x = sh[k]; if (k < wv) base = max(base, x); }
  return max(base, m);
}
// a[i] <- sum of a[k], k < i, for i < n (values possibly produced by atomics); returns the total
__device__ int block_scan_array(int* a, int n, int* sh) {
  const int tid = threadIdx.x;
  const int per = (n + NT - 1) / NT;
  const int lo = min(tid * per, n), hi = min(lo + per, n);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += ld_l2(a + i);
  int total;
  int base = block_excl_scan(s, sh, total);
  for (int i = lo; i < hi; ++i) { const int v = ld_l2(a + i); a[i] = base; base += v; }
  __syncthreads();
  return total;
}

// ascending sort of the 1024 keys of a block, one per thread, in LDS (strides below 64 stay inside a wavefront: shuffles)
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
  const int lo = __shfl_xor((int)(unsigned)v, m, 64), hi = __shfl_xor((int)(unsigned)(v >> 32), m, 64);
  return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ void bitonic_sort_tile(u64* shk) {
  const int tid = threadIdx.x;
  u64 v = shk[tid];
  for (int k = 2; k <= NT; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      u64 o;
      if (j >= 64) {
        __syncthreads();
        shk[tid] = v;
        __syncthreads();
        o = shk[tid ^ j];
      } else {
        o = shfl_xor_u64(v, j);
      }
      const bool keep_min = ((tid & j) == 0) == ((tid & k) == 0);
      v = keep_min ? (v < o ? v : o) : (v < o ? o : v);
    }
  }
  __syncthreads();
  shk[tid] = v;
  __syncthreads();
}

// pos(i) = fill[bin(i)] + number of items i' < i of the same bin, for items 0..n-1; fill[b] holds the start of bin b on entry and
// its end on return.  Tiles of 1024 items: sort (bin, thread) in LDS, runs of equal bins take consecutive places.
template <class BinF, class OutF>
__device__ void stable_place(int n, BinF bin_of, int* fill, OutF out, u64* shk, int* shi) {
  const int tid = threadIdx.x;
  for (int base = 0; base < n; base += NT) {
    const int i = base + tid;
    const unsigned b = i < n ? (unsigned)bin_of(i) : 0xffffffffu;
    __syncthreads();
    shk[tid] = ((u64)b << 32) | (unsigned)tid;
    __syncthreads();
    bitonic_sort_tile(shk);
    const u64 me = shk[tid];
    const unsigned mb = (unsigned)(me >> 32);
    const bool head = tid == 0 || (unsigned)(shk[tid - 1] >> 32) != mb;
    const bool tail = tid == NT - 1 || (unsigned)(shk[tid + 1] >> 32) != mb;
    const int hp = block_incl_max_scan(head ? tid : 0, shi);
    const int r = tid - hp;
    int f = 0;
    if (mb != 0xffffffffu) { f = fill[mb]; out(base + (int)(unsigned)me, f + r); }
    __syncthreads();
    if (mb != 0xffffffffu && tail) fill[mb] = f + r + 1;
    __syncthreads();
  }
}

// ascending bitonic sort of arr[0..npad) (npad a power of two; negative entries sort last)
template <class LessF>
__device__ void block_bitonic_idx(int* arr, int npad, LessF less) {
  const int tid = threadIdx.x;
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (npad >> 1); t += NT) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const int a = arr[i], b = arr[l];
        const bool up = (i & k) == 0;
        const bool sw = up ? less(b, a) : less(a, b);
        if (sw) { arr[i] = b; arr[l] = a; }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ void pack_poses_dev(const int* obs, int r0, int r1, u64& o0, u64& o1) {
  u64 out[2] = {~0ull, ~0ull};
  for (int r = r0; r < r1; ++r) {
    const int s = r - r0, sh = 48 - 16 * (s & 3);
    const u64 v = (u64)(unsigned)obs[r] << sh, m = ~(0xffffull << sh);
    if (s < 4) out[0] = (out[0] & m) | v; else out[1] = (out[1] & m) | v;
  }
  o0 = out[0]; o1 = out[1];
}
__device__ __forceinline__ int units_of(int k) { if (k <= kItemPoses) return 1; const int np = (k + kItemPoses - 1) / kItemPoses; return np * (np + 1) / 2; }
__device__ __forceinline__ int tiles_of_dev(int n_poses) { return (6 * n_poses + 15) / 16; }

// =============================================================================================
// k_pack_pre1
// =============================================================================================
__global__ __launch_bounds__(NT) void k_pack_pre1(PackArgs a) {
  __shared__ int shi[32];
  __shared__ unsigned sh_bad;
  __shared__ u64 sh_dup;
  __shared__ long long sh_pb;
  __shared__ int sh_k;
  const int w = blockIdx.x, tid = threadIdx.x;
  const PWin pw = a.win[w];
  PSum& sum = a.sum[w];
  const int L = pw.L, E = pw.E, P = pw.P, NPw = pw.P + pw.F;
  const bool rig = pw.flags & 2;
  const S1 s = s1_of(a.s1 + pw.s1, L, E, rig);
  const int* r_epose = a.r_epose + pw.edge_off;
  const int* r_epoint = a.r_epoint + pw.edge_off;
  const unsigned char* r_kind = a.r_kind + pw.edge_off;
  if (tid == 0) { sh_bad = 0xffffffffu; sh_dup = ~0ull; sh_pb = 0; sh_k = (P >= 0xffff) ? 1 : 0; }
  for (int j = tid; j <= L; j += NT) s.lmo[j] = 0;
  for (int j = tid; j < L; j += NT) s.nfree[j] = 0;
  __syncthreads();
  // ---- A: validation + landmark histogram
  const bool kb8 = pw.flags & 1;
  long long tc_last = clock64();
  int tc_i = 0;
#define OSH_TC() do { if (tid == 0) { const long long _n = clock64(); sum.cyc[tc_i] = _n - tc_last; tc_last = _n; } ++tc_i; } while (0)
  unsigned bad = 0xffffffffu;
  for (int base = 0; base < E; base += NT) {
    const int e = base + tid;
    int il = -1 - (tid & 63);           // lanes without a countable edge: a key of their own
    if (e < E) {
      const int ip = r_epose[e], jl = r_epoint[e], kd = r_kind[e];
      int cls = -1;
      if (ip < 0 || ip >= NPw || jl < 0 || jl >= L || kd > OSH_EDGE_BODY) cls = 0;
      else if (kb8 && kd == OSH_EDGE_STEREO) cls = 1;
      else if (kd == OSH_EDGE_BODY && !rig) cls = 2;
      if (cls >= 0) bad = min(bad, ((unsigned)e << 2) | (unsigned)cls); else il = jl;
    }
    int head, len;
    wave_runs(il, head, len);
    if (il >= 0 && head == (tid & 63)) wg_add(&s.lmo[il], len);
  }
  if (bad != 0xffffffffu) atomicMin(&sh_bad, bad);
  __syncthreads();
  if (sh_bad != 0xffffffffu) {
    if (tid == 0) { sum.err_a = sh_bad; sum.err_k = 0; sum.err_d = ~0ull; sum.nu = 0; sum.internal = 0; }
    return;
  }
  OSH_TC();   // 0: histogram
  // ---- B: offsets
  block_scan_array(s.lmo, L + 1, shi);
  for (int j = tid; j <= L; j += NT) s.fill[j] = s.lmo[j];
  __syncthreads();
  OSH_TC();   // 1: offsets
  // ---- C: scatter (any order inside a landmark)
  for (int base = 0; base < E; base += NT) {
    const int e = base + tid;
    int il = -1 - (tid & 63), key = 0;
    if (e < E) { il = r_epoint[e]; key = (r_epose[e] << 2) | r_kind[e]; }
    int head, len;
    wave_runs(il, head, len);
    int x0 = 0;
    if (il >= 0 && head == (tid & 63)) x0 = wg_add(&s.fill[il], len);
    const int x = __shfl(x0, head, 64) + ((tid & 63) - head);
    if (e < E) { s.t_e[x] = e; s.t_key[x] = key; s.t_lm[x] = il; }
  }
  __syncthreads();
  OSH_TC();   // 2: scatter
  // ---- D: order inside a landmark = rank of the unique key (pose, kind); a pose twice on one landmark is refused
  u64 dup = ~0ull;
  for (int xb = tid; xb < E; xb += 2 * NT) {   // two positions per thread and step: their dependent gathers are in flight together
    int e[2], key[2], il[2], lo[2], hi[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int x = min(xb + u * NT, E - 1); e[u] = s.t_e[x]; key[u] = s.t_key[x]; il[u] = s.t_lm[x]; }
#pragma unroll
    for (int u = 0; u < 2; ++u) { lo[u] = s.lmo[il[u]]; hi[u] = s.lmo[il[u] + 1]; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int x = xb + u * NT;
      if (x >= E) continue;
      int rank = 0, nf = 0, same = 0, partner = -1;
      for (int y = lo[u]; y < hi[u]; ++y) {
        const int ky = s.t_key[y];
        rank += ky < key[u];
        if (y != x && (ky >> 2) == (key[u] >> 2)) { ++same; partner = ky & 3; }
        nf += (ky >> 2) < P;
      }
      // a pose twice on one landmark: only the (left mono, right body) pair of a fisheye rig, which shares one Hessian block
      const int mine = key[u] & 3;
      const bool pair_ok = rig && same == 1 && ((mine == OSH_EDGE_MONO && partner == OSH_EDGE_BODY) || (mine == OSH_EDGE_BODY && partner == OSH_EDGE_MONO));
      if (same > 0 && !pair_ok) dup = min(dup, ((u64)(unsigned)il[u] << 32) | (unsigned)(key[u] >> 2));
      s.order[lo[u] + rank] = e[u];
      s.sepose[lo[u] + rank] = key[u] >> 2;
      s.slm[lo[u] + rank] = il[u];
      if (rig) s.sec[lo[u] + rank] = (same == 1 && mine == OSH_EDGE_BODY) ? 1 : 0;   // merged into the sorted edge before it (its mono partner)
      if (x == lo[u] && !rig) s.nfree[il[u]] = nf;
    }
  }
  if (tid == 0) s.sepose[E] = 0;
  if (dup != ~0ull) atomicMin(&sh_dup, dup);
  __syncthreads();
  int E2 = E;
  if (rig) {
    // merged right edges leave the sorted lists: exclusive count of them before every position, then the compacted copies
    if (tid == 0) s.sec[E] = 0;
    __syncthreads();
    const int nmerged = block_scan_array(s.sec, E + 1, shi);   // sec[x] <- merged edges before x; a merged edge x has sec[x + 1] == sec[x] + 1
    E2 = E - nmerged;
    for (int x = tid; x < E; x += NT) {
      const int before = s.sec[x];
      if (s.sec[x + 1] != before) continue;                    // x itself is a merged right edge
      const int nx = x - before, e = s.order[x], jl = s.slm[x];
      const bool has2 = x + 1 < s.lmo[jl + 1] && s.sec[x + 2 <= E ? x + 2 : E] != s.sec[x + 1];
      const int kd = r_kind[e];
      s.order2[nx] = e; s.sepose2[nx] = s.sepose[x]; s.slm2[nx] = jl;
      s.second2[nx] = has2 ? s.order[x + 1] : -1;
      s.kind2[nx] = has2 ? kKindBoth : (kd == OSH_EDGE_BODY ? kKindBody : kd);
    }
    for (int j = tid; j <= L; j += NT) s.lmo2[j] = s.lmo[j] - s.sec[s.lmo[j]];
    if (tid == 0) s.sepose2[E2] = 0;
    __syncthreads();
    for (int j = tid; j < L; j += NT) {   // optimisable observers: poses come ascending, the optimisable ones first
      int nf = 0;
      for (int x = s.lmo2[j]; x < s.lmo2[j + 1]; ++x) nf += s.sepose2[x] < P;
      s.nfree[j] = nf;
    }
    __syncthreads();
  }
  OSH_TC();   // 3: order inside the landmarks
  // ---- E: plan units per landmark
  long long pbk = 0;
  int toomany = 0;
  for (int j = tid; j < L; j += NT) {
    const int k = s.nfree[j];
    if (k > 254) toomany = 1;
    s.unit_off[j] = units_of(min(k, 254));
    pbk += (long long)k * (k + 1) / 2;
  }
  if (tid == 0) s.unit_off[L] = 0;
  if (toomany) atomicOr(&sh_k, 1);
  atomicAdd((u64*)&sh_pb, (u64)pbk);
  __syncthreads();
  const int nu = block_scan_array(s.unit_off, L + 1, shi);
  OSH_TC();   // 4: units per landmark
  if (tid == 0) {
    sum.err_a = 0xffffffffu; sum.err_d = sh_dup; sum.err_k = sh_k; sum.nu = nu; sum.pair_blocks = sh_pb; sum.internal = 0; sum.E2 = E2;
  }
}

// =============================================================================================
// k_pack_pre2
// =============================================================================================
__device__ __forceinline__ int unpack_poses_dev(u64 k0, u64 k1, int* out) {
  int n = 0;
  for (int s = 0; s < kItemPoses; ++s) {
    const unsigned v = (unsigned)(((s < 4 ? k0 : k1) >> (48 - 16 * (s & 3))) & 0xffff);
    if (v != 0xffff) out[n++] = (int)v;
  }
  return n;
}
__device__ __forceinline__ int set_union_dev(const int* a, int na, const int* b, int nb, int* out) {
  int i = 0, j = 0, n = 0;
  while (i < na && j < nb) { if (a[i] < b[j]) out[n++] = a[i++]; else if (b[j] < a[i]) out[n++] = b[j++]; else { out[n++] = a[i++]; ++j; } }
  while (i < na) out[n++] = a[i++];
  while (j < nb) out[n++] = b[j++];
  return n;
}
// slot bytes of the observers obs[r0..r1) among the poses S[0..ns): byte `slot` = rank r of the edge inside its landmark
__device__ __forceinline__ u64 pack_slots_dev(const unsigned short* S, int ns, const int* obs, int r0, int r1) {
  u64 v = ~0ull;
  for (int r = r0; r < r1; ++r) {
    const int o = obs[r];
    int slot = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) slot += (t < ns && (int)S[t] < o) ? 1 : 0;
    v &= ~(0xffull << (8 * slot));
    v |= (u64)(unsigned)r << (8 * slot);
  }
  return v;
}

// The greedy merge of schur_plan.h:plan_window (one thread): consecutive keys whose pose sets unite to <= 8 poses without
// changing the tile count share an item; an item takes at most 64 units.  KF(q, g): word q of the g-th smallest key, CF(g): its units.
template <class KeyF, class CntF>
__device__ void greedy_items(int ng, KeyF KF, CntF CF, DBuild* builds, int* wk, int* out5, const int item_max) {
  int* curX = wk; int* curY = wk + 16; int* gX = wk + 32; int* gY = wk + 40; int* ux = wk + 48; int* uy = wk + 64;
  int ncx = 0, ncy = 0, cur0 = 0, cur1 = 0, nb = 0, n_sym = 0, n_cross = 0, recs_sym = 0, recs_cross = 0;
  bool cur_sym = true;
  auto flush = [&]() {
    for (int base = cur0; base < cur1; base += item_max) {
      DBuild bd;
      bd.base = base; bd.n = min(cur1, base + item_max) - base;
      bd.shape = ncx | (ncy << 8) | ((cur_sym ? 1 : 0) << 16);
      for (int q = 0; q < 8; ++q) { bd.X[q] = q < ncx ? (unsigned short)curX[q] : (unsigned short)0xffff; bd.Y[q] = q < ncy ? (unsigned short)curY[q] : (unsigned short)0xffff; }
      if (cur_sym) { bd.cls_idx = n_sym++; bd.rec_rel = recs_sym; recs_sym += bd.n; } else { bd.cls_idx = n_cross++; bd.rec_rel = recs_cross; recs_cross += bd.n; }
      bd.live = 0; bd.clive = 0;
      builds[nb++] = bd;
    }
    cur0 = cur1;
  };
  int x = 0;
  for (int g = 0; g < ng; ++g) {
    const int x1 = x + CF(g);
    const int ngx = unpack_poses_dev(KF(1, g), KF(2, g), gX), ngy = unpack_poses_dev(KF(3, g), KF(4, g), gY);
    const bool g_sym = KF(0, g) == 0;
    bool merged = false;
    if (cur1 > cur0 && cur_sym == g_sym && (cur1 - cur0) < item_max) {
      const int nux = set_union_dev(curX, ncx, gX, ngx, ux), nuy = set_union_dev(curY, ncy, gY, ngy, uy);
      if (nux <= kItemPoses && nuy <= kItemPoses && tiles_of_dev(nux) == tiles_of_dev(ncx) && tiles_of_dev(nux) == tiles_of_dev(ngx) &&
          tiles_of_dev(nuy) == tiles_of_dev(ncy) && tiles_of_dev(nuy) == tiles_of_dev(ngy)) {
        for (int q = 0; q < nux; ++q) curX[q] = ux[q];
        for (int q = 0; q < nuy; ++q) curY[q] = uy[q];
        ncx = nux; ncy = nuy;
        merged = true;
      }
    }
    if (!merged) {
      if (cur1 > cur0) flush();
      for (int q = 0; q < ngx; ++q) curX[q] = gX[q];
      for (int q = 0; q < ngy; ++q) curY[q] = gY[q];
      ncx = ngx; ncy = ngy;
      cur_sym = g_sym;
    }
    cur1 = x1;
    x = x1;
  }
  if (cur1 > cur0) flush();
  out5[0] = nb; out5[1] = n_sym; out5[2] = n_cross; out5[3] = recs_sym; out5[4] = recs_cross;
}

__global__ __launch_bounds__(NT, 8) void k_pack_pre2(PackArgs a) {
  __shared__ u64 shk[NT];
  __shared__ u64 shK[5 * GL];
  __shared__ int shcnt[GL];
  __shared__ int shord[GL];
  __shared__ int shi[32];
  __shared__ int sh_wk[96];
  __shared__ int sh_n[16];
  __shared__ long long sh_ts;
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const PWin pw = a.win[w];
  PSum& sum = a.sum[w];
  const int L = pw.L, nu = pw.nu, tcap = pw.tcap;
  S1 s = s1_of(a.s1 + pw.s1, L, pw.E, (pw.flags & 2) != 0);
  s1_final(s);
  const S2 z = s2_of(a.s2 + pw.s2, nu, tcap, L);
  if (tid < 16) sh_n[tid] = 0;
  if (tid == 0) sh_ts = 0;
  long long tc_last = clock64();
  int tc_i = 8;
  // ---- F: units and their keys
  for (int j = tid; j < L; j += NT) {
    const int k = s.nfree[j];
    const int* obs = s.sepose + s.lmo[j];
    int u = s.unit_off[j];
    if (k <= kItemPoses) {
      u64 k1, k2;
      pack_poses_dev(obs, 0, k, k1, k2);
      z.uk[0][u] = 0; z.uk[1][u] = k1; z.uk[2][u] = k2; z.uk[3][u] = k1; z.uk[4][u] = k2;
      z.ulm[u] = j; z.uab[u] = 0;
    } else {
      const int np = (k + kItemPoses - 1) / kItemPoses;
      for (int pa = 0; pa < np; ++pa)
        for (int pb = pa; pb < np; ++pb) {
          u64 x0, x1, y0, y1;
          pack_poses_dev(obs, pa * kItemPoses, min(k, pa * kItemPoses + kItemPoses), x0, x1);
          pack_poses_dev(obs, pb * kItemPoses, min(k, pb * kItemPoses + kItemPoses), y0, y1);
          z.uk[0][u] = (pa == pb) ? 0 : 1; z.uk[1][u] = x0; z.uk[2][u] = x1; z.uk[3][u] = y0; z.uk[4][u] = y1;
          z.ulm[u] = j; z.uab[u] = pa | (pb << 8);
          ++u;
        }
    }
  }
  for (int t = tid; t < tcap; t += NT) { z.table[t] = -1; z.gcount[t] = 0; }
  __syncthreads();
  OSH_TC();   // 8: unit keys
  // ---- G: units with equal keys -> one table slot
  for (int i = tid; i < nu; i += NT) {
    u64 k[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) k[q] = z.uk[q][i];
    u64 h = k[0] * 0x9e3779b97f4a7c15ull;
#pragma unroll
    for (int q = 1; q < 5; ++q) { h ^= k[q]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
    int slot = (int)(h & (u64)(tcap - 1));
    for (;;) {
      int cur = ld_l2(&z.table[slot]);
      if (cur < 0) {
        int expected = -1;
        if (__hip_atomic_compare_exchange_strong(&z.table[slot], &expected, i, __ATOMIC_RELAXED, __ATOMIC_RELAXED, OSH_WG)) cur = i;
        else cur = expected;
      }
      if (cur == i) break;
      bool same = true;
#pragma unroll
      for (int q = 0; q < 5; ++q) same &= z.uk[q][cur] == k[q];
      if (same) break;
      slot = (slot + 1) & (tcap - 1);
    }
    z.ugid[i] = slot;
    wg_add(&z.gcount[slot], 1);
  }
  __syncthreads();
  OSH_TC();   // 9: grouping
  // compact the used slots (any order: they are sorted next)
  for (int t = tid; t < tcap; t += NT) {
    const int rep = ld_l2(&z.table[t]);
    if (rep >= 0) {
      const int c = atomicAdd(&sh_n[0], 1);
      z.gslot[c] = t;
      z.gcnt[c] = ld_l2(&z.gcount[t]);
#pragma unroll
      for (int q = 0; q < 5; ++q) z.gk[q][c] = z.uk[q][rep];
    }
  }
  __syncthreads();
  const int ng = sh_n[0];
  int npad = 2;
  while (npad < ng) npad <<= 1;
  const bool in_lds = ng <= GL;
  OSH_TC();   // 10: compaction
  // ---- H: sort the distinct keys (unit_less of schur_plan.h on the key words; keys are distinct, so the order is total)
  int* arr = in_lds ? shord : z.glist;
  for (int t = tid; t < npad; t += NT) arr[t] = t < ng ? t : -1;
  if (in_lds) {
    for (int c = tid; c < ng; c += NT) {
#pragma unroll
      for (int q = 0; q < 5; ++q) shK[q * GL + c] = z.gk[q][c];
    }
    __syncthreads();
    block_bitonic_idx(arr, npad, [&](int x, int y) {
      if (x < 0) return false;
      if (y < 0) return true;
#pragma unroll
      for (int q = 0; q < 5; ++q) { const u64 p = shK[q * GL + x], r = shK[q * GL + y]; if (p != r) return p < r; }
      return false;
    });
  } else {
    __syncthreads();
    block_bitonic_idx(arr, npad, [&](int x, int y) {
      if (x < 0) return false;
      if (y < 0) return true;
#pragma unroll
      for (int q = 0; q < 5; ++q) { const u64 p = z.gk[q][x], r = z.gk[q][y]; if (p != r) return p < r; }
      return false;
    });
  }
  OSH_TC();   // 11: sort of the distinct keys
  // rank of every slot, start of every key's run of units
  for (int g = tid; g < ng; g += NT) { const int c = arr[g]; z.grank[z.gslot[c]] = g; z.gfill[g] = z.gcnt[c]; }
  __syncthreads();
  block_scan_array(z.gfill, ng, shi);
  const bool masks = pw.P <= 64;   // pose sets as 64-bit masks: the merge walk is a few scalar operations per key
  u64* M0 = in_lds ? shK : z.uxs;                 // (uxs / uys / gcount are free until the slot bytes are formed)
  u64* M1 = in_lds ? shK + GL : z.uys;
  u64* M2 = in_lds ? shK + 2 * GL : reinterpret_cast<u64*>(z.uk[0]);   // unit keys of word 0 are no longer read
  if (masks) {
    // per sorted key: masks of its row / column poses, number of units, symmetry
    u64 mx[(GL + NT - 1) / NT], my[(GL + NT - 1) / NT], sc[(GL + NT - 1) / NT];
    if (in_lds) {
#pragma unroll
      for (int r = 0; r < (GL + NT - 1) / NT; ++r) {
        const int g = tid + r * NT;
        mx[r] = my[r] = sc[r] = 0;
        if (g < ng) {
          const int c = arr[g];
          int tmp[8];
          const int nx = unpack_poses_dev(z.gk[1][c], z.gk[2][c], tmp);
          for (int q = 0; q < nx; ++q) mx[r] |= 1ull << tmp[q];
          const int ny = unpack_poses_dev(z.gk[3][c], z.gk[4][c], tmp);
          for (int q = 0; q < ny; ++q) my[r] |= 1ull << tmp[q];
          sc[r] = (u64)(unsigned)z.gcnt[c] | ((z.gk[0][c] == 0) ? (1ull << 63) : 0ull);
        }
      }
      __syncthreads();   // the unsorted keys in shK are dead
#pragma unroll
      for (int r = 0; r < (GL + NT - 1) / NT; ++r) {
        const int g = tid + r * NT;
        if (g < ng) { M0[g] = mx[r]; M1[g] = my[r]; M2[g] = sc[r]; }
      }
    } else {
      for (int g = tid; g < ng; g += NT) {
        const int c = arr[g];
        int tmp[8];
        u64 ax = 0, ay = 0;
        const int nx = unpack_poses_dev(z.gk[1][c], z.gk[2][c], tmp);
        for (int q = 0; q < nx; ++q) ax |= 1ull << tmp[q];
        const int ny = unpack_poses_dev(z.gk[3][c], z.gk[4][c], tmp);
        for (int q = 0; q < ny; ++q) ay |= 1ull << tmp[q];
        M0[g] = ax; M1[g] = ay; M2[g] = (u64)(unsigned)z.gcnt[c] | ((z.gk[0][c] == 0) ? (1ull << 63) : 0ull);
      }
    }
    __syncthreads();
  } else if (in_lds) {
    // the sorted keys for the generic merge walk
    u64 kq[5] = {0, 0, 0, 0, 0};
    int cn = 0;
    if (tid < ng) {
      const int c = arr[tid];
#pragma unroll
      for (int q = 0; q < 5; ++q) kq[q] = z.gk[q][c];
      cn = z.gcnt[c];
    }
    __syncthreads();
    if (tid < ng) {
#pragma unroll
      for (int q = 0; q < 5; ++q) shK[q * GL + tid] = kq[q];
      shcnt[tid] = cn;
    }
    __syncthreads();
  }
  OSH_TC();   // 12: ranks, starts, staging
  // ---- J: items.  The greedy merge of schur_plan.h:plan_window is sequential by nature: thread 0 walks the sorted keys.
  if (masks) {
    // spans of merged keys, written over the consumed inputs (span s is emitted after key s was read)
    if (tid == 0) {
      u64 cx = 0, cy = 0;
      int cur0 = 0, cur1 = 0, ns = 0;
      bool csym = true;
      auto emit = [&]() { M0[ns] = cx; M1[ns] = cy; M2[ns] = (u64)(unsigned)cur0 | ((u64)(unsigned)(cur1 - cur0) << 32) | (csym ? (1ull << 63) : 0ull); ++ns; cur0 = cur1; };
      u64 nx0 = ng ? M0[0] : 0, ny0 = ng ? M1[0] : 0, ns0 = ng ? M2[0] : 0;
      for (int g = 0; g < ng; ++g) {
        const u64 gx = nx0, gy = ny0, sc = ns0;
        if (g + 1 < ng) { nx0 = M0[g + 1]; ny0 = M1[g + 1]; ns0 = M2[g + 1]; }
        const bool gsym = (sc >> 63) != 0;
        bool merged = false;
        if (cur1 > cur0 && csym == gsym && (cur1 - cur0) < a.item_max) {
          const u64 ux = cx | gx, uy = cy | gy;
          const int nux = __popcll(ux), nuy = __popcll(uy);
          if (nux <= kItemPoses && nuy <= kItemPoses && tiles_of_dev(nux) == tiles_of_dev(__popcll(cx)) && tiles_of_dev(nux) == tiles_of_dev(__popcll(gx)) &&
              tiles_of_dev(nuy) == tiles_of_dev(__popcll(cy)) && tiles_of_dev(nuy) == tiles_of_dev(__popcll(gy))) {
            cx = ux; cy = uy;
            merged = true;
          }
        }
        if (!merged) {
          if (cur1 > cur0) emit();
          cx = gx; cy = gy; csym = gsym;
        }
        cur1 += (int)(unsigned)(sc & 0x7fffffffull);
      }
      if (cur1 > cur0) emit();
      sh_n[9] = ns;
    }
    __syncthreads();
    // spans -> items of at most 64 units, in order (symmetric spans come first: key word 0 sorts them so)
    const int ns = sh_n[9];
    int carry_b = 0;
    for (int base = 0; base < ns; base += NT) {
      const int sidx = base + tid;
      int nbs = 0, n = 0, b0 = 0;
      bool sym = false;
      u64 mx = 0, my = 0;
      if (sidx < ns) {
        mx = M0[sidx]; my = M1[sidx];
        const u64 sc = M2[sidx];
        b0 = (int)(unsigned)sc; n = (int)((sc >> 32) & 0x7fffffffull); sym = (sc >> 63) != 0;
        nbs = (n + a.item_max - 1) / a.item_max;
        if (sym) { atomicAdd(&sh_n[2], nbs); atomicAdd(&sh_n[4], n); } else { atomicAdd(&sh_n[3], nbs); atomicAdd(&sh_n[5], n); }
      }
      int tot;
      const int ex = block_excl_scan(nbs, shi, tot);
      if (sidx < ns) {
        DBuild bd;
        int ncx = 0, ncy = 0;
        for (int q = 0; q < 8; ++q) { bd.X[q] = 0xffff; bd.Y[q] = 0xffff; }
        for (u64 m = mx; m; m &= m - 1) bd.X[ncx++] = (unsigned short)(__ffsll((long long)m) - 1);
        for (u64 m = my; m; m &= m - 1) bd.Y[ncy++] = (unsigned short)(__ffsll((long long)m) - 1);
        bd.shape = ncx | (ncy << 8) | ((sym ? 1 : 0) << 16);
        bd.live = 0; bd.clive = 0;
        for (int k = 0; k < nbs; ++k) {
          bd.base = b0 + k * a.item_max; bd.n = min(n - k * a.item_max, a.item_max);
          bd.cls_idx = carry_b + ex + k;     // index among all items; rebased for cross items below
          bd.rec_rel = bd.base;
          z.builds[carry_b + ex + k] = bd;
        }
      }
      carry_b += tot;
    }
    __syncthreads();
    if (tid == 0) sh_n[1] = carry_b;
    // cross items count from their own zero
    const int nsymb = sh_n[2], nsymr = sh_n[4];
    for (int b = tid; b < carry_b; b += NT) {
      if (!((z.builds[b].shape >> 16) & 1)) { z.builds[b].cls_idx -= nsymb; z.builds[b].rec_rel -= nsymr; }
    }
  } else if (tid == 0) {
    if (in_lds) greedy_items(ng, [&](int q, int g) { return shK[q * GL + g]; }, [&](int g) { return shcnt[g]; }, z.builds, sh_wk, sh_n + 1, a.item_max);
    else greedy_items(ng, [&](int q, int g) { return z.gk[q][arr[g]]; }, [&](int g) { return z.gcnt[arr[g]]; }, z.builds, sh_wk, sh_n + 1, a.item_max);
  }
  __syncthreads();
  OSH_TC();   // 13: greedy merge
  // ---- I: units behind their key, creation order (landmark, part pair) kept
  stable_place(nu, [&](int i) { return z.grank[z.ugid[i]]; }, z.gfill,
               [&](int i, int pos) { z.uorder[pos] = i; z.plm[pos] = z.ulm[i]; z.pab[pos] = z.uab[i]; }, shk, shi);   // (landmark, part pair) travel with the unit: read in placement order later
  OSH_TC();   // 14: stable placement of the units
  // ---- landmark renumbering along the owner units (part pair (0,0): one per landmark), then the new offsets
  int carry = 0;
  for (int base = 0; base < nu; base += NT) {
    const int p = base + tid;
    int u = 0, own = 0;
    if (p < nu) { u = z.plm[p]; own = z.pab[p] == 0 ? 1 : 0; }
    int tot;
    const int ex = block_excl_scan(own, shi, tot);
    if (own) { z.old2new[u] = carry + ex; z.perm[carry + ex] = u; }
    carry += tot;
  }
  __syncthreads();
  for (int jn = tid; jn < L; jn += NT) { const int jo = z.perm[jn]; z.lmo_new[jn] = s.lmo[jo + 1] - s.lmo[jo]; }
  if (tid == 0) z.lmo_new[L] = 0;
  __syncthreads();
  block_scan_array(z.lmo_new, L + 1, shi);
  for (int jo = tid; jo < L; jo += NT) z.delta[jo] = z.lmo_new[z.old2new[jo]] - s.lmo[jo];   // new place of a sorted edge = old place + delta of its landmark
  OSH_TC();   // 15: renumbering + offsets
  // ---- chunks of the landmark-major kernels: consecutive landmarks, <= 1024 edges and <= 256 landmarks (lba_pack.h)
  int* nxt = (L <= LCH) ? reinterpret_cast<int*>(shK) : z.nextc;
  for (int j = tid; j < L; j += NT) {
    int lo = j + 1, hi = min(L, j + kBlock);
    const int e0 = z.lmo_new[j];
    while (lo < hi) {   // largest m in [j+1, hi] with m == j+1 or lmo[m] - lmo[j] <= chunk_max_edges(nw)
      const int mid = (lo + hi + 1) >> 1;
      if (z.lmo_new[mid] - e0 <= a.chunk_edges) lo = mid; else hi = mid - 1;
    }
    nxt[j] = lo;
  }
  __syncthreads();
  if (tid == 0) {
    int j = 0, n = 0;
    while (j < L) { const int j1 = nxt[j]; z.chunks[n++] = Chunk{w, j, j1}; j = j1; }
    sh_n[6] = n;
  }
  __syncthreads();
  OSH_TC();   // 16: chunks
  // ---- K: slot bytes of every unit of every item, live pose pairs of the item
  const int nb = sh_n[1];
  for (int b = wv; b < nb; b += NWV) {
    const DBuild bd = z.builds[b];
    const int nx = bd.shape & 0xff, ny = (bd.shape >> 8) & 0xff;
    const bool sym = (bd.shape >> 16) & 1;
    u64 xs = ~0ull, ys = ~0ull;
    if (lane < bd.n) {
      const int p = bd.base + lane;
      const int lm = z.plm[p], ab = z.pab[p];
      const int pa = ab & 0xff, pb = ab >> 8;
      const int k = s.nfree[lm];
      const int* obs = s.sepose + s.lmo[lm];
      xs = pack_slots_dev(bd.X, nx, obs, pa * kItemPoses, min(k, pa * kItemPoses + kItemPoses));
      ys = pack_slots_dev(bd.Y, ny, obs, pb * kItemPoses, min(k, pb * kItemPoses + kItemPoses));
      z.uxs[p] = xs; z.uys[p] = ys;
    }
    unsigned xp = 0, yp = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { if (((xs >> (8 * q)) & 0xff) != kAbsent) xp |= 1u << q; if (((ys >> (8 * q)) & 0xff) != kAbsent) yp |= 1u << q; }
    u64 live = 0;
#pragma unroll
    for (int sa = 0; sa < 8; ++sa)
      if (xp & (1u << sa)) live |= (u64)(yp & (sym ? (0xffu << sa) & 0xffu : 0xffu)) << (8 * sa);
    unsigned lo = (unsigned)live, hi = (unsigned)(live >> 32), cl = sym ? xp : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo |= (unsigned)__shfl_xor((int)lo, o, 64); hi |= (unsigned)__shfl_xor((int)hi, o, 64); cl |= (unsigned)__shfl_xor((int)cl, o, 64); }
    if (lane == 0) {
      z.builds[b].live = ((u64)hi << 32) | lo;
      z.builds[b].clive = (int)cl;
      atomicAdd(&sh_n[7], __popc(lo) + __popc(hi));
      atomicAdd(&sh_n[8], __popc(cl));
      const int tx = tiles_of_dev(nx), ty = tiles_of_dev(ny);
      const long long tiles = sym ? (long long)tx * (tx + 1) / 2 : (long long)tx * ty;
      atomicAdd((u64*)&sh_ts, (u64)(tiles * 6 * (long long)((bd.n + 7) / 8)));
    }
  }
  __syncthreads();
  OSH_TC();   // 17: slot bytes, live pairs
  if (tid == 0) {
    sum.ng = ng; sum.n_builds = sh_n[1]; sum.n_sym = sh_n[2]; sum.n_cross = sh_n[3]; sum.recs_sym = sh_n[4]; sum.recs_cross = sh_n[5];
    sum.n_chunks = sh_n[6]; sum.n_contrib = sh_n[7]; sum.n_ccontrib = sh_n[8]; sum.tile_steps = sh_ts;
    sum.internal = (carry != L) ? 1 : 0;
  }
}

// =============================================================================================
// k_pack_post
// =============================================================================================
__global__ __launch_bounds__(NT) void k_pack_post(PackArgs a) {
  __shared__ u64 shk[NT];
  __shared__ int shi[32];
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const PWin pw = a.win[w];
  const PSum sm = a.sum[w];
  const int L = pw.L, E = pw.E2, P = pw.P, NPw = pw.P + pw.F, nu = pw.nu;   // E: sorted edges
  S1 s = s1_of(a.s1 + pw.s1, L, pw.E, (pw.flags & 2) != 0);
  s1_final(s);
  const S2 z = s2_of(a.s2 + pw.s2, nu, pw.tcap, L);
  const int nb = sm.n_builds, nc = sm.n_contrib, ncc = sm.n_ccontrib;
  const S3 t = s3_of(a.s3 + pw.s3, P, nb, nc, ncc);
  const int nblk = P * (P + 1) / 2;
  PSum& sum = a.sum[w];
  long long tc_last = clock64();
  int tc_i = 18;
  // ---- arena 0: poses, cameras, renumbered landmarks and their offsets
  for (int i = tid; i < NPw * 7; i += NT) a.a_pose[(size_t)pw.pose_off * 7 + i] = a.r_pose[(size_t)pw.pose_off * 7 + i];
  for (int i = tid; i < NPw * 5; i += NT) a.a_cam[(size_t)pw.pose_off * 5 + i] = a.r_cam[(size_t)pw.pose_off * 5 + i];
  for (int i = tid; i < P; i += NT) a.a_fpw[pw.fpose_off + i] = w;
  for (int jn = tid; jn < L; jn += NT) {
    const int jo = z.perm[jn];
#pragma unroll
    for (int k = 0; k < 3; ++k) a.a_pt[((size_t)pw.pt_off + jn) * 3 + k] = a.r_pt[((size_t)pw.pt_off + jo) * 3 + k];
    a.a_lmperm[pw.pt_off + jn] = jo;
    a.ptwin[pw.pt_off + jn] = w;
  }
  for (int jn = tid; jn <= L; jn += NT) a.a_lmoff[pw.lmoff_off + jn] = z.lmo_new[jn];
  // ---- arena 0: edges at their renumbered places
  for (int xo = tid; xo < E; xo += NT) {
    const int jo = s.slm[xo], e = s.order[xo];
    const size_t g = (size_t)pw.edge_off + xo + z.delta[jo];
    a.a_epose[g] = s.sepose[xo];
    a.a_epoint[g] = z.old2new[jo];
    a.a_eorig[g] = e;
    a.a_ekind[g] = s.kind2 ? (unsigned char)s.kind2[xo] : a.r_kind[(size_t)pw.edge_off + e];
    if (a.rec_f32) reinterpret_cast<float4*>(a.a_rec)[g] = reinterpret_cast<const float4*>(a.r_rec)[(size_t)pw.edge_off + e];
    else {
      const double2* src = reinterpret_cast<const double2*>(a.r_rec) + ((size_t)pw.edge_off + e) * 2;
      double2* dst = reinterpret_cast<double2*>(a.a_rec) + g * 2;
      dst[0] = src[0]; dst[1] = src[1];
    }
    if (a.has_rig) {
      // second observation record of the sorted edge: the merged right-camera edge's, else a copy of the edge's own (a lone body
      // edge keeps its observation in the first record too); u v 0 invSigma2, as lba_pack.h writes it
      const int e2 = s.second2 ? s.second2[xo] : -1;
      const double2* src = reinterpret_cast<const double2*>(a.r_rec) + ((size_t)pw.edge_off + (e2 >= 0 ? e2 : e)) * 2;
      const double2 p0 = src[0], p1 = src[1];
      double2* dst = reinterpret_cast<double2*>(a.a_rec2) + g * 2;
      dst[0] = p0; dst[1] = make_double2(0.0, fabs(p1.y));
      a.a_eorig2[g] = e2;
    }
  }
  for (int c = tid; c < sm.n_chunks; c += NT) a.a_chunks[pw.chunk_off + c] = z.chunks[c];
  __syncthreads();
  OSH_TC();   // 18: arena 0
  // ---- contribution counts per block of S / per pose
  for (int k = tid; k <= nblk; k += NT) t.cnt[k] = 0;
  for (int k = tid; k <= P; k += NT) t.ccnt[k] = 0;
  __syncthreads();
  auto blk = [P](int i, int j) { return i * P - i * (i - 1) / 2 + (j - i); };
  for (int b = tid; b <= nb; b += NT) {
    int pl = 0, pc = 0;
    if (b < nb) {
      const DBuild bd = z.builds[b];
      pl = __popcll(bd.live); pc = __popc((unsigned)bd.clive);
      for (int sa = 0; sa < 8; ++sa) {
        if (bd.clive & (1 << sa)) wg_add(&t.ccnt[bd.X[sa] + 1], 1);
        for (int sb = 0; sb < 8; ++sb) if ((bd.live >> (8 * sa + sb)) & 1) wg_add(&t.cnt[blk(bd.X[sa], bd.Y[sb]) + 1], 1);
      }
    }
    t.bpre[b] = pl; t.cpre[b] = pc;
  }
  __syncthreads();
  // inclusive running sums of cnt / ccnt (entry k+1 holds the count of k): cnt[k] = start of block k afterwards
  {
    const int per = (nblk + 1 + NT - 1) / NT;
    const int lo = min(tid * per, nblk + 1), hi = min(lo + per, nblk + 1);
    int sacc = 0;
    for (int i = lo; i < hi; ++i) sacc += ld_l2(t.cnt + i);
    int tot;
    int base = block_excl_scan(sacc, shi, tot);
    for (int i = lo; i < hi; ++i) { base += ld_l2(t.cnt + i); t.cnt[i] = base; }
    __syncthreads();
  }
  {
    const int per = (P + 1 + NT - 1) / NT;
    const int lo = min(tid * per, P + 1), hi = min(lo + per, P + 1);
    int sacc = 0;
    for (int i = lo; i < hi; ++i) sacc += ld_l2(t.ccnt + i);
    int tot;
    int base = block_excl_scan(sacc, shi, tot);
    for (int i = lo; i < hi; ++i) { base += ld_l2(t.ccnt + i); t.ccnt[i] = base; }
    __syncthreads();
  }
  block_scan_array(t.bpre, nb + 1, shi);
  block_scan_array(t.cpre, nb + 1, shi);
  OSH_TC();   // 19: contribution counts + scans
  // ---- block ranges for k_schur_reduce / k_pose_reduce
  for (int i = tid; i < P; i += NT) {
    for (int j = i; j < P; ++j) {
      const int k = blk(i, j);
      a.a_rblk[(size_t)pw.rblk_off + k] = RBlk{w, i | (j << 16), pw.contrib_off + t.cnt[k], t.cnt[k + 1] - t.cnt[k]};
    }
    const int st = pw.ccontrib_off + t.ccnt[i], cn = t.ccnt[i + 1] - t.ccnt[i];
    a.a_rblk[(size_t)pw.rblk_off + nblk + i] = RBlk{w, (int)(0xffff0000u | (unsigned)i), st, cn};
    a.a_crange[pw.fpose_off + i] = I2{st, cn};
  }
  // ---- items, records, the live entries of every item
  for (int b = wv; b < nb; b += NWV) {
    const DBuild bd = z.builds[b];
    const bool sym = (bd.shape >> 16) & 1;
    const size_t it = (size_t)(sym ? pw.sym_item_off : pw.cross_item_off) + bd.cls_idx;
    const int rec0 = (sym ? pw.sym_rec_off : pw.cross_rec_off) + bd.rec_rel;
    if (lane == 0) a.a_items[it] = SItem{w, rec0, bd.n, bd.shape};
    a.a_spair[it * 64 + lane] = -1;
    if (lane < 8) {
      a.a_scslot[it * 8 + lane] = -1;
      a.a_posex[it * 8 + lane] = bd.X[lane] == 0xffff ? -1 : (int)bd.X[lane];
      a.a_posey[it * 8 + lane] = bd.Y[lane] == 0xffff ? -1 : (int)bd.Y[lane];
    }
    if (lane < bd.n) {
      const int p = bd.base + lane;
      const int lm = z.plm[p], ab = z.pab[p];
      const int pa = ab & 0xff, pb = ab >> 8;
      const int k = s.nfree[lm];
      const int a0 = pa * kItemPoses, a1 = min(k, a0 + kItemPoses);
      const int jn = z.old2new[lm];
      const u64 xs = z.uxs[p], ys = z.uys[p];
      SRec r;
      r.lm = jn; r.e_first = z.lmo_new[jn];
      r.x_lo = (unsigned)xs; r.x_hi = (unsigned)(xs >> 32); r.y_lo = (unsigned)ys; r.y_hi = (unsigned)(ys >> 32);
      r.flags = ((pa == 0 && pb == 0) ? (1 | (min(k, kItemPoses) << 8)) : 0) | (a0 << 16) | ((a1 - a0) << 24);
      r.pad = z.lmo_new[jn + 1] - z.lmo_new[jn];
      a.a_recs[(size_t)rec0 + lane] = r;
    }
    if ((bd.live >> lane) & 1) t.ent[t.bpre[b] + __popcll(bd.live & ((1ull << lane) - 1ull))] = (b << 6) | lane;
    if (lane < 8 && ((bd.clive >> lane) & 1)) t.cent[t.cpre[b] + __popc((unsigned)bd.clive & ((1u << lane) - 1u))] = (b << 3) | lane;
  }
  __syncthreads();
  OSH_TC();   // 20: block ranges, items, records
  // ---- contribution slots: the contributions of one block of S are contiguous, in item order (schur_plan.h)
  stable_place(nc, [&](int q) { const int en = t.ent[q]; const DBuild* bd = z.builds + (en >> 6); return blk(bd->X[(en >> 3) & 7], bd->Y[en & 7]); }, t.cnt,
               [&](int q, int pos) {
                 const int en = t.ent[q];
                 const DBuild* bd = z.builds + (en >> 6);
                 const bool sym = (bd->shape >> 16) & 1;
                 const size_t it = (size_t)(sym ? pw.sym_item_off : pw.cross_item_off) + bd->cls_idx;
                 a.a_spair[it * 64 + (en & 63)] = pw.contrib_off + pos;
               }, shk, shi);
  stable_place(ncc, [&](int q) { const int en = t.cent[q]; return (int)z.builds[en >> 3].X[en & 7]; }, t.ccnt,
               [&](int q, int pos) {
                 const int en = t.cent[q];
                 const DBuild* bd = z.builds + (en >> 3);
                 const size_t it = (size_t)pw.sym_item_off + bd->cls_idx;   // only symmetric items own rhs slots
                 a.a_scslot[it * 8 + (en & 7)] = pw.ccontrib_off + pos;
               }, shk, shi);
  OSH_TC();   // 21: contribution slots
#undef OSH_TC
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace dpack

using namespace dpack;

bool device_pack_supported(int nw, const osh_lba_problem* pr) {
  (void)nw; (void)pr;
  return true;   // (fisheye-rig batches included since the merged left / right pairs are compacted on the device)
}

namespace {

template <class F>
void parallel_windows(int nw, int n_threads, F f) {
  n_threads = std::max(1, std::min(n_threads, nw));
  if (n_threads == 1) { for (int w = 0; w < nw; ++w) f(w); return; }
  std::atomic<int> next{0};
  auto worker = [&]() { for (int w = next.fetch_add(1); w < nw; w = next.fetch_add(1)) f(w); };
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
  worker();
  for (std::thread& t : pool) t.join();
}

#define DP_HIP(call)                                                                                              \
  do {                                                                                                            \
    hipError_t _e = (call);                                                                                       \
    if (_e != hipSuccess) {                                                                                       \
      pb.err = OSH_ERR_DEVICE;                                                                                    \
      std::snprintf(pb.msg, sizeof(pb.msg), "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
      return pb.err;                                                                                              \
    }                                                                                                             \
  } while (0)

}  // namespace

int device_pack_batch(DevPackState& st, hipStream_t s, int nw, const osh_lba_problem* pr, int n_threads, PackedBatch& pb, DevBuf* d_arena, DevBuf& d_ptwin) {
  const auto t0 = std::chrono::steady_clock::now();
  if (pack_describe(nw, pr, pb) != OSH_OK) return pb.err;
  auto fail = [&](int code, const char* fmt, auto... a) {
    pb.err = code;
    if constexpr (sizeof...(a) == 0) std::snprintf(pb.msg, sizeof(pb.msg), "%s", fmt); else std::snprintf(pb.msg, sizeof(pb.msg), fmt, a...);
    return code;
  };
  const size_t NP = pb.NP, NL = pb.NL, NE = pb.NE;
  // ---- raw staging: poses (normalised as g2o::SE3Quat does), cameras, landmarks, edges in the caller's order; observation
  // records narrowed to float32 when every value is one (checked on the way; otherwise the batch is staged again with doubles)
  size_t o_pose = 0, o_cam = 0, o_pt = 0, o_ep = 0, o_el = 0, o_kind = 0, o_rec = 0, raw_bytes = 0;
  bool f32 = !pb.has_rig;   // a rig batch keeps double records (lba_pack.h)
  for (int attempt = 0; attempt < 2; ++attempt) {
    size_t o = 0;
    auto put = [&](size_t& off, size_t b) { off = o; o += al256(std::max<size_t>(b, 8)); };
    put(o_pose, NP * 56); put(o_cam, NP * 40); put(o_pt, NL * 24); put(o_ep, NE * 4); put(o_el, NE * 4); put(o_kind, NE); put(o_rec, NE * (f32 ? 16 : 32));
    raw_bytes = o;
    unsigned char* h = static_cast<unsigned char*>(st.h_raw.reserve(raw_bytes));
    if (!h) return fail(OSH_ERR_DEVICE, "cannot allocate %zu bytes of staging memory", raw_bytes);
    std::atomic<int> inexact{0};
    auto stage_window = [&](int w) {
      const osh_lba_problem& p = pr[w];
      const WinDesc& d = pb.win[w];
      const int NPw = p.n_free + p.n_fixed;
      double* hp = reinterpret_cast<double*>(h + o_pose) + (size_t)d.pose_off * 7;
      for (int i = 0; i < NPw; ++i) pack_pose(p.pose_qt + 7 * (size_t)i, hp + 7 * (size_t)i);
      if (NPw) std::memcpy(reinterpret_cast<double*>(h + o_cam) + (size_t)d.pose_off * 5, p.pose_cam, (size_t)NPw * 40);
      if (p.n_points) std::memcpy(reinterpret_cast<double*>(h + o_pt) + (size_t)d.pt_off * 3, p.points, (size_t)p.n_points * 24);
      const size_t E = (size_t)p.n_edges;
      if (!E) return;
      std::memcpy(reinterpret_cast<int*>(h + o_ep) + d.edge_off, p.edge_pose, E * 4);
      std::memcpy(reinterpret_cast<int*>(h + o_el) + d.edge_off, p.edge_point, E * 4);
      std::memcpy(h + o_kind + d.edge_off, p.edge_kind, E);
      // the sign of the information carries the edge kind for the pinhole kernels (negative = monocular), as in lba_pack.h
      if (f32) {
        float* r = reinterpret_cast<float*>(h + o_rec) + (size_t)d.edge_off * 4;
        bool exact = true;
        for (size_t e = 0; e < E; ++e) {
          const double o0 = p.edge_obs[3 * e], o1 = p.edge_obs[3 * e + 1], o2 = p.edge_obs[3 * e + 2];
          const double inf = (p.edge_kind[e] == OSH_EDGE_STEREO) ? p.edge_info[e] : -p.edge_info[e];
          const float f0 = (float)o0, f1 = (float)o1, f2 = (float)o2, f3 = (float)inf;
          r[4 * e] = f0; r[4 * e + 1] = f1; r[4 * e + 2] = f2; r[4 * e + 3] = f3;
          exact &= ((double)f0 == o0) & ((double)f1 == o1) & ((double)f2 == o2) & ((double)f3 == inf);
        }
        if (!exact) inexact.store(1, std::memory_order_relaxed);
      } else {
        double* r = reinterpret_cast<double*>(h + o_rec) + (size_t)d.edge_off * 4;
        for (size_t e = 0; e < E; ++e) {
          r[4 * e] = p.edge_obs[3 * e]; r[4 * e + 1] = p.edge_obs[3 * e + 1]; r[4 * e + 2] = p.edge_obs[3 * e + 2];
          r[4 * e + 3] = (p.edge_kind[e] == OSH_EDGE_STEREO) ? p.edge_info[e] : -p.edge_info[e];
        }
      }
    };
    // The staging pass and the host-to-device copy overlap: the windows are staged slice by slice (worker threads, in order), and a
    // slice's sub-ranges of the seven raw arrays leave as asynchronous copies as soon as its last window is staged -- staged first and
    // copied afterwards, an upload of 512 windows spent 13 ms staging and then 25-31 ms copying 1.09 GB while the pipeline's upload stage
    // had become as long as its optimize stage.
    if (st.d_raw.reserve(raw_bytes) != OSH_OK) return fail(OSH_ERR_DEVICE, "%s", get_error());
    if (st.timing) for (hipEvent_t& e : st.ev) if (!e) DP_HIP(hipEventCreate(&e));
    if (st.timing) (void)hipEventRecord(st.ev[0], s);
    const int n_slices = (nw >= 128 && n_threads > 1) ? std::min(8, nw / 32) : 1;
    hipError_t copy_err = hipSuccess;
    auto copy_slice = [&](int w0, int w1) {
      const WinDesc& a0 = pb.win[w0];
      const size_t p0 = (size_t)a0.pose_off, l0 = (size_t)a0.pt_off, e0 = (size_t)a0.edge_off;
      const size_t p1 = w1 < nw ? (size_t)pb.win[w1].pose_off : NP, l1 = w1 < nw ? (size_t)pb.win[w1].pt_off : NL, e1 = w1 < nw ? (size_t)pb.win[w1].edge_off : NE;
      unsigned char* dst = static_cast<unsigned char*>(st.d_raw.p);
      auto cp = [&](size_t off, size_t unit, size_t a, size_t b) {
        if (b > a && copy_err == hipSuccess) copy_err = hipMemcpyAsync(dst + off + a * unit, h + off + a * unit, (b - a) * unit, hipMemcpyHostToDevice, s);
      };
      cp(o_pose, 56, p0, p1); cp(o_cam, 40, p0, p1); cp(o_pt, 24, l0, l1); cp(o_ep, 4, e0, e1); cp(o_el, 4, e0, e1); cp(o_kind, 1, e0, e1);
      cp(o_rec, f32 ? 16 : 32, e0, e1);
    };
    if (n_slices == 1) {
      parallel_windows(nw, n_threads, stage_window);
      if (!(f32 && inexact.load())) copy_slice(0, nw);
    } else {
      std::vector<int> lo(n_slices + 1);
      for (int k = 0; k <= n_slices; ++k) lo[k] = (int)((long long)k * nw / n_slices);
      std::vector<unsigned char> slice_of(nw);
      for (int k = 0; k < n_slices; ++k) for (int w = lo[k]; w < lo[k + 1]; ++w) slice_of[w] = (unsigned char)k;
      std::atomic<int> done[8];
      for (int k = 0; k < 8; ++k) done[k].store(0);
      std::atomic<int> next{0};
      auto worker = [&]() {
        for (int w = next.fetch_add(1); w < nw; w = next.fetch_add(1)) { stage_window(w); done[slice_of[w]].fetch_add(1, std::memory_order_release); }
      };
      std::vector<std::thread> pool;
      for (int t = 0; t < n_threads - 1; ++t) pool.emplace_back(worker);
      for (int k = 0; k < n_slices; ++k) {
        while (done[k].load(std::memory_order_acquire) < lo[k + 1] - lo[k]) std::this_thread::yield();
        if (f32 && inexact.load()) continue;        // the batch is staged again with double records: nothing more to send
        copy_slice(lo[k], lo[k + 1]);
      }
      for (std::thread& t : pool) t.join();
    }
    if (copy_err != hipSuccess) return fail(OSH_ERR_DEVICE, "host-to-device copy of the staged windows: %s", hipGetErrorString(copy_err));
    if (f32 && inexact.load()) (void)hipStreamSynchronize(s);   // copies of the float32 staging may still be reading the buffer
    if (!f32 || !inexact.load()) break;
    f32 = false;
  }
  pb.rec_f32 = f32;
  const auto t1 = std::chrono::steady_clock::now();
  st.host_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();

  st.raw_bytes = raw_bytes;
  auto mark = [&](int k) { if (st.timing) (void)hipEventRecord(st.ev[k], s); };
  mark(1);   // (ev[0] was recorded before the first slice's copies)

  // ---- control block: window descriptors of the packer + summaries
  const size_t ctl_bytes = al256((size_t)nw * sizeof(PWin)) + al256((size_t)nw * sizeof(PSum));
  unsigned char* hc = static_cast<unsigned char*>(st.h_ctl.reserve(ctl_bytes));
  if (!hc) return fail(OSH_ERR_DEVICE, "cannot allocate pinned control block");
  if (st.d_ctl.reserve(ctl_bytes) != OSH_OK) return fail(OSH_ERR_DEVICE, "%s", get_error());
  PWin* hw = reinterpret_cast<PWin*>(hc);
  PSum* hs = reinterpret_cast<PSum*>(hc + al256((size_t)nw * sizeof(PWin)));
  PWin* dw = reinterpret_cast<PWin*>(st.d_ctl.p);
  PSum* ds = reinterpret_cast<PSum*>(static_cast<unsigned char*>(st.d_ctl.p) + al256((size_t)nw * sizeof(PWin)));
  size_t s1_total = 0;
  for (int w = 0; w < nw; ++w) {
    const WinDesc& d = pb.win[w];
    PWin q{};
    q.P = d.P; q.F = d.F; q.L = d.L; q.E = d.in_edges; q.flags = (d.kb8_on ? 1 : 0) | (d.rig_on ? 2 : 0);
    q.pose_off = d.pose_off; q.fpose_off = d.fpose_off; q.pt_off = d.pt_off; q.edge_off = d.edge_off; q.lmoff_off = d.lmoff_off;
    q.s1 = (long long)s1_total;
    s1_total += (s1_ints(d.L, d.in_edges, d.rig_on != 0) + 63) & ~(size_t)63;
    hw[w] = q;
  }
  if (st.d_s1.reserve(s1_total * 4) != OSH_OK) return fail(OSH_ERR_DEVICE, "%s", get_error());
  DP_HIP(hipMemcpyAsync(dw, hw, (size_t)nw * sizeof(PWin), hipMemcpyHostToDevice, s));
  PackArgs a{};
  a.win = dw; a.sum = ds;
  unsigned char* dr = static_cast<unsigned char*>(st.d_raw.p);
  a.r_pose = reinterpret_cast<const double*>(dr + o_pose); a.r_cam = reinterpret_cast<const double*>(dr + o_cam); a.r_pt = reinterpret_cast<const double*>(dr + o_pt);
  a.r_epose = reinterpret_cast<const int*>(dr + o_ep); a.r_epoint = reinterpret_cast<const int*>(dr + o_el); a.r_kind = dr + o_kind; a.r_rec = dr + o_rec;
  a.rec_f32 = f32 ? 1 : 0;
  a.item_max = item_max_lm(nw);
  a.chunk_edges = chunk_max_edges(nw);
  a.s1 = st.d_s1.as<int>();
  mark(2);
  hipLaunchKernelGGL(k_pack_pre1, dim3((unsigned)nw), dim3(NT), 0, s, a);
  mark(3);
  DP_HIP(hipGetLastError());
  DP_HIP(hipMemcpyAsync(hs, ds, (size_t)nw * sizeof(PSum), hipMemcpyDeviceToHost, s));
  DP_HIP(hipStreamSynchronize(s));
  // ---- round trip 1: errors in the order lba_pack.h reports them, sizes of the plan scratch
  for (int w = 0; w < nw; ++w) {
    const PSum& m = hs[w];
    if (m.err_a != 0xffffffffu) {
      const int e = (int)(m.err_a >> 2), cls = (int)(m.err_a & 3);
      if (cls == 0) return fail(OSH_ERR_INVALID, "window %d edge %d: index or kind out of range", w, e);
      if (cls == 1) return fail(OSH_ERR_UNSUPPORTED, "window %d: a KannalaBrandt8 window takes monocular and body edges only (edge %d is a rectified-stereo edge)", w, e);
      return fail(OSH_ERR_INVALID, "window %d edge %d: a body edge (EdgeSE3ProjectXYZToBody) needs kb8, cam2 and trl", w, e);
    }
    if (m.err_d != ~0ull)
      return fail(OSH_ERR_UNSUPPORTED, "window %d: landmark %d is observed twice by pose %d with edge kinds that do not form a "
                  "fisheye-rig pair (left EdgeSE3ProjectXYZ + right EdgeSE3ProjectXYZToBody)", w, (int)(m.err_d >> 32), (int)(unsigned)m.err_d);
    if (m.err_k) return fail(OSH_ERR_UNSUPPORTED, "window %d: a landmark has more than 254 optimisable observers", w);
  }
  size_t s2_total = 0;
  for (int w = 0; w < nw; ++w) {
    PWin& q = hw[w];
    q.nu = hs[w].nu;
    q.E2 = hs[w].E2;
    int tcap = 64;
    while ((long long)tcap < 2ll * q.nu) tcap <<= 1;
    q.tcap = tcap;
    q.s2 = (long long)s2_total;
    s2_total += al256(s2_bytes(q.nu, tcap, q.L));
    pb.win[w].E = q.E2;
    pb.pair_blocks += hs[w].pair_blocks;
  }
  if (st.d_s2.reserve(s2_total) != OSH_OK) return fail(OSH_ERR_DEVICE, "%s", get_error());
  a.s2 = st.d_s2.as<unsigned char>();
  DP_HIP(hipMemcpyAsync(dw, hw, (size_t)nw * sizeof(PWin), hipMemcpyHostToDevice, s));
  mark(4);
  hipLaunchKernelGGL(k_pack_pre2, dim3((unsigned)nw), dim3(NT), 0, s, a);
  mark(5);
  DP_HIP(hipGetLastError());
  DP_HIP(hipMemcpyAsync(hs, ds, (size_t)nw * sizeof(PSum), hipMemcpyDeviceToHost, s));
  DP_HIP(hipStreamSynchronize(s));
  // ---- round trip 2: offsets of the plan sections (the sums of lba_pack.h's phase 2)
  size_t chunk = 0, sym_item = 0, cross_item = 0, sym_rec = 0, cross_rec = 0, rblk = 0, contrib = 0, ccontrib = 0, s3_total = 0;
  for (int w = 0; w < nw; ++w) {
    const PSum& m = hs[w];
    if (m.internal) return fail(OSH_ERR_DEVICE, "window %d: device plan owns a wrong number of landmarks", w);
    chunk += m.n_chunks; sym_item += m.n_sym; cross_item += m.n_cross; sym_rec += m.recs_sym; cross_rec += m.recs_cross;
    rblk += (size_t)pb.win[w].P * (pb.win[w].P + 1) / 2 + pb.win[w].P; contrib += m.n_contrib; ccontrib += m.n_ccontrib;
    pb.tile_steps += m.tile_steps;
  }
  pb.n_chunks = chunk; pb.n_sym = sym_item; pb.n_items = sym_item + cross_item; pb.n_recs = sym_rec + cross_rec;
  pb.n_rblk = rblk; pb.n_contrib = contrib; pb.n_ccontrib = ccontrib;
  if (contrib > 0x7fffff00u / 36 * 16 || pb.n_recs > 0x7fffff00u) return fail(OSH_ERR_UNSUPPORTED, "batch too large for 32-bit contribution offsets");
  {
    size_t c = 0, si = 0, ci = 0, sr = 0, cr = 0, rb = 0, co = 0, cc = 0;
    for (int w = 0; w < nw; ++w) {
      const PSum& m = hs[w];
      PWin& q = hw[w];
      WinDesc& d = pb.win[w];
      q.chunk_off = (int)c; q.sym_item_off = (int)si; q.cross_item_off = (int)(sym_item + ci); q.sym_rec_off = (int)sr; q.cross_rec_off = (int)(sym_rec + cr);
      q.rblk_off = (int)rb; q.contrib_off = (int)co; q.ccontrib_off = (int)cc;
      q.s3 = (long long)s3_total;
      s3_total += (s3_ints(q.P, m.n_builds, m.n_contrib, m.n_ccontrib) + 63) & ~(size_t)63;
      d.chunk_off = (int)c; d.n_chunks = m.n_chunks; d.sitem_off = (int)si; d.n_sitems = m.n_sym;
      c += m.n_chunks; si += m.n_sym; ci += m.n_cross; sr += m.recs_sym; cr += m.recs_cross;
      rb += (size_t)d.P * (d.P + 1) / 2 + d.P; co += m.n_contrib; cc += m.n_ccontrib;
    }
  }
  if (st.d_s3.reserve(s3_total * 4) != OSH_OK) return fail(OSH_ERR_DEVICE, "%s", get_error());
  a.s3 = st.d_s3.as<int>();
  pack_layout0(pb);
  pack_layout1(pb);
  pb.arena[0] = pb.arena[1] = nullptr;
  for (int k = 0; k < 2; ++k) if (d_arena[k].reserve(pb.arena_bytes[k]) != OSH_OK) return fail(OSH_ERR_DEVICE, "%s", get_error());
  if (d_ptwin.reserve(std::max<size_t>(NL * 4, 8)) != OSH_OK) return fail(OSH_ERR_DEVICE, "%s", get_error());
  auto sec = [&](int sidx) { return static_cast<unsigned char*>(d_arena[pb.sec_arena(sidx)].p) + pb.sec_off(sidx); };
  a.a_pose = reinterpret_cast<double*>(sec(PackedBatch::POSE)); a.a_cam = reinterpret_cast<double*>(sec(PackedBatch::CAM)); a.a_pt = reinterpret_cast<double*>(sec(PackedBatch::PT));
  a.a_rec = sec(PackedBatch::EREC); a.a_epose = reinterpret_cast<int*>(sec(PackedBatch::EPOSE)); a.a_epoint = reinterpret_cast<int*>(sec(PackedBatch::EPOINT));
  a.a_eorig = reinterpret_cast<int*>(sec(PackedBatch::EORIG)); a.a_lmoff = reinterpret_cast<int*>(sec(PackedBatch::LMOFF)); a.a_lmperm = reinterpret_cast<int*>(sec(PackedBatch::LMPERM));
  a.a_fpw = reinterpret_cast<int*>(sec(PackedBatch::FPW)); a.a_ekind = sec(PackedBatch::EKIND);
  a.has_rig = pb.has_rig ? 1 : 0;
  a.a_rec2 = reinterpret_cast<double*>(sec(PackedBatch::EREC2)); a.a_eorig2 = reinterpret_cast<int*>(sec(PackedBatch::EORIG2));
  a.a_chunks = reinterpret_cast<Chunk*>(sec(PackedBatch::CHUNKS)); a.a_items = reinterpret_cast<SItem*>(sec(PackedBatch::ITEMS)); a.a_recs = reinterpret_cast<SRec*>(sec(PackedBatch::RECS));
  a.a_spair = reinterpret_cast<int*>(sec(PackedBatch::SPAIR)); a.a_scslot = reinterpret_cast<int*>(sec(PackedBatch::SCSLOT));
  a.a_posex = reinterpret_cast<int*>(sec(PackedBatch::POSEX)); a.a_posey = reinterpret_cast<int*>(sec(PackedBatch::POSEY));
  a.a_rblk = reinterpret_cast<RBlk*>(sec(PackedBatch::RBLK)); a.a_crange = reinterpret_cast<I2*>(sec(PackedBatch::CRANGE));
  a.ptwin = d_ptwin.as<int>();
  DP_HIP(hipMemcpyAsync(dw, hw, (size_t)nw * sizeof(PWin), hipMemcpyHostToDevice, s));
  DP_HIP(hipMemcpyAsync(sec(PackedBatch::WIN), pb.win.data(), (size_t)nw * sizeof(WinDesc), hipMemcpyHostToDevice, s));
  mark(6);
  hipLaunchKernelGGL(k_pack_post, dim3((unsigned)nw), dim3(NT), 0, s, a);
  mark(7);
  DP_HIP(hipGetLastError());
  DP_HIP(hipStreamSynchronize(s));   // pb.win (pageable) and the staging are free again
  if (st.timing) {
    for (int k = 0; k < 4; ++k) { float ms = 0.f; if (hipEventElapsedTime(&ms, st.ev[2 * k], st.ev[2 * k + 1]) == hipSuccess) st.ev_ms[k] = ms; }
    DP_HIP(hipMemcpy(hs, ds, (size_t)nw * sizeof(PSum), hipMemcpyDeviceToHost));
    for (int k = 0; k < 24; ++k) { double acc = 0; for (int w = 0; w < nw; ++w) acc += (double)hs[w].cyc[k]; st.cyc_mean[k] = acc / nw; }
  }
  st.device_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
  return OSH_OK;
}

int device_pack_compare(DevPackState& st, hipStream_t s, int nw, const osh_lba_problem* pr, int64_t stats[4]) {
  PackedBatch pd, ph;
  DevBuf d_arena[2], d_ptwin;
  if (device_pack_batch(st, s, nw, pr, default_pack_threads(nw), pd, d_arena, d_ptwin) != OSH_OK) { set_error("device packer: %s", pd.msg); return pd.err; }
  std::vector<unsigned char> hmem[2];
  if (pack_batch(nw, pr, [&](int which, size_t bytes) { hmem[which].resize(bytes); return (void*)hmem[which].data(); }, default_pack_threads(nw), ph) != OSH_OK) {
    set_error("host packer: %s", ph.msg);
    return ph.err;
  }
#define CMP(field) if (pd.field != ph.field) { set_error("packers disagree on " #field ": device %lld, host %lld", (long long)pd.field, (long long)ph.field); return OSH_ERR_DEVICE; }
  CMP(NP) CMP(NFP) CMP(NL) CMP(NE) CMP(NLO) CMP(NOUT) CMP(S_total) CMP(n_chunks) CMP(n_items) CMP(n_sym) CMP(n_recs) CMP(n_rblk) CMP(n_contrib) CMP(n_ccontrib)
  CMP(tile_steps) CMP(pair_blocks) CMP(n_max) CMP(np_max) CMP(has_kb8) CMP(rec_f32) CMP(arena_bytes[0]) CMP(arena_bytes[1])
#undef CMP
  static const char* names[PackedBatch::SEC_COUNT] = {"POSE", "CAM", "PT", "EREC", "EREC2", "EPOSE", "EPOINT", "EORIG", "EORIG2", "LMOFF", "LMPERM", "FPW", "EKIND",
                                                      "WIN", "CHUNKS", "ITEMS", "RECS", "SPAIR", "SCSLOT", "POSEX", "POSEY", "RBLK", "CRANGE"};
  int64_t bytes = 0, nsec = 0;
  std::vector<unsigned char> dev;
  for (int k = 0; k < PackedBatch::SEC_COUNT; ++k) {
    if (pd.off[k] != ph.off[k] || pd.bytes[k] != ph.bytes[k]) { set_error("section %s: offsets differ", names[k]); return OSH_ERR_DEVICE; }
    if (!ph.bytes[k]) continue;
    dev.resize(ph.bytes[k]);
    if (hipMemcpy(dev.data(), static_cast<unsigned char*>(d_arena[ph.sec_arena(k)].p) + ph.off[k], ph.bytes[k], hipMemcpyDeviceToHost) != hipSuccess) {
      set_error("section %s: copy failed", names[k]);
      return OSH_ERR_DEVICE;
    }
    const unsigned char* ref = ph.arena[ph.sec_arena(k)] + ph.off[k];
    if (k == PackedBatch::WIN) {
      // struct padding is not defined: compare field by field
      for (int w = 0; w < nw; ++w) {
        const WinDesc &x = reinterpret_cast<const WinDesc*>(dev.data())[w], &y = reinterpret_cast<const WinDesc*>(ref)[w];
        const bool eq = x.P == y.P && x.F == y.F && x.L == y.L && x.E == y.E && x.pose_off == y.pose_off && x.fpose_off == y.fpose_off && x.pt_off == y.pt_off &&
                        x.edge_off == y.edge_off && x.lmoff_off == y.lmoff_off && x.chunk_off == y.chunk_off && x.n_chunks == y.n_chunks && x.sitem_off == y.sitem_off &&
                        x.n_sitems == y.n_sitems && x.n == y.n && x.max_iter == y.max_iter && x.kb8_on == y.kb8_on && x.rig_on == y.rig_on && x.in_edges == y.in_edges &&
                        x.out_off == y.out_off && x.S_off == y.S_off && x.huber_mono == y.huber_mono && x.huber_stereo == y.huber_stereo && x.lambda_init == y.lambda_init &&
                        !std::memcmp(x.kb8, y.kb8, sizeof(x.kb8)) && !std::memcmp(x.cam2, y.cam2, sizeof(x.cam2)) && !std::memcmp(x.trl, y.trl, sizeof(x.trl));
        if (!eq) { set_error("section WIN: descriptor of window %d differs", w); return OSH_ERR_DEVICE; }
      }
    } else if (k == PackedBatch::RECS) {
      for (size_t r = 0; r < ph.n_recs; ++r)
        if (std::memcmp(dev.data() + r * sizeof(SRec), ref + r * sizeof(SRec), sizeof(SRec))) {
          const SRec &x = reinterpret_cast<const SRec*>(dev.data())[r], &y = reinterpret_cast<const SRec*>(ref)[r];
          set_error("section RECS: record %zu differs: device {lm %d e %d x %08x%08x y %08x%08x flags %x pad %d} host {lm %d e %d x %08x%08x y %08x%08x flags %x pad %d}", r,
                    x.lm, x.e_first, x.x_hi, x.x_lo, x.y_hi, x.y_lo, x.flags, x.pad, y.lm, y.e_first, y.x_hi, y.x_lo, y.y_hi, y.y_lo, y.flags, y.pad);
          return OSH_ERR_DEVICE;
        }
    } else if (ph.has_rig && (k == PackedBatch::EREC || k == PackedBatch::EREC2 || k == PackedBatch::EPOSE || k == PackedBatch::EPOINT || k == PackedBatch::EORIG ||
                              k == PackedBatch::EORIG2 || k == PackedBatch::EKIND)) {
      // a window with merged pairs leaves the tail of its edge range unused: compare the sorted edges of every window only
      const size_t el = k == PackedBatch::EKIND ? 1 : (k == PackedBatch::EREC || k == PackedBatch::EREC2) ? 32 : 4;
      for (int w = 0; w < nw; ++w) {
        const size_t o = (size_t)ph.win[w].edge_off * el, nb2 = (size_t)ph.win[w].E * el;
        if (nb2 && std::memcmp(dev.data() + o, ref + o, nb2)) {
          size_t at = 0;
          while (dev[o + at] == ref[o + at]) ++at;
          set_error("section %s differs in window %d at sorted edge %zu", names[k], w, at / el);
          return OSH_ERR_DEVICE;
        }
      }
    } else if (std::memcmp(dev.data(), ref, ph.bytes[k])) {
      size_t at = 0;
      while (dev[at] == ref[at]) ++at;
      set_error("section %s differs at byte %zu of %zu (int index %zu: device %d, host %d)", names[k], at, ph.bytes[k], at / 4,
                reinterpret_cast<const int*>(dev.data())[at / 4], reinterpret_cast<const int*>(ref)[at / 4]);
      return OSH_ERR_DEVICE;
    }
    bytes += (int64_t)ph.bytes[k];
    ++nsec;
  }
  // window of every landmark (k_gather_out)
  if (ph.NL) {
    std::vector<int> pw(ph.NL);
    if (hipMemcpy(pw.data(), d_ptwin.p, ph.NL * 4, hipMemcpyDeviceToHost) != hipSuccess) { set_error("ptwin: copy failed"); return OSH_ERR_DEVICE; }
    for (int w = 0; w < nw; ++w)
      for (int j = 0; j < ph.win[w].L; ++j)
        if (pw[(size_t)ph.win[w].pt_off + j] != w) { set_error("ptwin differs in window %d", w); return OSH_ERR_DEVICE; }
  }
  if (stats) { stats[0] = bytes; stats[1] = nsec; stats[2] = (int64_t)ph.n_items; stats[3] = (int64_t)ph.n_recs; }
  return OSH_OK;
}

}  // namespace osh
