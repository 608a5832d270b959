// schur_plan_check.cpp -- host-only self check of the Schur work plan (schur_plan.h).
// Builds the plan of one window exactly as osh_lba_upload does and verifies that every pair of
// optimisable observers of every landmark is covered by exactly one item record, that every
// contribution slot is written by exactly one (item, pose pair) and lies in the range of its block
// of S, and that every landmark has exactly one dinv owner.  Needs no GPU.
#include "common.h"
#include "schur_plan.h"
#include <map>
#include <set>

using namespace osh;

extern "C" int osh_lba_schur_plan_stats(const osh_lba_problem* p, int64_t stats[8]) {
  if (!p || !stats) { set_error("osh_lba_schur_plan_stats: NULL argument"); return OSH_ERR_INVALID; }
  const int P = p->n_free, L = p->n_points;
  // optimisable-pose edges of every landmark, poses ascending (what upload's sort produces)
  std::vector<std::vector<int>> obs(L);
  for (int e = 0; e < p->n_edges; ++e) {
    const int ip = p->edge_pose[e], il = p->edge_point[e];
    if (ip < 0 || ip >= P + p->n_fixed || il < 0 || il >= L) { set_error("edge %d out of range", e); return OSH_ERR_INVALID; }
    if (ip < P) obs[il].push_back(ip);
  }
  std::vector<int> lmo(L + 1, 0), nfree(L), epose;
  for (int j = 0; j < L; ++j) {
    std::sort(obs[j].begin(), obs[j].end());
    for (size_t k = 1; k < obs[j].size(); ++k)
      if (obs[j][k] == obs[j][k - 1]) { set_error("landmark %d observed twice by pose %d", j, obs[j][k]); return OSH_ERR_UNSUPPORTED; }
    nfree[j] = (int)obs[j].size();
    lmo[j] = (int)epose.size();
    epose.insert(epose.end(), obs[j].begin(), obs[j].end());
  }
  lmo[L] = (int)epose.size();
  epose.push_back(0);
  SchurPlan plan;
  std::vector<plan_detail::Build> builds;
  std::vector<SRec> brecs;
  plan_detail::PlanScratch psc;
  // (a call with one window cuts its items at item_max_lm(1) landmarks; OSH_LBA_ITEM_MAX overrides it: the tests run the check at 8, 24 and 64)
  if (!plan_window(0, P, L, lmo.data(), nfree.data(), epose.data(), builds, brecs, plan, psc, item_max_lm(1))) { set_error("landmark with > 254 observers"); return OSH_ERR_UNSUPPORTED; }
  std::vector<int> build_win(builds.size(), 0);
  finish_plan(build_win, builds, brecs, plan);

  // ---- verification
  std::map<std::pair<int, int>, std::pair<int, int>> range;   // block -> [start, start+count)
  std::vector<std::pair<int, int>> crange(P);
  for (const RBlk& rb : plan.rblk) {
    const int i = rb.ij & 0xffff, j = (rb.ij >> 16) & 0xffff;
    if (j == 0xffff) crange[i] = {rb.start, rb.start + rb.count};
    else range[{i, j}] = {rb.start, rb.start + rb.count};
  }
  if ((int)range.size() != P * (P + 1) / 2) { set_error("plan: %zu blocks of S, expected %d", range.size(), P * (P + 1) / 2); return OSH_ERR_DEVICE; }
  std::vector<int> used(plan.n_contrib, 0), cused(plan.n_ccontrib, 0), owner(L, 0);
  std::vector<std::set<std::pair<int, int>>> covered(L);
  std::vector<std::set<int>> ccovered(L);
  for (size_t it = 0; it < plan.items.size(); ++it) {
    const SItem& I = plan.items[it];
    const bool sym = (I.shape >> 16) & 1;
    if (sym != (it < (size_t)plan.n_sym)) { set_error("plan: item %zu on the wrong side of n_sym", it); return OSH_ERR_DEVICE; }
    const int* X = &plan.pose_x[it * 8];
    const int* Y = &plan.pose_y[it * 8];
    std::set<std::pair<int, int>> live;
    std::set<int> clive;
    for (int r = 0; r < I.n_lm; ++r) {
      const SRec& R = plan.recs[(size_t)I.rec_off + r];
      const unsigned long long xs = R.x_lo | ((unsigned long long)R.x_hi << 32), ys = R.y_lo | ((unsigned long long)R.y_hi << 32);
      if (R.flags & 1) owner[R.lm]++;
      if (R.e_first != lmo[R.lm]) { set_error("plan: record of landmark %d has a wrong first edge", R.lm); return OSH_ERR_DEVICE; }
      for (int sa = 0; sa < 8; ++sa) {
        const unsigned ra = (unsigned)(xs >> (8 * sa)) & 0xff;
        if (ra == kAbsent) continue;
        if ((int)ra >= nfree[R.lm] || obs[R.lm][ra] != X[sa]) { set_error("plan: item %zu slot %d does not match landmark %d", it, sa, R.lm); return OSH_ERR_DEVICE; }
        if (sym) { if (!ccovered[R.lm].insert((int)ra).second) { set_error("plan: rhs term of landmark %d rank %u twice", R.lm, ra); return OSH_ERR_DEVICE; } clive.insert(sa); }
        for (int sb = sym ? sa : 0; sb < 8; ++sb) {
          const unsigned rb = (unsigned)(ys >> (8 * sb)) & 0xff;
          if (rb == kAbsent) continue;
          if ((int)rb >= nfree[R.lm] || obs[R.lm][rb] != Y[sb] || rb < ra) { set_error("plan: item %zu column slot %d does not match landmark %d", it, sb, R.lm); return OSH_ERR_DEVICE; }
          if (!covered[R.lm].insert({(int)ra, (int)rb}).second) { set_error("plan: pair (%u,%u) of landmark %d covered twice", ra, rb, R.lm); return OSH_ERR_DEVICE; }
          live.insert({sa, sb});
        }
      }
    }
    for (int sa = 0; sa < 8; ++sa) {
      const int cs = plan.c_slot[it * 8 + sa];
      if ((cs >= 0) != (clive.count(sa) > 0)) { set_error("plan: item %zu rhs slot %d liveness mismatch", it, sa); return OSH_ERR_DEVICE; }
      if (cs >= 0) {
        if (cs < crange[X[sa]].first || cs >= crange[X[sa]].second) { set_error("plan: rhs slot outside its pose range"); return OSH_ERR_DEVICE; }
        cused[cs]++;
      }
      for (int sb = 0; sb < 8; ++sb) {
        const int ps = plan.pair_slot[it * 64 + sa * 8 + sb];
        if ((ps >= 0) != (live.count({sa, sb}) > 0)) { set_error("plan: item %zu pair (%d,%d) liveness mismatch", it, sa, sb); return OSH_ERR_DEVICE; }
        if (ps >= 0) {
          const auto rg = range.find({X[sa], Y[sb]});
          if (rg == range.end() || ps < rg->second.first || ps >= rg->second.second) { set_error("plan: contribution slot outside its block range"); return OSH_ERR_DEVICE; }
          used[ps]++;
        }
      }
    }
  }
  for (int j = 0; j < L; ++j) {
    const long long k = nfree[j];
    if ((long long)covered[j].size() != k * (k + 1) / 2 || (long long)ccovered[j].size() != k || owner[j] != 1) {
      set_error("plan: landmark %d: %zu of %lld pairs, %zu of %lld rhs terms, %d owners", j, covered[j].size(), k * (k + 1) / 2, ccovered[j].size(), k, owner[j]);
      return OSH_ERR_DEVICE;
    }
  }
  for (int u : used) if (u != 1) { set_error("plan: a contribution slot is written %d times", u); return OSH_ERR_DEVICE; }
  for (int u : cused) if (u != 1) { set_error("plan: a rhs contribution slot is written %d times", u); return OSH_ERR_DEVICE; }
  stats[0] = (int64_t)plan.items.size(); stats[1] = plan.n_sym; stats[2] = (int64_t)plan.recs.size();
  stats[3] = (int64_t)plan.n_contrib; stats[4] = (int64_t)plan.n_ccontrib; stats[5] = plan.tile_steps; stats[6] = plan.pair_blocks;
  stats[7] = (int64_t)plan.rblk.size();
  return OSH_OK;
}
