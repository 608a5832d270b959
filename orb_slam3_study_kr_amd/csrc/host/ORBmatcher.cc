// ORBmatcher.cc -- ORB_SLAM3::ORBmatcher::SearchByProjection on MI355X (host side).
//
// The candidate loops of the reference (src/ORBmatcher.cc:84-120, 1743-1768, 1949-1964, 499-520) run on the GPU as one
// batched nearest / second-nearest Hamming search (osh_orb_*).  The candidates themselves are generated on the device
// too: the host hands over the keypoint positions and every query's window (centre, radius, level range, u_right test)
// and osh_orb_upload_grid reproduces Frame::GetFeaturesInArea / KeyFrame::GetFeaturesInArea including their candidate
// order.  What stays on the host is the projection geometry of each entry point and the sequential "this slot was just
// taken" rule, replayed exactly as SURVEY.md 8a prescribes: occupancy only ever removes candidates, so only a query whose
// best or second-best slot was claimed earlier in the same call is re-scanned with the reference's left-to-right loop
// (its candidate list is then rebuilt by the frame's own GetFeaturesInArea).
#include "ORBmatcher.h"

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

#include "orbslam3_hip.h"

namespace ORB_SLAM3 {

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

// src/ORBmatcher.cc:2058-2074: 8 x 32-bit popcount of a XOR b
int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
  const uint32_t* pa = a.ptr<uint32_t>();
  const uint32_t* pb = b.ptr<uint32_t>();
  int dist = 0;
  for (int i = 0; i < 8; ++i) dist += __builtin_popcount(pa[i] ^ pb[i]);
  return dist;
}

float ORBmatcher::RadiusByViewingCos(const float& viewCos) { return viewCos > 0.998 ? 2.5 : 4.0; }  // :215-221

namespace {

struct ThreadCtx {   // destroyed at thread exit (Tracking / LoopClosing threads come and go with System instances)
  osh_orb_ctx* ctx = nullptr;
  ~ThreadCtx() { if (ctx) osh_orb_destroy(ctx); }
};

osh_orb_ctx* thread_ctx() {
  static thread_local ThreadCtx holder;
  if (!holder.ctx) {
    const char* dev = std::getenv("ORBSLAM3_HIP_DEVICE");
    if (osh_orb_create(dev ? std::atoi(dev) : 0, &holder.ctx) != OSH_OK) {
      std::fprintf(stderr, "ORBmatcher: cannot create the HIP matcher context: %s\n", osh_last_error());
      holder.ctx = nullptr;
    }
  }
  return holder.ctx;
}

}  // namespace

osh_orb_ctx* HostMatcherContext() { return thread_ctx(); }

namespace {

struct Search {
  std::vector<uint8_t> qdesc;            // [nq*32]
  std::vector<float> win;                // [nq*3] x, y, r of the query's GetFeaturesInArea call
  std::vector<int32_t> lev;              // [nq*2] minLevel, maxLevel
  std::vector<float> ur;                 // [nq*2] predicted u_right and tolerance (empty: no u_right test in this entry point)
  std::vector<int32_t> best_idx, best_dist, second_dist, best_level, second_level, second_idx;
  int nq() const { return (int)(win.size() / 3); }
  void add(const cv::Mat& d, float x, float y, float r, int minLevel, int maxLevel) {
    const uint8_t* dp = d.ptr<uint8_t>(0);
    qdesc.insert(qdesc.end(), dp, dp + 32);
    win.push_back(x); win.push_back(y); win.push_back(r);
    lev.push_back(minLevel); lev.push_back(maxLevel);
  }
};

// the frame / keyframe whose keypoints are searched
struct Train {
  const cv::Mat* desc = nullptr;
  int row0 = 0;                          // first descriptor row of this keypoint set (right-camera keypoints: Nleft)
  std::vector<int32_t> level;
  std::vector<float> xy, uright;         // uright empty: no stereo test
  std::vector<uint8_t> skip;             // slots that are no candidates when the call starts
  float min_x = 0, min_y = 0, winv = 0, hinv = 0;
  int cols = 0, rows = 0;
  int n() const { return (int)level.size(); }
};

Train train_of(const Frame& F) {
  Train t;
  t.desc = &F.mDescriptors;
  t.level.resize(F.N); t.xy.resize((size_t)F.N * 2); t.skip.assign(F.N, 0);
  for (int i = 0; i < F.N; ++i) { t.level[i] = F.mvKeysUn[i].octave; t.xy[2 * i] = F.mvKeysUn[i].pt.x; t.xy[2 * i + 1] = F.mvKeysUn[i].pt.y; }
  t.min_x = F.mnMinX; t.min_y = F.mnMinY; t.winv = F.mfGridElementWidthInv; t.hinv = F.mfGridElementHeightInv;
  t.cols = FRAME_GRID_COLS; t.rows = FRAME_GRID_ROWS;
  return t;
}

// queries and the train side's grid become device resident; nullptr (and a message) on failure or for an empty search
osh_orb_ctx* upload_search(const Search& s, const Train& t) {
  if (s.nq() == 0) return nullptr;
  osh_orb_ctx* ctx = thread_ctx();
  if (!ctx) return nullptr;
  osh_orb_batch b;
  b.n_pairs = 1; b.n_query = s.nq(); b.n_train = t.n();
  b.query_desc = s.qdesc.data(); b.train_desc = t.desc->ptr<uint8_t>(t.row0); b.train_level = t.level.data();
  b.cand_off = nullptr; b.cand_idx = nullptr; b.pair_cand_base = nullptr;
  osh_orb_grid g;
  g.train_xy = t.xy.data(); g.train_uright = t.uright.empty() ? nullptr : t.uright.data(); g.train_skip = t.skip.data();
  g.min_x = t.min_x; g.min_y = t.min_y; g.cell_w_inv = t.winv; g.cell_h_inv = t.hinv; g.cols = t.cols; g.rows = t.rows;
  g.query_window = s.win.data(); g.query_levels = s.lev.data(); g.query_uright = s.ur.empty() ? nullptr : s.ur.data();
  if (osh_orb_upload_grid(ctx, &b, &g) != OSH_OK) {
    std::fprintf(stderr, "ORBmatcher: device upload failed: %s\n", osh_last_error());
    return nullptr;
  }
  return ctx;
}

// one batched device search of all queries; the candidates come from the train side's grid
bool device_search(Search& s, const Train& t) {
  const int nq = s.nq();
  for (auto* v : {&s.best_idx, &s.best_dist, &s.second_dist, &s.best_level, &s.second_level, &s.second_idx}) v->assign(nq, -1);
  if (nq == 0) return true;
  osh_orb_ctx* ctx = upload_search(s, t);
  if (!ctx) return false;
  if (osh_orb_match(ctx) != OSH_OK ||
      osh_orb_download(ctx, s.best_idx.data(), s.best_dist.data(), s.second_dist.data(), s.best_level.data(),
                       s.second_level.data(), s.second_idx.data()) != OSH_OK) {
    std::fprintf(stderr, "ORBmatcher: device search failed: %s\n", osh_last_error());
    return false;
  }
  return true;
}

// the reference's scan of one query's candidates with the current occupancy (only for contested queries): `area` is the
// entry point's own GetFeaturesInArea call, the static filters are the ones the device applied
template <class Area>
void rescan(const Search& s, int q, const Train& t, const std::vector<uint8_t>& occupied, Area area,
            int& bestIdx, int& bestDist, int& bestDist2, int& bestLevel, int& bestLevel2) {
  bestDist = 256; bestLevel = -1; bestDist2 = 256; bestLevel2 = -1; bestIdx = -1;
  const uint32_t* qd = reinterpret_cast<const uint32_t*>(&s.qdesc[(size_t)q * 32]);
  const float r = s.win[3 * q + 2];
  if (!(r > 0.f)) return;
  const int minLevel = s.lev[2 * q], maxLevel = s.lev[2 * q + 1];
  const std::vector<size_t> cand = area(s.win[3 * q], s.win[3 * q + 1], r, minLevel, maxLevel);
  for (const size_t i : cand) {
    const int idx = (int)i;
    if (t.skip[idx] || occupied[idx]) continue;
    if (t.level[idx] < minLevel || (maxLevel >= 0 && t.level[idx] > maxLevel)) continue;
    if (!s.ur.empty() && !t.uright.empty() && t.uright[idx] > 0) {
      const float er = std::fabs(s.ur[2 * q] - t.uright[idx]);
      if (er > s.ur[2 * q + 1]) continue;
    }
    const uint32_t* td = t.desc->ptr<uint32_t>(t.row0 + idx);
    int dist = 0;
    for (int k = 0; k < 8; ++k) dist += __builtin_popcount(qd[k] ^ td[k]);
    if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = t.level[idx]; bestIdx = idx; }
    else if (dist < bestDist2) { bestLevel2 = t.level[idx]; bestDist2 = dist; }
  }
}

}  // namespace

namespace {

// one camera of a fisheye stereo frame as a search target: left keypoints mvKeys / mGrid / descriptor rows [0, Nleft), right
// keypoints mvKeysRight / mGridRight / rows [Nleft, N) (src/Frame.cc:406-416, 697-699)
Train train_of_rig(const Frame& F, bool right) {
  Train t;
  const std::vector<cv::KeyPoint>& keys = right ? F.mvKeysRight : F.mvKeys;
  const int n = right ? (F.N - F.Nleft) : F.Nleft;
  t.desc = &F.mDescriptors;
  t.row0 = right ? F.Nleft : 0;
  t.level.resize(n); t.xy.resize((size_t)n * 2); t.skip.assign(n, 0);
  for (int i = 0; i < n; ++i) { t.level[i] = keys[i].octave; t.xy[2 * i] = keys[i].pt.x; t.xy[2 * i + 1] = keys[i].pt.y; }
  t.min_x = F.mnMinX; t.min_y = F.mnMinY; t.winv = F.mfGridElementWidthInv; t.hinv = F.mfGridElementHeightInv;
  t.cols = FRAME_GRID_COLS; t.rows = FRAME_GRID_ROWS;
  return t;
}

// src/ORBmatcher.cc:43-213 on a fisheye stereo frame (Nleft != -1): per map point the left-camera pass (:60-141) and then the
// right-camera pass (:144-210).  Both passes of ALL points are searched on the device first (two batched launches, one per
// camera); the sequential part -- slots claimed by earlier points, the stereo partner a match also claims, the `continue` of a
// failed left ratio test that skips the point's right pass -- is replayed on the host in the reference's order.
// (A free function: the class declaration stays the reference's own, include/ORBmatcher.h.)
int search_local_points_rig(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th, const bool bFarPoints,
                            const float thFarPoints, const float mfNNratio) {
  const int TH_HIGH = ORBmatcher::TH_HIGH;
  auto RadiusByViewingCos = [](float viewCos) { return (float)(viewCos > 0.998 ? 2.5 : 4.0); };   // src/ORBmatcher.cc:215-221
  const bool bFactor = th != 1.0;
  const int NL = F.Nleft, NR = F.N - F.Nleft;
  Train tl = train_of_rig(F, false), tr = train_of_rig(F, true);
  auto occupied_at = [&F](int slot) { return (F.mvpMapPoints[slot] && F.mvpMapPoints[slot]->Observations() > 0) ? 1 : 0; };
  for (int i = 0; i < NL; ++i) tl.skip[i] = occupied_at(i);             // :88-90 at call entry
  for (int i = 0; i < NR; ++i) tr.skip[i] = occupied_at(i + NL);        // :164-166
  auto area_l = [&F](float x, float y, float r, int lo, int hi) { return F.GetFeaturesInArea(x, y, r, lo, hi, false); };
  auto area_r = [&F](float x, float y, float r, int lo, int hi) { return F.GetFeaturesInArea(x, y, r, lo, hi, true); };
  Search sl, sr;
  struct Q { MapPoint* mp; int ql, qr; };
  std::vector<Q> qs;
  for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
    MapPoint* pMP = vpMapPoints[iMP];
    if (!pMP->mbTrackInView && !pMP->mbTrackInViewR) continue;
    if (bFarPoints && pMP->mTrackDepth > thFarPoints) continue;
    if (pMP->isBad()) continue;
    Q q{pMP, -1, -1};
    if (pMP->mbTrackInView) {
      const int nPredictedLevel = pMP->mnTrackScaleLevel;
      float r = RadiusByViewingCos(pMP->mTrackViewCos);
      if (bFactor) r *= th;
      q.ql = sl.nq();
      sl.add(pMP->GetDescriptor(), pMP->mTrackProjX, pMP->mTrackProjY, r * F.mvScaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
    }
    if (pMP->mbTrackInViewR && pMP->mnTrackScaleLevelR != -1) {
      const int nPredictedLevel = pMP->mnTrackScaleLevelR;
      const float r = RadiusByViewingCos(pMP->mTrackViewCosR);   // no th factor in the right-camera pass (:148)
      q.qr = sr.nq();
      sr.add(pMP->GetDescriptor(), pMP->mTrackProjXR, pMP->mTrackProjYR, r * F.mvScaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
    }
    qs.push_back(q);
  }
  if (!device_search(sl, tl) || !device_search(sr, tr)) return 0;

  int nmatches = 0;
  std::vector<uint8_t> taken_l(NL, 0), taken_r(NR, 0);   // slots claimed during this call by points with observations
  for (const Q& q : qs) {
    MapPoint* pMP = q.mp;
    const bool claims = pMP->Observations() > 0;
    if (q.ql >= 0) {
      int bestIdx = sl.best_idx[q.ql], bestDist = sl.best_dist[q.ql], bestDist2 = sl.second_dist[q.ql];
      int bestLevel = sl.best_level[q.ql], bestLevel2 = sl.second_level[q.ql];
      if ((bestIdx >= 0 && taken_l[bestIdx]) || (sl.second_idx[q.ql] >= 0 && taken_l[sl.second_idx[q.ql]]))
        rescan(sl, q.ql, tl, taken_l, area_l, bestIdx, bestDist, bestDist2, bestLevel, bestLevel2);
      if (bestDist <= TH_HIGH) {
        if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;   // also skips this point's right-camera pass (:125-126)
        if (bestLevel != bestLevel2 || bestDist <= mfNNratio * bestDist2) {
          F.mvpMapPoints[bestIdx] = pMP;
          if (claims) taken_l[bestIdx] = 1;
          if (F.mvLeftToRightMatch[bestIdx] != -1) {   // also match with the stereo observation at the right camera (:131-135)
            F.mvpMapPoints[F.mvLeftToRightMatch[bestIdx] + NL] = pMP;
            if (claims) taken_r[F.mvLeftToRightMatch[bestIdx]] = 1;
            nmatches++;
          }
          nmatches++;
        }
      }
    }
    if (q.qr >= 0) {
      int bestIdx = sr.best_idx[q.qr], bestDist = sr.best_dist[q.qr], bestDist2 = sr.second_dist[q.qr];
      int bestLevel = sr.best_level[q.qr], bestLevel2 = sr.second_level[q.qr];
      if ((bestIdx >= 0 && taken_r[bestIdx]) || (sr.second_idx[q.qr] >= 0 && taken_r[sr.second_idx[q.qr]]))
        rescan(sr, q.qr, tr, taken_r, area_r, bestIdx, bestDist, bestDist2, bestLevel, bestLevel2);
      if (bestDist <= TH_HIGH) {
        if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
        if (F.mvRightToLeftMatch[bestIdx] != -1) {     // :199-203
          F.mvpMapPoints[F.mvRightToLeftMatch[bestIdx]] = pMP;
          if (claims) taken_l[F.mvRightToLeftMatch[bestIdx]] = 1;
          nmatches++;
        }
        F.mvpMapPoints[bestIdx + NL] = pMP;
        if (claims) taken_r[bestIdx] = 1;
        nmatches++;
      }
    }
  }
  return nmatches;
}

}  // namespace

// src/ORBmatcher.cc:43-213: Nleft == -1 layouts (monocular, rectified stereo, RGB-D) here, fisheye stereo frames above.
int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th, const bool bFarPoints,
                                   const float thFarPoints) {
  if (F.Nleft != -1) return search_local_points_rig(F, vpMapPoints, th, bFarPoints, thFarPoints, mfNNratio);
  const bool bFactor = th != 1.0;
  Train t = train_of(F);
  t.uright = F.mvuRight;                                            // stereo consistency window (:92-97)
  for (int i = 0; i < F.N; ++i) t.skip[i] = (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) ? 1 : 0;   // :88-90 at call entry
  Search s;
  std::vector<MapPoint*> qMP;
  for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
    MapPoint* pMP = vpMapPoints[iMP];
    if (!pMP->mbTrackInView && !pMP->mbTrackInViewR) continue;
    if (bFarPoints && pMP->mTrackDepth > thFarPoints) continue;
    if (pMP->isBad()) continue;
    if (!pMP->mbTrackInView) continue;
    const int nPredictedLevel = pMP->mnTrackScaleLevel;
    float r = RadiusByViewingCos(pMP->mTrackViewCos);   // window size depends on the viewing direction (:66-72)
    if (bFactor) r *= th;
    const float win = r * F.mvScaleFactors[nPredictedLevel];
    // candidates: GetFeaturesInArea(mTrackProjX, mTrackProjY, win, level-1, level) minus occupied slots, |ur - uR| <= win
    s.add(pMP->GetDescriptor(), pMP->mTrackProjX, pMP->mTrackProjY, win, nPredictedLevel - 1, nPredictedLevel);
    s.ur.push_back(pMP->mTrackProjXR); s.ur.push_back(win);
    qMP.push_back(pMP);
  }
  // search, acceptance rule (:123-139) and the sequential slot occupancy (:88-90) all on the device: osh_orb_match_local_points
  osh_orb_ctx* ctx = upload_search(s, t);
  if (!ctx) return 0;
  std::vector<uint8_t> blocks(qMP.size());
  for (size_t q = 0; q < qMP.size(); ++q) blocks[q] = qMP[q]->Observations() > 0 ? 1 : 0;
  std::vector<int32_t> assignment(F.N, -1);
  int32_t nmatches = 0;
  if (osh_orb_match_local_points(ctx, mfNNratio, TH_HIGH, nullptr, blocks.data(), assignment.data(), &nmatches, nullptr, nullptr) != OSH_OK) {
    std::fprintf(stderr, "ORBmatcher: device search failed: %s\n", osh_last_error());
    return 0;
  }
  for (int i = 0; i < F.N; ++i) if (assignment[i] >= 0) F.mvpMapPoints[i] = qMP[assignment[i]];
  return nmatches;
}

// src/ORBmatcher.cc:2012-2053
void ORBmatcher::ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int n = (int)histo[i].size();
    if (n > max1) { max3 = max2; max2 = max1; max1 = n; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (n > max2) { max3 = max2; max2 = n; ind3 = ind2; ind2 = i; }
    else if (n > max3) { max3 = n; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

namespace {

// src/ORBmatcher.cc:1676-1887 on fisheye stereo frames (CurrentFrame.Nleft != -1): per map point of the last frame the search
// among the current frame's LEFT keypoints (:1696-1792, no u_right test) and then, unless the left candidate list was empty
// (`continue` at :1732), among its RIGHT keypoints with the point moved through Trl and projected with mpCamera (:1794-1858).
// Both searches of all points run on the device first; the sequential part is replayed on the host.
int search_last_frame_rig(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bForward, const bool bBackward,
                          const bool mbCheckOrientation, void (*three_maxima)(std::vector<int>*, int, int&, int&, int&)) {
  const int HISTO_LENGTH = ORBmatcher::HISTO_LENGTH, TH_HIGH = ORBmatcher::TH_HIGH;
  std::vector<int> rotHist[30];
  for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
  const float factor = 1.0f / HISTO_LENGTH;
  const Sophus::SE3f Tcw = CurrentFrame.GetPose();
  const int NL = CurrentFrame.Nleft, NR = CurrentFrame.N - CurrentFrame.Nleft;
  Train tl = train_of_rig(CurrentFrame, false), tr = train_of_rig(CurrentFrame, true);
  auto occupied_at = [&CurrentFrame](int slot) { return (CurrentFrame.mvpMapPoints[slot] && CurrentFrame.mvpMapPoints[slot]->Observations() > 0) ? 1 : 0; };
  for (int i = 0; i < NL; ++i) tl.skip[i] = occupied_at(i);
  for (int i = 0; i < NR; ++i) tr.skip[i] = occupied_at(i + NL);
  auto area_l = [&CurrentFrame](float x, float y, float r, int lo, int hi) { return CurrentFrame.GetFeaturesInArea(x, y, r, lo, hi, false); };
  auto area_r = [&CurrentFrame](float x, float y, float r, int lo, int hi) { return CurrentFrame.GetFeaturesInArea(x, y, r, lo, hi, true); };
  Search sl, sr;
  std::vector<int> qLast;
  for (int i = 0; i < LastFrame.N; i++) {
    MapPoint* pMP = LastFrame.mvpMapPoints[i];
    if (!pMP || LastFrame.mvbOutlier[i]) continue;
    const Eigen::Vector3f x3Dc = Tcw * pMP->GetWorldPos();
    const float invzc = 1.0 / x3Dc(2);
    if (invzc < 0) continue;
    const Eigen::Vector2f uv = CurrentFrame.mpCamera->project(x3Dc);
    if (uv(0) < CurrentFrame.mnMinX || uv(0) > CurrentFrame.mnMaxX) continue;
    if (uv(1) < CurrentFrame.mnMinY || uv(1) > CurrentFrame.mnMaxY) continue;
    const int nLastOctave = (LastFrame.Nleft == -1 || i < LastFrame.Nleft) ? LastFrame.mvKeys[i].octave : LastFrame.mvKeysRight[i - LastFrame.Nleft].octave;
    const float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
    int lo, hi;
    if (bForward) { lo = nLastOctave; hi = -1; }
    else if (bBackward) { lo = 0; hi = nLastOctave; }
    else { lo = nLastOctave - 1; hi = nLastOctave + 1; }
    sl.add(pMP->GetDescriptor(), uv(0), uv(1), radius, lo, hi);
    const Eigen::Vector3f x3Dr = CurrentFrame.GetRelativePoseTrl() * x3Dc;       // :1795
    const Eigen::Vector2f uvr = CurrentFrame.mpCamera->project(x3Dr);            // :1796 (mpCamera, as in the reference)
    sr.add(pMP->GetDescriptor(), uvr(0), uvr(1), radius, lo, hi);
    qLast.push_back(i);
  }
  if (!device_search(sl, tl) || !device_search(sr, tr)) return 0;

  int nmatches = 0;
  std::vector<uint8_t> taken_l(NL, 0), taken_r(NR, 0);
  auto last_kp = [&LastFrame](int i) -> const cv::KeyPoint& {
    return (LastFrame.Nleft == -1) ? LastFrame.mvKeysUn[i] : (i < LastFrame.Nleft) ? LastFrame.mvKeys[i] : LastFrame.mvKeysRight[i - LastFrame.Nleft];
  };
  for (int q = 0; q < sl.nq(); ++q) {
    MapPoint* pMP = LastFrame.mvpMapPoints[qLast[q]];
    const bool claims = pMP->Observations() > 0;
    int d2, l1, l2;
    {
      int bestIdx2 = sl.best_idx[q], bestDist = sl.best_dist[q];
      if (bestIdx2 >= 0 && taken_l[bestIdx2]) rescan(sl, q, tl, taken_l, area_l, bestIdx2, bestDist, d2, l1, l2);
      // `if(vIndices2.empty()) continue;` (:1731-1732) also skips the right-camera search of this point: the device reports
      // "no candidate" for an empty list AND for a list whose entries are all occupied, so the (rare) no-candidate case asks
      if (bestIdx2 < 0 && area_l(sl.win[3 * q], sl.win[3 * q + 1], sl.win[3 * q + 2], sl.lev[2 * q], sl.lev[2 * q + 1]).empty()) continue;
      if (bestDist <= TH_HIGH) {
        CurrentFrame.mvpMapPoints[bestIdx2] = pMP;
        if (claims) taken_l[bestIdx2] = 1;
        nmatches++;
        if (mbCheckOrientation) {
          float rot = last_kp(qLast[q]).angle - CurrentFrame.mvKeys[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(bestIdx2);
        }
      }
    }
    {
      int bestIdx2 = sr.best_idx[q], bestDist = sr.best_dist[q];
      if (bestIdx2 >= 0 && taken_r[bestIdx2]) rescan(sr, q, tr, taken_r, area_r, bestIdx2, bestDist, d2, l1, l2);
      if (bestDist <= TH_HIGH) {
        CurrentFrame.mvpMapPoints[bestIdx2 + NL] = pMP;
        if (claims) taken_r[bestIdx2] = 1;
        nmatches++;
        if (mbCheckOrientation) {
          float rot = last_kp(qLast[q]).angle - CurrentFrame.mvKeysRight[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(bestIdx2 + NL);
        }
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (const int slot : rotHist[i]) { CurrentFrame.mvpMapPoints[slot] = nullptr; nmatches--; }
    }
  }
  return nmatches;
}

// ComputeThreeMaxima is a protected member; the rig search above (a free function) gets this restatement of :2012-2053
void three_maxima_fn(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int n = (int)histo[i].size();
    if (n > max1) { max3 = max2; max2 = max1; max1 = n; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (n > max2) { max3 = max2; max2 = n; ind3 = ind2; ind2 = i; }
    else if (n > max3) { max3 = n; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

}  // namespace

// src/ORBmatcher.cc:1676-1887.
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) {
  std::vector<int> rotHist[30];
  for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
  const float factor = 1.0f / HISTO_LENGTH;
  const Sophus::SE3f Tcw = CurrentFrame.GetPose();
  // twc = -Rcw^T tcw ; tlc = Tlw * twc  (forward / backward motion test, :1686-1693)
  const Eigen::Quaternionf qc = Tcw.unit_quaternion();
  const Sophus::SE3f Twc_rot(Eigen::Quaternionf(qc.w(), -qc.x(), -qc.y(), -qc.z()), Eigen::Vector3f(0, 0, 0));
  const Eigen::Vector3f mt(-Tcw.translation()(0), -Tcw.translation()(1), -Tcw.translation()(2));
  const Eigen::Vector3f twc = Twc_rot * mt;
  const Eigen::Vector3f tlc = LastFrame.GetPose() * twc;
  const bool bForward = tlc(2) > CurrentFrame.mb && !bMono;
  const bool bBackward = -tlc(2) > CurrentFrame.mb && !bMono;
  if (CurrentFrame.Nleft != -1) return search_last_frame_rig(CurrentFrame, LastFrame, th, bForward, bBackward, mbCheckOrientation, three_maxima_fn);

  Train t = train_of(CurrentFrame);
  t.uright = CurrentFrame.mvuRight;
  for (int i = 0; i < CurrentFrame.N; ++i)
    t.skip[i] = (CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0) ? 1 : 0;
  auto area = [&CurrentFrame](float x, float y, float r, int lo, int hi) { return CurrentFrame.GetFeaturesInArea(x, y, r, lo, hi); };
  Search s;
  std::vector<int> qLast;  // index in LastFrame of every query
  for (int i = 0; i < LastFrame.N; i++) {
    MapPoint* pMP = LastFrame.mvpMapPoints[i];
    if (!pMP || LastFrame.mvbOutlier[i]) continue;
    const Eigen::Vector3f x3Dc = Tcw * pMP->GetWorldPos();
    const float invzc = 1.0 / x3Dc(2);
    if (invzc < 0) continue;
    const Eigen::Vector2f uv = CurrentFrame.mpCamera->project(x3Dc);
    if (uv(0) < CurrentFrame.mnMinX || uv(0) > CurrentFrame.mnMaxX) continue;
    if (uv(1) < CurrentFrame.mnMinY || uv(1) > CurrentFrame.mnMaxY) continue;
    const int nLastOctave = (LastFrame.Nleft == -1 || i < LastFrame.Nleft) ? LastFrame.mvKeys[i].octave : LastFrame.mvKeysRight[i - LastFrame.Nleft].octave;
    const float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];   // window scales with the octave
    // level range by the motion direction (:1744-1750): forward -> [octave, inf), backward -> [0, octave], else octave +- 1
    int lo, hi;
    if (bForward) { lo = nLastOctave; hi = -1; }
    else if (bBackward) { lo = 0; hi = nLastOctave; }
    else { lo = nLastOctave - 1; hi = nLastOctave + 1; }
    s.add(pMP->GetDescriptor(), uv(0), uv(1), radius, lo, hi);
    s.ur.push_back(uv(0) - CurrentFrame.mbf * invzc); s.ur.push_back(radius);   // |ur - mvuRight| <= radius (:1762-1767)
    qLast.push_back(i);
  }
  if (!device_search(s, t)) return 0;

  int nmatches = 0;
  std::vector<uint8_t> taken(CurrentFrame.N, 0);
  for (int q = 0; q < s.nq(); ++q) {
    int bestIdx2 = s.best_idx[q], bestDist = s.best_dist[q], d2, l1, l2;
    if (bestIdx2 >= 0 && taken[bestIdx2]) rescan(s, q, t, taken, area, bestIdx2, bestDist, d2, l1, l2);
    if (bestDist > TH_HIGH) continue;
    MapPoint* pMP = LastFrame.mvpMapPoints[qLast[q]];
    CurrentFrame.mvpMapPoints[bestIdx2] = pMP;
    if (pMP->Observations() > 0) taken[bestIdx2] = 1;
    nmatches++;
    if (mbCheckOrientation) {
      float rot = LastFrame.mvKeysUn[qLast[q]].angle - CurrentFrame.mvKeysUn[bestIdx2].angle;
      if (rot < 0.0) rot += 360.0f;
      int bin = (int)std::round(rot * factor);   // factor = 1/30 (sic): only bins 0..12 are ever hit
      if (bin == HISTO_LENGTH) bin = 0;
      rotHist[bin].push_back(bestIdx2);
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (const int slot : rotHist[i]) { CurrentFrame.mvpMapPoints[slot] = nullptr; nmatches--; }
    }
  }
  return nmatches;
}

// src/ORBmatcher.cc:1889-2010 (Tracking::Relocalization): the keyframe's map points are projected with the current
// pose estimate; best candidate only; a slot holding ANY map point is skipped (:1952) -- also one filled earlier in this
// call; accept bestDist <= ORBdist; rotation histogram with the keyframe keypoint's angle.
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                                   const int ORBdist) {
  const Sophus::SE3f Tcw = CurrentFrame.GetPose();
  const Eigen::Vector3f Ow = Tcw.inverse().translation();
  std::vector<int> rotHist[30];
  for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
  const float factor = 1.0f / HISTO_LENGTH;
  const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();

  // on a fisheye stereo frame the reference searches the LEFT keypoints only here (GetFeaturesInArea's bRight defaults to false)
  Train t = CurrentFrame.Nleft == -1 ? train_of(CurrentFrame) : train_of_rig(CurrentFrame, false);
  for (int i = 0; i < t.n(); ++i) t.skip[i] = CurrentFrame.mvpMapPoints[i] ? 1 : 0;   // any matched slot (:1952)
  auto area = [&CurrentFrame](float x, float y, float r, int lo, int hi) { return CurrentFrame.GetFeaturesInArea(x, y, r, lo, hi); };
  Search s;
  std::vector<int> qKF;  // keypoint index in pKF of every query
  for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
    MapPoint* pMP = vpMPs[i];
    if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
    const Eigen::Vector3f x3Dw = pMP->GetWorldPos();
    const Eigen::Vector3f x3Dc = Tcw * x3Dw;
    const Eigen::Vector2f uv = CurrentFrame.mpCamera->project(x3Dc);
    if (uv(0) < CurrentFrame.mnMinX || uv(0) > CurrentFrame.mnMaxX) continue;
    if (uv(1) < CurrentFrame.mnMinY || uv(1) > CurrentFrame.mnMaxY) continue;
    // predicted scale level from the distance to the camera centre (:1922-1933)
    const float px = x3Dw(0) - Ow(0), py = x3Dw(1) - Ow(1), pz = x3Dw(2) - Ow(2);
    const float dist3D = std::sqrt(px * px + py * py + pz * pz);
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
    const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
    s.add(pMP->GetDescriptor(), uv(0), uv(1), radius, nPredictedLevel - 1, nPredictedLevel + 1);
    qKF.push_back((int)i);
  }
  if (!device_search(s, t)) return 0;

  int nmatches = 0;
  std::vector<uint8_t> taken(CurrentFrame.N, 0);
  for (int q = 0; q < s.nq(); ++q) {
    int bestIdx2 = s.best_idx[q], bestDist = s.best_dist[q], d2, l1, l2;
    if (bestIdx2 >= 0 && taken[bestIdx2]) rescan(s, q, t, taken, area, bestIdx2, bestDist, d2, l1, l2);
    if (bestDist > ORBdist) continue;
    CurrentFrame.mvpMapPoints[bestIdx2] = vpMPs[qKF[q]];
    taken[bestIdx2] = 1;
    nmatches++;
    if (mbCheckOrientation) {
      const cv::KeyPoint& kpCF = CurrentFrame.Nleft == -1 ? CurrentFrame.mvKeysUn[bestIdx2] : CurrentFrame.mvKeys[bestIdx2];
      float rot = pKF->mvKeysUn[qKF[q]].angle - kpCF.angle;
      if (rot < 0.0) rot += 360.0f;
      int bin = (int)std::round(rot * factor);
      if (bin == HISTO_LENGTH) bin = 0;
      rotHist[bin].push_back(bestIdx2);
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (const int slot : rotHist[i]) { CurrentFrame.mvpMapPoints[slot] = nullptr; nmatches--; }
    }
  }
  return nmatches;
}

namespace {

// Common body of the two Sim3 overloads (src/ORBmatcher.cc:427-532, 534-646).  They differ in how the point is projected
// (camera model object vs. the keyframe's pinhole intrinsics with a float 1/z) and in the extra vpMatchedKF output.
int search_by_sim3(KeyFrame* pKF, Sophus::Sim3f& Scw, const std::vector<MapPoint*>& vpPoints, const std::vector<KeyFrame*>* vpPointsKFs,
                   std::vector<MapPoint*>& vpMatched, std::vector<KeyFrame*>* vpMatchedKF, int th, float ratioHamming, int th_low) {
  const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
  const Eigen::Vector3f ts = Scw.translation();
  const float sc = Scw.scale();
  const Sophus::SE3f Tcw(Scw.rotationMatrix(), Eigen::Vector3f(ts(0) / sc, ts(1) / sc, ts(2) / sc));
  const Eigen::Vector3f Ow = Tcw.inverse().translation();
  std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
  spAlreadyFound.erase(static_cast<MapPoint*>(NULL));

  const int N = (int)vpMatched.size();
  Train t;
  t.desc = &pKF->mDescriptors;
  t.level.resize(N); t.xy.resize((size_t)N * 2); t.skip.assign(N, 0);
  for (int i = 0; i < N; ++i) {
    t.level[i] = pKF->mvKeysUn[i].octave; t.xy[2 * i] = pKF->mvKeysUn[i].pt.x; t.xy[2 * i + 1] = pKF->mvKeysUn[i].pt.y;
    t.skip[i] = vpMatched[i] ? 1 : 0;                                  // matched slots are skipped (:501-502)
  }
  t.min_x = (float)pKF->mnMinX; t.min_y = (float)pKF->mnMinY; t.winv = pKF->mfGridElementWidthInv; t.hinv = pKF->mfGridElementHeightInv;
  t.cols = pKF->mnGridCols; t.rows = pKF->mnGridRows;
  auto area = [pKF](float x, float y, float r, int, int) { return pKF->GetFeaturesInArea(x, y, r); };
  Search s;
  std::vector<int> qMP;
  for (int iMP = 0, iendMP = (int)vpPoints.size(); iMP < iendMP; iMP++) {
    MapPoint* pMP = vpPoints[iMP];
    if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
    const Eigen::Vector3f p3Dw = pMP->GetWorldPos();
    const Eigen::Vector3f p3Dc = Tcw * p3Dw;
    if (p3Dc(2) < 0.0) continue;
    float u, v;
    if (!vpPointsKFs) {
      const Eigen::Vector2f uv = pKF->mpCamera->project(p3Dc);   // :467
      u = uv(0); v = uv(1);
    } else {
      const float invz = 1 / p3Dc(2);                            // :573-578
      const float x = p3Dc(0) * invz, y = p3Dc(1) * invz;
      u = fx * x + cx; v = fy * y + cy;
    }
    if (!pKF->IsInImage(u, v)) continue;
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    const float px = p3Dw(0) - Ow(0), py = p3Dw(1) - Ow(1), pz = p3Dw(2) - Ow(2);
    const float dist = std::sqrt(px * px + py * py + pz * pz);
    if (dist < minDistance || dist > maxDistance) continue;
    const Eigen::Vector3f Pn = pMP->GetNormal();                 // viewing angle below 60 degrees (:485-489)
    const float dotp = px * Pn(0) + py * Pn(1) + pz * Pn(2);
    if (dotp < 0.5 * dist) continue;
    const int nPredictedLevel = pMP->PredictScale(dist, pKF);
    const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
    // KeyFrame::GetFeaturesInArea(u, v, radius) has no level arguments; the loop keeps levels L-1 .. L (:504-507)
    s.add(pMP->GetDescriptor(), u, v, radius, std::max(nPredictedLevel - 1, 0), nPredictedLevel);
    qMP.push_back(iMP);
  }
  if (!device_search(s, t)) return 0;

  int nmatches = 0;
  std::vector<uint8_t> taken(N, 0);
  for (int q = 0; q < s.nq(); ++q) {
    int bestIdx = s.best_idx[q], bestDist = s.best_dist[q], d2, l1, l2;
    if (bestIdx >= 0 && taken[bestIdx]) rescan(s, q, t, taken, area, bestIdx, bestDist, d2, l1, l2);
    if (bestIdx >= 0 && bestDist <= th_low * ratioHamming) {      // int <= float (:523,636)
      vpMatched[bestIdx] = vpPoints[qMP[q]];
      if (vpMatchedKF) (*vpMatchedKF)[bestIdx] = (*vpPointsKFs)[qMP[q]];
      taken[bestIdx] = 1;
      nmatches++;
    }
  }
  return nmatches;
}

}  // namespace

int ORBmatcher::SearchByProjection(KeyFrame* pKF, Sophus::Sim3f& Scw, const std::vector<MapPoint*>& vpPoints,
                                   std::vector<MapPoint*>& vpMatched, int th, float ratioHamming) {
  return search_by_sim3(pKF, Scw, vpPoints, nullptr, vpMatched, nullptr, th, ratioHamming, TH_LOW);
}

int ORBmatcher::SearchByProjection(KeyFrame* pKF, Sophus::Sim3<float>& Scw, const std::vector<MapPoint*>& vpPoints,
                                   const std::vector<KeyFrame*>& vpPointsKFs, std::vector<MapPoint*>& vpMatched,
                                   std::vector<KeyFrame*>& vpMatchedKF, int th, float ratioHamming) {
  return search_by_sim3(pKF, Scw, vpPoints, &vpPointsKFs, vpMatched, &vpMatchedKF, th, ratioHamming, TH_LOW);
}

namespace {

// the keypoints of a keyframe as a search target (mvKeysUn, mGrid: what KeyFrame::GetFeaturesInArea(x, y, r) walks)
Train train_of(KeyFrame* pKF, int N) {
  Train t;
  t.desc = &pKF->mDescriptors;
  t.level.resize(N); t.xy.resize((size_t)N * 2); t.skip.assign(N, 0);
  for (int i = 0; i < N; ++i) { t.level[i] = pKF->mvKeysUn[i].octave; t.xy[2 * i] = pKF->mvKeysUn[i].pt.x; t.xy[2 * i + 1] = pKF->mvKeysUn[i].pt.y; }
  t.min_x = (float)pKF->mnMinX; t.min_y = (float)pKF->mnMinY; t.winv = pKF->mfGridElementWidthInv; t.hinv = pKF->mfGridElementHeightInv;
  t.cols = pKF->mnGridCols; t.rows = pKF->mnGridRows;
  return t;
}

// One direction of SearchBySim3 (src/ORBmatcher.cc:1497-1576 / :1578-1657): the map points of keyframe `from` that are not matched
// yet, moved into the camera of keyframe `to` by Tfw and then S, searched among `to`'s keypoints of levels L-1 .. L around the
// projection.  No slot occupancy: every query is independent, so the device result IS the loop's result.
bool sim3_direction(KeyFrame* to_kf, const float fx, const float fy, const float cx, const float cy, const Sophus::SE3f& Tfw, const Sophus::Sim3f& S,
                    const std::vector<MapPoint*>& vpFrom, const std::vector<bool>& vbAlready, int n_to, float th, int th_high, std::vector<int>& vnMatch) {
  Train t = train_of(to_kf, n_to);
  Search s;
  std::vector<int> qSlot;
  for (int i = 0, n = (int)vpFrom.size(); i < n; ++i) {
    MapPoint* pMP = vpFrom[i];
    if (!pMP || vbAlready[i]) continue;
    if (pMP->isBad()) continue;
    const Eigen::Vector3f p3Dw = pMP->GetWorldPos();
    const Eigen::Vector3f p3Df = Tfw * p3Dw;
    const Eigen::Vector3f p3Dt = S * p3Df;
    if (p3Dt(2) < 0.0) continue;                                   // depth must be positive
    const float invz = 1.0 / p3Dt(2);
    const float x = p3Dt(0) * invz, y = p3Dt(1) * invz;
    const float u = fx * x + cx, v = fy * y + cy;
    if (!to_kf->IsInImage(u, v)) continue;
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    const float dist3D = std::sqrt(p3Dt(0) * p3Dt(0) + p3Dt(1) * p3Dt(1) + p3Dt(2) * p3Dt(2));
    if (dist3D < minDistance || dist3D > maxDistance) continue;   // inside the scale invariance region
    const int nPredictedLevel = pMP->PredictScale(dist3D, to_kf);
    const float radius = th * to_kf->mvScaleFactors[nPredictedLevel];
    s.add(pMP->GetDescriptor(), u, v, radius, std::max(nPredictedLevel - 1, 0), nPredictedLevel);
    qSlot.push_back(i);
  }
  if (!device_search(s, t)) return false;
  for (int q = 0; q < s.nq(); ++q)
    if (s.best_idx[q] >= 0 && s.best_dist[q] <= th_high) vnMatch[qSlot[q]] = s.best_idx[q];
  return true;
}

}  // namespace

// src/ORBmatcher.cc:1457-1674: both directions as one batched device search each, then the agreement check (:1659-1673).
int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const Sophus::Sim3f& S12, const float th) {
  const Sophus::SE3f T1w = pKF1->GetPose();
  const Sophus::SE3f T2w = pKF2->GetPose();
  const Sophus::Sim3f S21 = S12.inverse();
  const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
  const int N1 = (int)vpMapPoints1.size();
  const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
  const int N2 = (int)vpMapPoints2.size();
  std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
  for (int i = 0; i < N1; i++) {
    MapPoint* pMP = vpMatches12[i];
    if (pMP) {
      vbAlreadyMatched1[i] = true;
      const int idx2 = std::get<0>(pMP->GetIndexInKeyFrame(pKF2));
      if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
    }
  }
  std::vector<int> vnMatch1(N1, -1), vnMatch2(N2, -1);
  // the reference projects with pKF1's fx fy cx cy in BOTH directions (:1459-1462, 1520-1521, 1600-1601)
  const float fx = pKF1->fx, fy = pKF1->fy, cx = pKF1->cx, cy = pKF1->cy;
  if (!sim3_direction(pKF2, fx, fy, cx, cy, T1w, S21, vpMapPoints1, vbAlreadyMatched1, N2, th, TH_HIGH, vnMatch1)) return 0;
  if (!sim3_direction(pKF1, fx, fy, cx, cy, T2w, S12, vpMapPoints2, vbAlreadyMatched2, N1, th, TH_HIGH, vnMatch2)) return 0;
  int nFound = 0;
  for (int i1 = 0; i1 < N1; i1++) {
    const int idx2 = vnMatch1[i1];
    if (idx2 >= 0 && vnMatch2[idx2] == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; nFound++; }
  }
  return nFound;
}

namespace {

// one batched device search over explicit candidate lists (osh_orb_upload with cand_off / cand_idx): best / second-best distance of
// every query among its list, positions in list order
bool device_search_lists(const std::vector<uint8_t>& qdesc, const cv::Mat& train, int n_train, const std::vector<int32_t>& off,
                         const std::vector<int32_t>& idx, std::vector<int32_t>& best_idx, std::vector<int32_t>& best_dist,
                         std::vector<int32_t>& second_dist, std::vector<int32_t>& second_idx) {
  const int nq = (int)off.size() - 1;
  best_idx.assign(nq, -1); best_dist.assign(nq, 256); second_dist.assign(nq, 256); second_idx.assign(nq, -1);
  if (nq <= 0) return true;
  osh_orb_ctx* ctx = thread_ctx();
  if (!ctx) return false;
  const int64_t base = 0;
  const int32_t none = 0;
  osh_orb_batch b;
  b.n_pairs = 1; b.n_query = nq; b.n_train = n_train;
  b.query_desc = qdesc.data(); b.train_desc = train.ptr<uint8_t>(0); b.train_level = nullptr;
  b.cand_off = off.data(); b.cand_idx = idx.empty() ? &none : idx.data(); b.pair_cand_base = &base;
  std::vector<int32_t> lv1(nq), lv2(nq);
  if (osh_orb_upload(ctx, &b) != OSH_OK || osh_orb_match(ctx) != OSH_OK ||
      osh_orb_download(ctx, best_idx.data(), best_dist.data(), second_dist.data(), lv1.data(), lv2.data(), second_idx.data()) != OSH_OK) {
    std::fprintf(stderr, "ORBmatcher: device search failed: %s\n", osh_last_error());
    return false;
  }
  return true;
}

}  // namespace

// src/ORBmatcher.cc:1148-1338.  Every map point is projected into the keyframe and its candidate list (the features in the search
// radius that pass the level and the reprojection-chi2 gates, in GetFeaturesInArea's order) is formed up front: none of that depends
// on what the loop does to earlier points.  ONE batched device search gives the best candidate of every point; the part that is
// order dependent -- "already in the keyframe", Replace in either direction, AddObservation / AddMapPoint -- is replayed in the
// reference's order on the map itself.
namespace {

// osh_orb_list_distances over explicit candidate lists: the Hamming distance of every (query, candidate) entry, in list order
bool device_list_distances(const std::vector<uint8_t>& qdesc, const cv::Mat& train, int n_train, const std::vector<int32_t>& off,
                           const std::vector<int32_t>& idx, std::vector<int32_t>& dist) {
  const int nq = (int)off.size() - 1;
  dist.assign(idx.size(), 256);
  if (nq <= 0 || idx.empty()) return true;
  osh_orb_ctx* ctx = thread_ctx();
  if (!ctx) return false;
  const int64_t base = 0;
  osh_orb_batch b;
  b.n_pairs = 1; b.n_query = nq; b.n_train = n_train;
  b.query_desc = qdesc.data(); b.train_desc = train.ptr<uint8_t>(0); b.train_level = nullptr;
  b.cand_off = off.data(); b.cand_idx = idx.data(); b.pair_cand_base = &base;
  if (osh_orb_upload(ctx, &b) != OSH_OK || osh_orb_list_distances(ctx, dist.data()) != OSH_OK) {
    std::fprintf(stderr, "ORBmatcher: device distances failed: %s\n", osh_last_error());
    return false;
  }
  return true;
}

}  // namespace

// src/ORBmatcher.cc:648-763 (monocular map initialisation).  The candidates of every level-0 keypoint of F1 are the level-0 features
// of F2 in the window around its previous match; the device returns the distance of every (keypoint, candidate) entry in one launch.
// The loop itself is order dependent -- a candidate is skipped while the distance it was last matched with is not larger
// (vMatchedDistance), and a new match displaces the old one of the same feature -- and is replayed as written.
int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize) {
  int nmatches = 0;
  vnMatches12 = std::vector<int>(F1.mvKeysUn.size(), -1);
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
  const float factor = 1.0f / HISTO_LENGTH;
  std::vector<int> vMatchedDistance(F2.mvKeysUn.size(), INT_MAX);
  std::vector<int> vnMatches21(F2.mvKeysUn.size(), -1);
  // queries: the level-0 keypoints with a non-empty window, in order
  std::vector<int> q1;
  std::vector<uint8_t> qdesc;
  std::vector<int32_t> off(1, 0), idx;
  for (size_t i1 = 0, iend1 = F1.mvKeysUn.size(); i1 < iend1; i1++) {
    const int level1 = F1.mvKeysUn[i1].octave;
    if (level1 > 0) continue;
    const std::vector<size_t> vIndices2 = F2.GetFeaturesInArea(vbPrevMatched[i1].x, vbPrevMatched[i1].y, (float)windowSize, level1, level1);
    if (vIndices2.empty()) continue;
    for (size_t i2 : vIndices2) idx.push_back((int32_t)i2);
    off.push_back((int32_t)idx.size());
    q1.push_back((int)i1);
    qdesc.insert(qdesc.end(), F1.mDescriptors.ptr<uint8_t>((int)i1), F1.mDescriptors.ptr<uint8_t>((int)i1) + 32);
  }
  std::vector<int32_t> dist;
  if (!device_list_distances(qdesc, F2.mDescriptors, F2.mDescriptors.rows, off, idx, dist)) return 0;
  for (size_t q = 0; q < q1.size(); ++q) {
    const int i1 = q1[q];
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (int e = off[q]; e < off[q + 1]; ++e) {
      const int i2 = idx[e], d = dist[e];
      if (vMatchedDistance[i2] <= d) continue;
      if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx2 = i2; }
      else if (d < bestDist2) bestDist2 = d;
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * mfNNratio) {
        if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        vnMatches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (mbCheckOrientation) {
          float rot = F1.mvKeysUn[i1].angle - F2.mvKeysUn[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(i1);
        }
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        const int idx1 = rotHist[i][j];
        if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
      }
    }
  }
  for (size_t i1 = 0, iend1 = vnMatches12.size(); i1 < iend1; i1++)      // update prev matched
    if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt;
  return nmatches;
}

// src/ORBmatcher.cc:907-1146.  No step of this search depends on an earlier one (vbMatched2 is read but never set), so every
// unmatched feature of keyframe 1 is a query whose candidates are the unmatched features of keyframe 2 in the same vocabulary
// node.  The device returns the distance of every (query, candidate) entry in one launch; the choice among the candidates --
// the reference's loop with its `dist > bestDist` rule (a later candidate at the same distance wins), the epipole test and the
// epipolar constraint, which is a virtual call on the camera object -- runs on the host over the entries within TH_LOW only.
int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                                       const bool bOnlyStereo, const bool bCoarse) {
  const DBoW2::FeatureVector& vFeatVec1 = pKF1->mFeatVec;
  const DBoW2::FeatureVector& vFeatVec2 = pKF2->mFeatVec;
  // Compute epipole in second image
  Sophus::SE3f T1w = pKF1->GetPose();
  Sophus::SE3f T2w = pKF2->GetPose();
  Sophus::SE3f Tw2 = pKF2->GetPoseInverse();
  Eigen::Vector3f Cw = pKF1->GetCameraCenter();
  Eigen::Vector3f C2 = T2w * Cw;
  Eigen::Vector2f ep = pKF2->mpCamera->project(C2);
  Sophus::SE3f T12;
  Sophus::SE3f Tll, Tlr, Trl, Trr;
  Eigen::Matrix3f R12;
  Eigen::Vector3f t12;
  GeometricCamera *pCamera1 = pKF1->mpCamera, *pCamera2 = pKF2->mpCamera;
  if (!pKF1->mpCamera2 && !pKF2->mpCamera2) {
    T12 = T1w * Tw2;
    R12 = T12.rotationMatrix();
    t12 = T12.translation();
  } else {
    Sophus::SE3f Tr1w = pKF1->GetRightPose();
    Sophus::SE3f Twr2 = pKF2->GetRightPoseInverse();
    Tll = T1w * Tw2; Tlr = T1w * Twr2; Trl = Tr1w * Tw2; Trr = Tr1w * Twr2;
  }
  Eigen::Matrix3f Rll = Tll.rotationMatrix(), Rlr = Tlr.rotationMatrix(), Rrl = Trl.rotationMatrix(), Rrr = Trr.rotationMatrix();
  Eigen::Vector3f tll = Tll.translation(), tlr = Tlr.translation(), trl = Trl.translation(), trr = Trr.translation();

  // ---- queries and candidate lists in the order of the vocabulary walk
  std::vector<int> q1;
  std::vector<uint8_t> qdesc;
  std::vector<int32_t> off(1, 0), idx;
  DBoW2::FeatureVector::const_iterator f1it = vFeatVec1.begin(), f1end = vFeatVec1.end();
  DBoW2::FeatureVector::const_iterator f2it = vFeatVec2.begin(), f2end = vFeatVec2.end();
  while (f1it != f1end && f2it != f2end) {
    if (f1it->first == f2it->first) {
      for (size_t i1 = 0, iend1 = f1it->second.size(); i1 < iend1; i1++) {
        const size_t idx1 = f1it->second[i1];
        if (pKF1->GetMapPoint(idx1)) continue;                       // already a MapPoint: skip
        const bool bStereo1 = (!pKF1->mpCamera2 && pKF1->mvuRight[idx1] >= 0);
        if (bOnlyStereo && !bStereo1) continue;
        for (size_t i2 = 0, iend2 = f2it->second.size(); i2 < iend2; i2++) {
          const size_t idx2 = f2it->second[i2];
          if (pKF2->GetMapPoint(idx2)) continue;                     // already matched / already a MapPoint
          const bool bStereo2 = (!pKF2->mpCamera2 && pKF2->mvuRight[idx2] >= 0);
          if (bOnlyStereo && !bStereo2) continue;
          idx.push_back((int32_t)idx2);
        }
        off.push_back((int32_t)idx.size());
        q1.push_back((int)idx1);
        qdesc.insert(qdesc.end(), pKF1->mDescriptors.ptr<uint8_t>((int)idx1), pKF1->mDescriptors.ptr<uint8_t>((int)idx1) + 32);
      }
      f1it++; f2it++;
    } else if (f1it->first < f2it->first) {
      f1it = vFeatVec1.lower_bound(f2it->first);
    } else {
      f2it = vFeatVec2.lower_bound(f1it->first);
    }
  }
  std::vector<int32_t> dist;
  if (!device_list_distances(qdesc, pKF2->mDescriptors, pKF2->mDescriptors.rows, off, idx, dist)) return 0;

  int nmatches = 0;
  std::vector<int> vMatches12(pKF1->N, -1);
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
  const float factor = 1.0f / HISTO_LENGTH;
  for (size_t q = 0; q < q1.size(); ++q) {
    const size_t idx1 = (size_t)q1[q];
    const bool bStereo1 = (!pKF1->mpCamera2 && pKF1->mvuRight[idx1] >= 0);
    const cv::KeyPoint& kp1 = (pKF1->NLeft == -1) ? pKF1->mvKeysUn[idx1]
                              : ((int)idx1 < pKF1->NLeft) ? pKF1->mvKeys[idx1] : pKF1->mvKeysRight[idx1 - pKF1->NLeft];
    const bool bRight1 = (pKF1->NLeft == -1 || (int)idx1 < pKF1->NLeft) ? false : true;
    int bestDist = TH_LOW;
    int bestIdx2 = -1;
    for (int e = off[q]; e < off[q + 1]; ++e) {
      const size_t idx2 = (size_t)idx[e];
      const int d = dist[e];
      if (d > TH_LOW || d > bestDist) continue;
      const bool bStereo2 = (!pKF2->mpCamera2 && pKF2->mvuRight[idx2] >= 0);
      const cv::KeyPoint& kp2 = (pKF2->NLeft == -1) ? pKF2->mvKeysUn[idx2]
                                : ((int)idx2 < pKF2->NLeft) ? pKF2->mvKeys[idx2] : pKF2->mvKeysRight[idx2 - pKF2->NLeft];
      const bool bRight2 = (pKF2->NLeft == -1 || (int)idx2 < pKF2->NLeft) ? false : true;
      if (!bStereo1 && !bStereo2 && !pKF1->mpCamera2) {
        const float distex = ep(0) - kp2.pt.x;
        const float distey = ep(1) - kp2.pt.y;
        if (distex * distex + distey * distey < 100 * pKF2->mvScaleFactors[kp2.octave]) continue;
      }
      if (pKF1->mpCamera2 && pKF2->mpCamera2) {
        if (bRight1 && bRight2) { R12 = Rrr; t12 = trr; T12 = Trr; pCamera1 = pKF1->mpCamera2; pCamera2 = pKF2->mpCamera2; }
        else if (bRight1 && !bRight2) { R12 = Rrl; t12 = trl; T12 = Trl; pCamera1 = pKF1->mpCamera2; pCamera2 = pKF2->mpCamera; }
        else if (!bRight1 && bRight2) { R12 = Rlr; t12 = tlr; T12 = Tlr; pCamera1 = pKF1->mpCamera; pCamera2 = pKF2->mpCamera2; }
        else { R12 = Rll; t12 = tll; T12 = Tll; pCamera1 = pKF1->mpCamera; pCamera2 = pKF2->mpCamera; }
      }
      if (bCoarse || pCamera1->epipolarConstrain(pCamera2, kp1, kp2, R12, t12, pKF1->mvLevelSigma2[kp1.octave], pKF2->mvLevelSigma2[kp2.octave])) {
        bestIdx2 = (int)idx2;
        bestDist = d;
      }
    }
    if (bestIdx2 >= 0) {
      const cv::KeyPoint& kp2 = (pKF2->NLeft == -1) ? pKF2->mvKeysUn[bestIdx2]
                                : (bestIdx2 < pKF2->NLeft) ? pKF2->mvKeys[bestIdx2] : pKF2->mvKeysRight[bestIdx2 - pKF2->NLeft];
      vMatches12[idx1] = bestIdx2;
      nmatches++;
      if (mbCheckOrientation) {
        float rot = kp1.angle - kp2.angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back((int)idx1);
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { vMatches12[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  vMatchedPairs.clear();
  vMatchedPairs.reserve(nmatches);
  for (size_t i = 0, iend = vMatches12.size(); i < iend; i++) {
    if (vMatches12[i] < 0) continue;
    vMatchedPairs.push_back(std::make_pair(i, (size_t)vMatches12[i]));
  }
  return nmatches;
}

namespace {

// The part of both Fuse overloads that does not depend on what their loops do to the map: per map point the projection gates
// (src/ORBmatcher.cc:1196-1260 / 1372-1412) and the candidate list -- the features of GetFeaturesInArea, in its order, that pass the
// level gate and (first overload, chi2_gate) the reprojection-error gate -- then ONE batched device search over the lists.
// qOf[i] = query of point i or -1; best / bestD by query.
bool fuse_search(KeyFrame* pKF, const Sophus::SE3f& Tcw, const Eigen::Vector3f& Ow, GeometricCamera* pCamera, const std::vector<MapPoint*>& vpMapPoints,
                 const float th, const bool bRight, const bool chi2_gate, std::vector<int>& qOf, std::vector<int32_t>& best, std::vector<int32_t>& bestD) {
  const float& bf = pKF->mbf;
  const int nMPs = (int)vpMapPoints.size();
  qOf.assign(nMPs, -1);
  std::vector<uint8_t> qdesc;
  std::vector<int32_t> off(1, 0), idx;
  int nq = 0;
  for (int i = 0; i < nMPs; i++) {
    MapPoint* pMP = vpMapPoints[i];
    if (!pMP) continue;
    const Eigen::Vector3f p3Dw = pMP->GetWorldPos();
    const Eigen::Vector3f p3Dc = Tcw * p3Dw;
    if (p3Dc(2) < 0.0f) continue;                                   // depth must be positive
    const float invz = 1 / p3Dc(2);
    const Eigen::Vector2f uv = pCamera->project(p3Dc);
    if (!pKF->IsInImage(uv(0), uv(1))) continue;                    // point must be inside the image
    const float ur = uv(0) - bf * invz;
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    const float px = p3Dw(0) - Ow(0), py = p3Dw(1) - Ow(1), pz = p3Dw(2) - Ow(2);
    const float dist3D = std::sqrt(px * px + py * py + pz * pz);
    if (dist3D < minDistance || dist3D > maxDistance) continue;     // inside the scale pyramid of the image
    const Eigen::Vector3f Pn = pMP->GetNormal();
    if (px * Pn(0) + py * Pn(1) + pz * Pn(2) < 0.5 * dist3D) continue;   // viewing angle below 60 degrees
    const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
    const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
    const std::vector<size_t> vIndices = pKF->GetFeaturesInArea(uv(0), uv(1), radius, bRight);
    if (vIndices.empty()) continue;
    for (size_t k : vIndices) {
      const cv::KeyPoint& kp = (!chi2_gate || pKF->NLeft == -1) ? pKF->mvKeysUn[k] : (!bRight) ? pKF->mvKeys[k] : pKF->mvKeysRight[k];
      const int& kpLevel = kp.octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      if (chi2_gate) {
        const float ex = uv(0) - kp.pt.x, ey = uv(1) - kp.pt.y;
        if (pKF->mvuRight[k] >= 0) {                                // reprojection error in stereo
          const float er = ur - pKF->mvuRight[k];
          const float e2 = ex * ex + ey * ey + er * er;
          if (e2 * pKF->mvInvLevelSigma2[kpLevel] > 7.8) continue;
        } else {
          const float e2 = ex * ex + ey * ey;
          if (e2 * pKF->mvInvLevelSigma2[kpLevel] > 5.99) continue;
        }
      }
      idx.push_back((int32_t)(bRight ? k + pKF->NLeft : k));
    }
    off.push_back((int32_t)idx.size());
    const cv::Mat dMP = pMP->GetDescriptor();
    qdesc.insert(qdesc.end(), dMP.ptr<uint8_t>(0), dMP.ptr<uint8_t>(0) + 32);
    qOf[i] = nq++;
  }
  std::vector<int32_t> secondD, secondI;
  best.clear(); bestD.clear();
  return nq == 0 || device_search_lists(qdesc, pKF->mDescriptors, pKF->mDescriptors.rows, off, idx, best, bestD, secondD, secondI);
}

}  // namespace

int ORBmatcher::Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th, const bool bRight) {
  GeometricCamera* pCamera;
  Sophus::SE3f Tcw;
  Eigen::Vector3f Ow;
  if (bRight) { Tcw = pKF->GetRightPose(); Ow = pKF->GetRightCameraCenter(); pCamera = pKF->mpCamera2; }
  else { Tcw = pKF->GetPose(); Ow = pKF->GetCameraCenter(); pCamera = pKF->mpCamera; }
  const int nMPs = (int)vpMapPoints.size();
  std::vector<int> qOf;
  std::vector<int32_t> best, bestD;
  if (!fuse_search(pKF, Tcw, Ow, pCamera, vpMapPoints, th, bRight, true, qOf, best, bestD)) return 0;

  int nFused = 0;
  for (int i = 0; i < nMPs; i++) {
    MapPoint* pMP = vpMapPoints[i];
    if (!pMP) continue;
    if (pMP->isBad()) continue;
    else if (pMP->IsInKeyFrame(pKF)) continue;
    if (qOf[i] < 0) continue;
    const int bestDist = bestD[qOf[i]], bestIdx = best[qOf[i]];
    if (bestIdx >= 0 && bestDist <= TH_LOW) {                       // already a MapPoint there: replace, otherwise add the measurement
      MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) {
          if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
          else pMPinKF->Replace(pMP);
        }
      } else {
        pMP->AddObservation(pKF, bestIdx);
        pKF->AddMapPoint(pMP, bestIdx);
      }
      nFused++;
    }
  }
  return nFused;
}

// src/ORBmatcher.cc:1340-1455 (loop closing / map merging): the same search through a Sim3 pose, no reprojection-error gate; a point
// whose best feature already holds a map point is reported in vpReplacePoint instead of being replaced.
int ORBmatcher::Fuse(KeyFrame* pKF, Sophus::Sim3f& Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint) {
  const Eigen::Vector3f ts = Scw.translation();
  const float sc = Scw.scale();
  const Sophus::SE3f Tcw(Scw.rotationMatrix(), Eigen::Vector3f(ts(0) / sc, ts(1) / sc, ts(2) / sc));
  const Eigen::Vector3f Ow = Tcw.inverse().translation();
  const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();    // as found at entry: the loop does not update it
  const int nPoints = (int)vpPoints.size();
  std::vector<int> qOf;
  std::vector<int32_t> best, bestD;
  if (!fuse_search(pKF, Tcw, Ow, pKF->mpCamera, vpPoints, th, false, false, qOf, best, bestD)) return 0;
  int nFused = 0;
  for (int iMP = 0; iMP < nPoints; iMP++) {
    MapPoint* pMP = vpPoints[iMP];
    if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
    if (qOf[iMP] < 0) continue;
    const int bestDist = bestD[qOf[iMP]], bestIdx = best[qOf[iMP]];
    if (bestIdx >= 0 && bestDist <= TH_LOW) {
      MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
      } else {
        pMP->AddObservation(pKF, bestIdx);
        pKF->AddMapPoint(pMP, bestIdx);
      }
      nFused++;
    }
  }
  return nFused;
}

// src/ORBmatcher.cc:223-420.  The candidate loops of every keyframe feature run as batched device searches over the feature
// lists of its vocabulary node (one search for the left / only camera, one for the right camera of a fisheye stereo frame); the
// "frame feature already matched" rule (:266-268) makes the loop sequential, so the queries are replayed in the reference's
// order and a query whose best or second-best candidate has been taken in the meantime is scanned again with the current matches.
int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
  const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
  vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(nullptr));
  const DBoW2::FeatureVector& vFeatVecKF = pKF->mFeatVec;
  const bool rig = F.Nleft != -1;
  // queries in visiting order, each with the frame-feature list of its node
  std::vector<int> qKF;
  std::vector<const std::vector<unsigned int>*> qList;
  DBoW2::FeatureVector::const_iterator KFit = vFeatVecKF.begin(), KFend = vFeatVecKF.end();
  DBoW2::FeatureVector::const_iterator Fit = F.mFeatVec.begin(), Fend = F.mFeatVec.end();
  while (KFit != KFend && Fit != Fend) {
    if (KFit->first == Fit->first) {
      for (const unsigned int realIdxKF : KFit->second) {
        MapPoint* pMP = vpMapPointsKF[realIdxKF];
        if (!pMP || pMP->isBad()) continue;
        qKF.push_back((int)realIdxKF);
        qList.push_back(&Fit->second);
      }
      KFit++; Fit++;
    } else if (KFit->first < Fit->first) {
      KFit = vFeatVecKF.lower_bound(Fit->first);
    } else {
      Fit = F.mFeatVec.lower_bound(KFit->first);
    }
  }
  const int nq = (int)qKF.size();
  if (nq == 0) return 0;
  std::vector<uint8_t> qdesc((size_t)nq * 32);
  std::vector<int32_t> offL(1, 0), idxL, offR(1, 0), idxR;
  for (int q = 0; q < nq; ++q) {
    std::memcpy(&qdesc[(size_t)q * 32], pKF->mDescriptors.ptr<uint8_t>(qKF[q]), 32);
    for (const unsigned int iF : *qList[q]) {
      if (!rig || (int)iF < F.Nleft) idxL.push_back((int32_t)iF); else idxR.push_back((int32_t)iF);
    }
    offL.push_back((int32_t)idxL.size()); offR.push_back((int32_t)idxR.size());
  }
  std::vector<int32_t> bL, dL, sL, siL, bR, dR, sR, siR;
  if (!device_search_lists(qdesc, F.mDescriptors, F.N, offL, idxL, bL, dL, sL, siL)) return 0;
  if (rig && !device_search_lists(qdesc, F.mDescriptors, F.N, offR, idxR, bR, dR, sR, siR)) return 0;

  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
  const float factor = 1.0f / HISTO_LENGTH;
  // the reference's candidate loop over one side of the node's list with the current matches (for contested queries only)
  auto rescan = [&](int q, bool right, int& bestDist1, int& bestIdxF, int& bestDist2) {
    bestDist1 = 256; bestIdxF = -1; bestDist2 = 256;
    const uint32_t* qd = reinterpret_cast<const uint32_t*>(&qdesc[(size_t)q * 32]);
    for (const unsigned int realIdxF : *qList[q]) {
      if (rig && (((int)realIdxF >= F.Nleft) != right)) continue;
      if (vpMapPointMatches[realIdxF]) continue;
      const uint32_t* td = F.mDescriptors.ptr<uint32_t>((int)realIdxF);
      int dist = 0;
      for (int k = 0; k < 8; ++k) dist += __builtin_popcount(qd[k] ^ td[k]);
      if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
  };
  auto histo = [&](int realIdxKF, int idxF) {
    const cv::KeyPoint& kp = (!pKF->mpCamera2) ? pKF->mvKeysUn[realIdxKF]
                             : (realIdxKF >= pKF->NLeft) ? pKF->mvKeysRight[realIdxKF - pKF->NLeft] : pKF->mvKeys[realIdxKF];
    const cv::KeyPoint& Fkp = (!rig) ? F.mvKeys[idxF] : (idxF >= F.Nleft) ? F.mvKeysRight[idxF - F.Nleft] : F.mvKeys[idxF];
    float rot = kp.angle - Fkp.angle;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    rotHist[bin].push_back(idxF);
  };
  for (int q = 0; q < nq; ++q) {
    MapPoint* pMP = vpMapPointsKF[qKF[q]];
    int bestDist1 = dL[q], bestIdxF = bL[q], bestDist2 = sL[q];
    if ((bestIdxF >= 0 && vpMapPointMatches[bestIdxF]) || (siL[q] >= 0 && vpMapPointMatches[siL[q]])) rescan(q, false, bestDist1, bestIdxF, bestDist2);
    int bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
    if (rig) {
      bestDist1R = dR[q]; bestIdxFR = bR[q]; bestDist2R = sR[q];
      if ((bestIdxFR >= 0 && vpMapPointMatches[bestIdxFR]) || (siR[q] >= 0 && vpMapPointMatches[siR[q]])) rescan(q, true, bestDist1R, bestIdxFR, bestDist2R);
    }
    if (bestDist1 <= TH_LOW) {
      if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
        vpMapPointMatches[bestIdxF] = pMP;
        if (mbCheckOrientation) histo(qKF[q], bestIdxF);
        nmatches++;
      }
      if (bestDist1R <= TH_LOW) {   // the right-camera best is taken without a ratio test ("|| true", :352)
        vpMapPointMatches[bestIdxFR] = pMP;
        if (mbCheckOrientation) histo(qKF[q], bestIdxFR);
        nmatches++;
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        vpMapPointMatches[rotHist[i][j]] = static_cast<MapPoint*>(nullptr);
        nmatches--;
      }
    }
  }
  return nmatches;
}

// src/ORBmatcher.cc:765-905: same scheme as the keyframe / frame variant.  The static candidate filters (keyframe 2's feature holds a
// good map point, index below mvKeysUn.size() on a fisheye rig keyframe) are applied when the lists are built; vbMatched2 is the
// sequential part that the ordered replay reproduces.
int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
  const std::vector<cv::KeyPoint>& vKeysUn1 = pKF1->mvKeysUn;
  const DBoW2::FeatureVector& vFeatVec1 = pKF1->mFeatVec;
  const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
  const std::vector<cv::KeyPoint>& vKeysUn2 = pKF2->mvKeysUn;
  const DBoW2::FeatureVector& vFeatVec2 = pKF2->mFeatVec;
  const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
  vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(nullptr));
  std::vector<bool> vbMatched2(vpMapPoints2.size(), false);
  std::vector<int> q1;
  std::vector<int32_t> off(1, 0), idx;
  DBoW2::FeatureVector::const_iterator f1it = vFeatVec1.begin(), f1end = vFeatVec1.end();
  DBoW2::FeatureVector::const_iterator f2it = vFeatVec2.begin(), f2end = vFeatVec2.end();
  while (f1it != f1end && f2it != f2end) {
    if (f1it->first == f2it->first) {
      for (const unsigned int idx1 : f1it->second) {
        if (pKF1->NLeft != -1 && idx1 >= pKF1->mvKeysUn.size()) continue;
        MapPoint* pMP1 = vpMapPoints1[idx1];
        if (!pMP1 || pMP1->isBad()) continue;
        q1.push_back((int)idx1);
        for (const unsigned int idx2 : f2it->second) {
          if (pKF2->NLeft != -1 && idx2 >= pKF2->mvKeysUn.size()) continue;
          MapPoint* pMP2 = vpMapPoints2[idx2];
          if (!pMP2 || pMP2->isBad()) continue;
          idx.push_back((int32_t)idx2);
        }
        off.push_back((int32_t)idx.size());
      }
      f1it++; f2it++;
    } else if (f1it->first < f2it->first) {
      f1it = vFeatVec1.lower_bound(f2it->first);
    } else {
      f2it = vFeatVec2.lower_bound(f1it->first);
    }
  }
  const int nq = (int)q1.size();
  if (nq == 0) return 0;
  std::vector<uint8_t> qdesc((size_t)nq * 32);
  for (int q = 0; q < nq; ++q) std::memcpy(&qdesc[(size_t)q * 32], pKF1->mDescriptors.ptr<uint8_t>(q1[q]), 32);
  std::vector<int32_t> b2, d1, d2, si;
  if (!device_search_lists(qdesc, pKF2->mDescriptors, (int)vpMapPoints2.size(), off, idx, b2, d1, d2, si)) return 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
  const float factor = 1.0f / HISTO_LENGTH;
  int nmatches = 0;
  for (int q = 0; q < nq; ++q) {
    const int idx1 = q1[q];
    int bestDist1 = d1[q], bestIdx2 = b2[q], bestDist2 = d2[q];
    if ((bestIdx2 >= 0 && vbMatched2[bestIdx2]) || (si[q] >= 0 && vbMatched2[si[q]])) {
      bestDist1 = 256; bestIdx2 = -1; bestDist2 = 256;
      const uint32_t* qd = reinterpret_cast<const uint32_t*>(&qdesc[(size_t)q * 32]);
      for (int c = off[q]; c < off[q + 1]; ++c) {
        const int i2 = idx[c];
        if (vbMatched2[i2]) continue;
        const uint32_t* td = pKF2->mDescriptors.ptr<uint32_t>(i2);
        int dist = 0;
        for (int k = 0; k < 8; ++k) dist += __builtin_popcount(qd[k] ^ td[k]);
        if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = i2; }
        else if (dist < bestDist2) bestDist2 = dist;
      }
    }
    if (bestDist1 < TH_LOW) {
      if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
        vpMatches12[idx1] = vpMapPoints2[bestIdx2];
        vbMatched2[bestIdx2] = true;
        if (mbCheckOrientation) {
          float rot = vKeysUn1[idx1].angle - vKeysUn2[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(idx1);
        }
        nmatches++;
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        vpMatches12[rotHist[i][j]] = static_cast<MapPoint*>(nullptr);
        nmatches--;
      }
    }
  }
  return nmatches;
}

}  // namespace ORB_SLAM3
