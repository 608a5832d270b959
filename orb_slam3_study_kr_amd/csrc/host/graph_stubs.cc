// graph_stubs.cc -- bodies of the few KeyFrame / MapPoint / Frame methods of the minimal test doubles in include/
// (a real ORB-SLAM3 tree supplies its own; only Optimizer.cc / ORBmatcher.cc are the drop-in).
#include <algorithm>
#include <cmath>
#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"

namespace ORB_SLAM3 {

std::mutex MapPoint::mGlobalMutex;

// src/KeyFrame.cc:109-122: cache Rcw, Twc and the IMU position Owb = Rwc * tcb + twc
void KeyFrame::SetPose(const Sophus::SE3f& Tcw) {
  mTcw = Tcw;
  mRcw = mTcw.rotationMatrix();
  mTwc = mTcw.inverse();
  if (mImuCalib.mbIsSet) mOwb = mTwc * mImuCalib.mTcb.translation();
  ++mnPoseSets;
}

// src/KeyFrame.cc:336-349
std::set<MapPoint*> KeyFrame::GetMapPoints() {
  std::set<MapPoint*> s;
  for (MapPoint* pMP : mvpMapPoints) {
    if (!pMP) continue;
    if (!pMP->isBad()) s.insert(pMP);
  }
  return s;
}

// src/KeyFrame.cc:309-332 (index looked up through the point's observation)
void KeyFrame::EraseMapPointMatch(MapPoint* pMP) {
  for (size_t i = 0; i < mvpMapPoints.size(); ++i)
    if (mvpMapPoints[i] == pMP) mvpMapPoints[i] = nullptr;
}

void MapPoint::AddObservation(KeyFrame* pKF, int idx) {
  // src/MapPoint.cc:140-165: the index selects the left / right slot of the (left, right) tuple of a fisheye rig
  std::tuple<int, int> indexes = mObservations.count(pKF) ? mObservations[pKF] : std::tuple<int, int>(-1, -1);
  if (pKF->NLeft != -1 && idx >= pKF->NLeft) std::get<1>(indexes) = idx;
  else std::get<0>(indexes) = idx;
  mObservations[pKF] = indexes;
  if (!pKF->mpCamera2 && pKF->mvuRight[idx] >= 0) nObs += 2;
  else nObs++;
}

// src/MapPoint.cc:248-297: this point hands its observations to pMP and goes bad
void MapPoint::Replace(MapPoint* pMP) {
  if (pMP->mnId == this->mnId) return;
  std::map<KeyFrame*, std::tuple<int, int>> obs = mObservations;
  mObservations.clear();
  mbBad = true;
  const int nvisible = mnVisible, nfound = mnFound;
  mpReplaced = pMP;
  for (auto& ob : obs) {
    KeyFrame* pKF = ob.first;
    const int leftIndex = std::get<0>(ob.second), rightIndex = std::get<1>(ob.second);
    if (!pMP->IsInKeyFrame(pKF)) {
      if (leftIndex != -1) { pKF->ReplaceMapPointMatch(leftIndex, pMP); pMP->AddObservation(pKF, leftIndex); }
      if (rightIndex != -1) { pKF->ReplaceMapPointMatch(rightIndex, pMP); pMP->AddObservation(pKF, rightIndex); }
    } else {
      if (leftIndex != -1) pKF->EraseMapPointMatch(leftIndex);
      if (rightIndex != -1) pKF->EraseMapPointMatch(rightIndex);
    }
  }
  pMP->IncreaseFound(nfound);
  pMP->IncreaseVisible(nvisible);
  pMP->ComputeDistinctiveDescriptors();
}

// src/MapPoint.cc:168-201: drop the observation; a point left with <= 2 observations goes bad
// src/MapPoint.cc:531-546 (float ratio, float log: `log(ratio)` resolves to the float overload there)
int MapPoint::PredictScale(const float& currentDist, Frame* pF) {
  const float ratio = mfMaxDistance / currentDist;
  int nScale = (int)std::ceil(std::log(ratio) / pF->mfLogScaleFactor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= pF->mnScaleLevels) nScale = pF->mnScaleLevels - 1;
  return nScale;
}

int MapPoint::PredictScale(const float& currentDist, KeyFrame* pKF) {
  const float ratio = mfMaxDistance / currentDist;
  int nScale = (int)std::ceil(std::log(ratio) / pKF->mfLogScaleFactor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= pKF->mnScaleLevels) nScale = pKF->mnScaleLevels - 1;
  return nScale;
}

// src/KeyFrame.cc:704-745, NLeft == -1 layout
std::vector<size_t> KeyFrame::GetFeaturesInArea(const float& x, const float& y, const float& r, const bool bRight) const {
  (void)bRight;
  std::vector<size_t> vIndices;
  vIndices.reserve(N);
  const float factorX = r, factorY = r;
  const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - factorX) * mfGridElementWidthInv));
  if (nMinCellX >= mnGridCols) return vIndices;
  const int nMaxCellX = std::min((int)mnGridCols - 1, (int)std::ceil((x - mnMinX + factorX) * mfGridElementWidthInv));
  if (nMaxCellX < 0) return vIndices;
  const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - factorY) * mfGridElementHeightInv));
  if (nMinCellY >= mnGridRows) return vIndices;
  const int nMaxCellY = std::min((int)mnGridRows - 1, (int)std::ceil((y - mnMinY + factorY) * mfGridElementHeightInv));
  if (nMaxCellY < 0) return vIndices;
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const std::vector<size_t>& vCell = mGrid[ix][iy];
      for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
        const cv::KeyPoint& kpUn = mvKeysUn[vCell[j]];
        const float distx = kpUn.pt.x - x;
        const float disty = kpUn.pt.y - y;
        if (std::fabs(distx) < r && std::fabs(disty) < r) vIndices.push_back(vCell[j]);
      }
    }
  return vIndices;
}

void MapPoint::EraseObservation(KeyFrame* pKF) {
  auto it = mObservations.find(pKF);
  if (it == mObservations.end()) return;
  const int leftIndex = std::get<0>(it->second), rightIndex = std::get<1>(it->second);
  if (leftIndex != -1) nObs -= (!pKF->mpCamera2 && pKF->mvuRight[leftIndex] >= 0) ? 2 : 1;
  if (rightIndex != -1) nObs--;
  mObservations.erase(it);
  if (nObs <= 2) mbBad = true;
}

// src/Frame.cc:397-417 + PosInGrid :726-736
void Frame::AssignFeaturesToGrid() {
  for (int i = 0; i < FRAME_GRID_COLS; ++i)
    for (int j = 0; j < FRAME_GRID_ROWS; ++j) { mGrid[i][j].clear(); mGridRight[i][j].clear(); }
  for (int i = 0; i < N; ++i) {
    // src/Frame.cc:406-416: left keypoints (distorted mvKeys on a fisheye rig) into mGrid, right ones into mGridRight
    const cv::KeyPoint& kp = (Nleft == -1) ? mvKeysUn[i] : (i < Nleft) ? mvKeys[i] : mvKeysRight[i - Nleft];
    const int posX = (int)std::round((kp.pt.x - mnMinX) * mfGridElementWidthInv);
    const int posY = (int)std::round((kp.pt.y - mnMinY) * mfGridElementHeightInv);
    if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) continue;
    if (Nleft == -1 || i < Nleft) mGrid[posX][posY].push_back(i);
    else mGridRight[posX][posY].push_back(i - Nleft);
  }
}

// Candidate generator with the semantics of src/Frame.cc:658-722: a square window |dx| < r, |dy| < r
// (strict), optional octave band, results ordered by grid column, then grid row, then insertion order.
std::vector<size_t> Frame::GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel,
                                             const int maxLevel, const bool bRight) const {
  std::vector<size_t> hits;
  hits.reserve(N);
  // first / last grid cell touched along one axis, or false when the window misses the grid
  auto cell_span = [](float centre, float origin, float radius, float inv, int ncell, int& lo, int& hi) {
    lo = std::max(0, (int)std::floor((centre - origin - radius) * inv));
    if (lo >= ncell) return false;
    hi = std::min(ncell - 1, (int)std::ceil((centre - origin + radius) * inv));
    return hi >= 0;
  };
  int c0, c1, r0, r1;
  if (!cell_span(x, mnMinX, r, mfGridElementWidthInv, FRAME_GRID_COLS, c0, c1)) return hits;
  if (!cell_span(y, mnMinY, r, mfGridElementHeightInv, FRAME_GRID_ROWS, r0, r1)) return hits;
  const bool band = (minLevel > 0) || (maxLevel >= 0);
  for (int gc = c0; gc <= c1; ++gc)
    for (int gr = r0; gr <= r1; ++gr)
      for (const size_t k : (!bRight ? mGrid[gc][gr] : mGridRight[gc][gr])) {
        const cv::KeyPoint& kp = (Nleft == -1) ? mvKeysUn[k] : (!bRight) ? mvKeys[k] : mvKeysRight[k];
        if (band && (kp.octave < minLevel || (maxLevel >= 0 && kp.octave > maxLevel))) continue;
        if (std::fabs(kp.pt.x - x) < r && std::fabs(kp.pt.y - y) < r) hits.push_back(k);
      }
  return hits;
}

}  // namespace ORB_SLAM3
