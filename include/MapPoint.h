/* MapPoint.h -- members of ORB_SLAM3::MapPoint used on the hot path (reference include/MapPoint.h;
 * src/MapPoint.cc:118-127,168-201,204,302,426-494,560).  Minimal test double: same names and meaning,
 * no locking (the real getters copy under mMutexPos / mMutexFeatures). */
#ifndef MAPPOINT_H
#define MAPPOINT_H
#include <map>
#include <mutex>
#include <tuple>
#include "orbslam3_compat.h"
namespace ORB_SLAM3 {
class KeyFrame;
class Frame;
class Map;
class MapPoint {
 public:
  MapPoint(long unsigned int id, const Eigen::Vector3f& Pos, Map* pMap) : mnId(id), mWorldPos(Pos), mpMap(pMap) {}
  void SetWorldPos(const Eigen::Vector3f& Pos) { mWorldPos = Pos; }
  Eigen::Vector3f GetWorldPos() { return mWorldPos; }
  std::map<KeyFrame*, std::tuple<int, int>> GetObservations() { return mObservations; }
  int Observations() { return nObs; }
  void AddObservation(KeyFrame* pKF, int idx);
  void EraseObservation(KeyFrame* pKF);
  bool isBad() { return mbBad; }
  bool IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }   // src/MapPoint.cc:420-424
  std::tuple<int, int> GetIndexInKeyFrame(KeyFrame* pKF) {                      // src/MapPoint.cc:411-418
    const auto it = mObservations.find(pKF);
    return it != mObservations.end() ? it->second : std::tuple<int, int>(-1, -1);
  }
  void Replace(MapPoint* pMP);                                                  // src/MapPoint.cc:248-297
  MapPoint* GetReplaced() { return mpReplaced; }
  void IncreaseVisible(int n = 1) { mnVisible += n; }
  void IncreaseFound(int n = 1) { mnFound += n; }
  void ComputeDistinctiveDescriptors() { ++mnDescriptorUpdates; }
  Map* GetMap() { return mpMap; }
  void UpdateNormalAndDepth() { ++mnNormalUpdates; }
  cv::Mat GetDescriptor() { return mDescriptor.clone(); }
  // scale-invariance distances (src/MapPoint.cc:502-512) and predicted pyramid level (:531-546)
  float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
  float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
  int PredictScale(const float& currentDist, Frame* pF);
  int PredictScale(const float& currentDist, KeyFrame* pKF);   // src/MapPoint.cc:514-529
  Eigen::Vector3f GetNormal() { return mNormalVector; }

  static std::mutex mGlobalMutex;             // include/MapPoint.h:151 (held while PoseOptimization reads the positions)
  long unsigned int mnId;
  long unsigned int mnBALocalForKF = 0;
  long unsigned int mnBALocalForMerge = 0;    // include/MapPoint.h:139
  Eigen::Vector3f mPosGBA;                    // include/MapPoint.h:147-148
  long unsigned int mnBAGlobalForKF = 0;
  // tracking scratch written by Frame::isInFrustum (src/Frame.cc:513-587), read by SearchByProjection
  float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackProjXR = 0, mTrackProjYR = 0;
  bool mbTrackInView = false, mbTrackInViewR = false;
  int mnTrackScaleLevel = 0, mnTrackScaleLevelR = -1;
  float mTrackViewCos = 1.f, mTrackViewCosR = 1.f;   // include/MapPoint.h:122-129
  float mfMinDistance = 0, mfMaxDistance = 0;
  Eigen::Vector3f mNormalVector;
  // test-double state
  Eigen::Vector3f mWorldPos;
  std::map<KeyFrame*, std::tuple<int, int>> mObservations;
  int nObs = 0;
  bool mbBad = false;
  Map* mpMap;
  cv::Mat mDescriptor;
  int mnNormalUpdates = 0;
  MapPoint* mpReplaced = nullptr;
  int mnVisible = 1, mnFound = 1, mnDescriptorUpdates = 0;
};
}  // namespace ORB_SLAM3
#endif
