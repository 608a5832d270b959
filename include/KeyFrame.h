/* KeyFrame.h -- members of ORB_SLAM3::KeyFrame used by Optimizer::LocalBundleAdjustment
 * (reference include/KeyFrame.h; src/KeyFrame.cc:109-131,237,309,367,681,1108).  Minimal test double. */
#ifndef KEYFRAME_H
#define KEYFRAME_H
#include <vector>
#include "CameraModels/GeometricCamera.h"
#include "Map.h"
#include "MapPoint.h"
#include "orbslam3_compat.h"
namespace ORB_SLAM3 {
class KeyFrame {
 public:
  KeyFrame(long unsigned int id, Map* pMap) : mnId(id), mpMap(pMap) {}
  void SetPose(const Sophus::SE3f& Tcw) { mTcw = Tcw; ++mnPoseSets; }
  Sophus::SE3f GetPose() { return mTcw; }
  std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { return mvpOrderedConnectedKeyFrames; }
  std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
  void EraseMapPointMatch(MapPoint* pMP);
  Sophus::SE3f GetRelativePoseTrl() { return mTrl; }
  bool isBad() { return mbBad; }
  Map* GetMap() { return mpMap; }

  long unsigned int mnId;
  long unsigned int mnBALocalForKF = 0, mnBAFixedForKF = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
  int N = 0, NLeft = -1;
  std::vector<cv::KeyPoint> mvKeysUn, mvKeysRight;
  std::vector<float> mvuRight;
  std::vector<float> mvInvLevelSigma2;
  GeometricCamera* mpCamera = nullptr;
  GeometricCamera* mpCamera2 = nullptr;
  // test-double state
  Sophus::SE3f mTcw, mTrl;
  std::vector<KeyFrame*> mvpOrderedConnectedKeyFrames;
  std::vector<MapPoint*> mvpMapPoints;
  bool mbBad = false;
  Map* mpMap;
  int mnPoseSets = 0;
};
}  // namespace ORB_SLAM3
#endif
