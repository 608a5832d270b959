/* Map.h -- members of ORB_SLAM3::Map used by Optimizer::LocalBundleAdjustment
 * (reference include/Map.h:84-85,97,141,155-156; src/Map.cc:153-189,291,341). */
#ifndef MAP_H
#define MAP_H
#include <mutex>
#include <set>
#include <vector>
namespace ORB_SLAM3 {
class KeyFrame;
class MapPoint;
class Map {
 public:
  long unsigned int GetInitKFid() { return mnInitKFid; }
  bool IsInertial() { return mbIsInertial; }
  long unsigned int KeyFramesInMap() { return mnKeyFrames; }   // src/Map.cc:165
  std::vector<KeyFrame*> GetAllKeyFrames() { return mvpKeyFrames; }   // src/Map.cc:153-163 (copies of the sets)
  std::vector<MapPoint*> GetAllMapPoints() { return mvpMapPoints; }
  long unsigned int GetMaxKFid() { return mnMaxKFid; }               // src/Map.cc:177-181
  KeyFrame* GetOriginKF() { return mpKFinitial; }                     // src/Map.cc:186-189
  void IncreaseChangeIndex() { ++mnMapChange; }
  int GetMapChangeIndex() { return mnMapChange; }
  std::mutex mMutexMapUpdate;
  std::set<long unsigned int> msOptKFs;
  std::set<long unsigned int> msFixedKFs;
  // test-double state (the real class keeps these private)
  long unsigned int mnInitKFid = 0;
  long unsigned int mnMaxKFid = 0;
  bool mbIsInertial = false;
  int mnMapChange = 0;
  long unsigned int mnKeyFrames = 0;
  std::vector<KeyFrame*> mvpKeyFrames;
  std::vector<MapPoint*> mvpMapPoints;
  KeyFrame* mpKFinitial = nullptr;
};
}  // namespace ORB_SLAM3
#endif
