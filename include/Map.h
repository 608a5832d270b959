/* Map.h -- members of ORB_SLAM3::Map used by Optimizer::LocalBundleAdjustment
 * (reference include/Map.h:141,155-156; src/Map.cc:181,291,341). */
#ifndef MAP_H
#define MAP_H
#include <mutex>
#include <set>
namespace ORB_SLAM3 {
class Map {
 public:
  long unsigned int GetInitKFid() { return mnInitKFid; }
  bool IsInertial() { return mbIsInertial; }
  long unsigned int KeyFramesInMap() { return mnKeyFrames; }   // src/Map.cc:165
  void IncreaseChangeIndex() { ++mnMapChange; }
  int GetMapChangeIndex() { return mnMapChange; }
  std::mutex mMutexMapUpdate;
  std::set<long unsigned int> msOptKFs;
  std::set<long unsigned int> msFixedKFs;
  // test-double state (the real class keeps these private)
  long unsigned int mnInitKFid = 0;
  bool mbIsInertial = false;
  int mnMapChange = 0;
  long unsigned int mnKeyFrames = 0;
};
}  // namespace ORB_SLAM3
#endif
