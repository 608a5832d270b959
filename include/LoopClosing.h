/* LoopClosing.h -- the one type of ORB_SLAM3::LoopClosing the hot path names: the keyframe -> corrected Sim3 map that
 * Optimizer::MergeInertialBA fills (reference include/LoopClosing.h:49-51). */
#ifndef LOOPCLOSING_H
#define LOOPCLOSING_H
#include <functional>
#include <map>
#include <set>
#include <utility>
#include "orbslam3_compat.h"
namespace ORB_SLAM3 {
class KeyFrame;
class LoopClosing {
 public:
  typedef std::pair<std::set<KeyFrame*>, int> ConsistentGroup;
  typedef std::map<KeyFrame*, g2o::Sim3, std::less<KeyFrame*>, Eigen::aligned_allocator<std::pair<KeyFrame* const, g2o::Sim3>>> KeyFrameAndPose;
};
}  // namespace ORB_SLAM3
#endif
