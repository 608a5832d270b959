// host_pack.h -- the flattened local-BA window produced by PackLocalBA (csrc/host/Optimizer.cc).
#pragma once
#include <cstdint>
#include <list>
#include <vector>
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"
#include "orbslam3_hip.h"

namespace ORB_SLAM3 {
struct LbaPack {
  std::list<KeyFrame*> lLocalKeyFrames, lFixedCameras;   // same containers / order as src/Optimizer.cc:1119,1163
  std::list<MapPoint*> lLocalMapPoints;                  // :1136
  std::vector<KeyFrame*> vPoseKFs;                       // pose order of the problem: optimisable (ascending id), then fixed
  std::vector<MapPoint*> vPointMPs;                      // ascending id
  std::vector<KeyFrame*> vEdgeKF;                        // per edge, insertion order
  std::vector<MapPoint*> vEdgeMP;
  int n_free = 0, n_fixed = 0, num_fixedKF = 0;
  const char* unsupported = nullptr;
  std::vector<double> pose_qt, pose_cam, points, edge_obs, edge_info;
  std::vector<int32_t> edge_pose, edge_point;
  std::vector<uint8_t> edge_kind;
  void fill(osh_lba_problem& p) const {
    p.n_free = n_free; p.n_fixed = n_fixed; p.n_points = (int32_t)vPointMPs.size(); p.n_edges = (int32_t)edge_pose.size();
    p.pose_qt = pose_qt.data(); p.pose_cam = pose_cam.data(); p.points = points.data();
    p.edge_pose = edge_pose.data(); p.edge_point = edge_point.data(); p.edge_kind = edge_kind.data();
    p.edge_obs = edge_obs.data(); p.edge_info = edge_info.data();
    p.huber_mono = p.huber_stereo = 0; p.lambda_init = 0; p.max_iterations = 10; p.stop_flag = nullptr;
  }
};
// Steps 1-6 of Optimizer::LocalBundleAdjustment; false when the window has no fixed keyframe.
bool PackLocalBA(KeyFrame* pKF, Map* pMap, LbaPack& pk);
}  // namespace ORB_SLAM3
