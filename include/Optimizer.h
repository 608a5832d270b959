/* Optimizer.h -- the entry points of ORB_SLAM3::Optimizer on the bundle-adjustment hot path, with the
 * reference's signatures (include/Optimizer.h:50-57,86).  Drop-in: same mangled symbols, same side effects. */
#ifndef OPTIMIZER_H
#define OPTIMIZER_H
#include "Frame.h"
#include "KeyFrame.h"
#include "LoopClosing.h"
#include "Map.h"
#include <vector>
#include "MapPoint.h"
namespace ORB_SLAM3 {
class Optimizer {
 public:
  // src/Optimizer.cc:61-392 / 53-58 (csrc/host/OptimizerGlobal.cc): every keyframe and point of the map, one optimize(nIterations)
  void static BundleAdjustment(const std::vector<KeyFrame*>& vpKF, const std::vector<MapPoint*>& vpMP, int nIterations = 5,
                               bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true);
  void static GlobalBundleAdjustemnt(Map* pMap, int nIterations = 5, bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                     const bool bRobust = true);
  // src/Optimizer.cc:815-1114 (csrc/host/OptimizerPose.cc): pose of a tracked frame from its map-point matches, four rounds of
  // optimize(10) with outlier re-classification; returns the number of inlier correspondences and sets mvbOutlier / the pose.
  int static PoseOptimization(Frame* pFrame);
  int static PoseInertialOptimizationLastKeyFrame(Frame* pFrame, bool bRecInit = false);   // include/Optimizer.h:60
  int static PoseInertialOptimizationLastFrame(Frame* pFrame, bool bRecInit = false);      // include/Optimizer.h:61
  // src/Optimizer.cc:1116-1498.  num_MPs is never assigned by the reference either.
  void static LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF,
                                    int& num_MPs, int& num_edges);
  // src/Optimizer.cc:3506-3955 (csrc/host/OptimizerGlobal.cc): the welding bundle adjustment of a map merge -- vpAdjustKF
  // optimised, vpFixedKF fixed, optimize(5) with Huber, outliers demoted and the kernel dropped, optimize(10)
  void static LocalBundleAdjustment(KeyFrame* pMainKF, std::vector<KeyFrame*> vpAdjustKF, std::vector<KeyFrame*> vpFixedKF, bool* pbStopFlag);
  // src/Optimizer.cc:2387-2964 (csrc/host/OptimizerInertial.cc); the num_* out-parameters are never assigned, as in the reference.
  void static LocalInertialBA(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs,
                              int& num_edges, bool bLarge = false, bool bRecInit = false);
  // src/Optimizer.cc:393-814 (csrc/host/OptimizerInertialMap.cc): visual-inertial BA of the whole map, one optimize(its) at lambda 1e-5.
  // vSingVal / bHess are unused by the reference too.  Maps of up to 1200 keyframes; bFixLocal: see INTEGRATION.md.
  void static FullInertialBA(Map* pMap, int its, const bool bFixLocal = false, const unsigned long nLoopKF = 0, bool* pbStopFlag = NULL,
                             bool bInit = false, float priorG = 1e2, float priorA = 1e6, Eigen::VectorXd* vSingVal = NULL, bool* bHess = NULL);
  // src/Optimizer.cc:3956-4498: the welding visual-inertial BA of a map merge (two temporal chains + up to 31 covisible keyframes)
  void static MergeInertialBA(KeyFrame* pCurrKF, KeyFrame* pMergeKF, bool* pbStopFlag, Map* pMap, LoopClosing::KeyFrameAndPose& corrPoses);
};
}  // namespace ORB_SLAM3
#endif
