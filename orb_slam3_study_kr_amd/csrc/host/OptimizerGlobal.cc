// OptimizerGlobal.cc -- ORB_SLAM3::Optimizer::BundleAdjustment / GlobalBundleAdjustemnt on MI355X.
//
// Same vertices, edge types and BlockSolver_6_3 as the local bundle adjustment; only the graph
// selection differs: every keyframe and map point handed in, the map's initial keyframe fixed, optional Huber kernel,
// ONE optimizer.optimize(nIterations) and no outlier pass (src/Optimizer.cc:61-392).  The device path is the one of
// Optimizer.cc; the reduced camera system of a window must fit the LDS-resident factorisation (about 230 keyframes).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <map>

#include "Optimizer.h"
#include "host_pack.h"
#include "orbslam3_hip.h"

namespace ORB_SLAM3 {

void PackBundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, LbaPack& pk,
                          std::vector<bool>& vbNotIncludedMP) {
  pk = LbaPack();
  vbNotIncludedMP.assign(vpMP.size(), false);
  Map* pMap = vpKFs[0]->GetMap();
  // pose vertices (:112-128): every keyframe that is not bad; fixed iff it is the map's initial keyframe.
  // Hessian order = ascending vertex id among the non-fixed vertices (g2o/core/sparse_optimizer.cpp:166-190)
  std::vector<KeyFrame*> vFree, vFixed;
  long unsigned int maxKFid = 0;
  for (KeyFrame* pKF : vpKFs) {
    if (pKF->isBad()) continue;
    (pKF->mnId == pMap->GetInitKFid() ? vFixed : vFree).push_back(pKF);
    if (pKF->mnId > maxKFid) maxKFid = pKF->mnId;
  }
  std::sort(vFree.begin(), vFree.end(), [](KeyFrame* a, KeyFrame* b) { return a->mnId < b->mnId; });
  pk.vPoseKFs = vFree;
  pk.vPoseKFs.insert(pk.vPoseKFs.end(), vFixed.begin(), vFixed.end());
  pk.n_free = (int)vFree.size();
  pk.n_fixed = (int)vFixed.size();
  pk.num_fixedKF = pk.n_fixed;
  std::map<KeyFrame*, int> poseIndex;
  for (size_t i = 0; i < pk.vPoseKFs.size(); ++i) poseIndex[pk.vPoseKFs[i]] = (int)i;
  for (KeyFrame* pKF : pk.vPoseKFs) {
    const Sophus::SE3f Tcw = pKF->GetPose();
    const Eigen::Quaterniond q = Tcw.unit_quaternion().cast<double>();   // :119-120 float -> double
    const Eigen::Vector3d t = Tcw.translation().cast<double>();
    const double qt[7] = {q.x(), q.y(), q.z(), q.w(), t[0], t[1], t[2]};
    pk.pose_qt.insert(pk.pose_qt.end(), qt, qt + 7);
    const double cam[5] = {pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mbf};   // :214-218
    pk.pose_cam.insert(pk.pose_cam.end(), cam, cam + 5);
  }
  // point vertices and edges (:134-300) in g2o insertion order: vpMP order x observation-map order.  A point without any
  // edge is removed again (:289-293); the others keep the Hessian order of their ids (ascending mnId).
  struct Ed { KeyFrame* kf; MapPoint* mp; int pose; uint8_t kind; double obs[3]; double info; };
  std::vector<Ed> edges;
  std::vector<MapPoint*> included;
  for (size_t i = 0; i < vpMP.size(); i++) {
    MapPoint* pMP = vpMP[i];
    if (pMP->isBad()) continue;
    const std::map<KeyFrame*, std::tuple<int, int>> observations = pMP->GetObservations();
    int nEdges = 0;
    for (const auto& ob : observations) {
      KeyFrame* pKF = ob.first;
      if (pKF->isBad() || pKF->mnId > maxKFid) continue;
      const auto pit = poseIndex.find(pKF);
      if (pit == poseIndex.end()) continue;   // optimizer.vertex(pKF->mnId) == NULL (:159)
      nEdges++;
      const int leftIndex = std::get<0>(ob.second);
      if (leftIndex != -1) {
        const cv::KeyPoint& kpUn = pKF->mvKeysUn[leftIndex];
        const float kp_ur = pKF->mvuRight[leftIndex];
        const bool stereo = kp_ur >= 0;   // mono if mvuRight < 0 (:167), stereo otherwise (:194)
        if (!stereo) {
          GeometricCamera* cam = pKF->mpCamera;
          if (!cam || cam->GetType() != GeometricCamera::CAM_PINHOLE || cam->getParameter(0) != pKF->fx ||
              cam->getParameter(1) != pKF->fy || cam->getParameter(2) != pKF->cx || cam->getParameter(3) != pKF->cy) {
            pk.unsupported = "monocular observation through a camera that is not the keyframe's pinhole model";
            return;
          }
        }
        Ed e;
        e.kf = pKF; e.mp = pMP; e.pose = pit->second; e.kind = stereo ? OSH_EDGE_STEREO : OSH_EDGE_MONO;
        e.obs[0] = kpUn.pt.x; e.obs[1] = kpUn.pt.y; e.obs[2] = stereo ? kp_ur : -1.0;
        e.info = pKF->mvInvLevelSigma2[kpUn.octave];
        edges.push_back(e);
      }
      if (pKF->mpCamera2 && std::get<1>(ob.second) != -1) {
        pk.unsupported = "right-camera (fisheye stereo) observation";   // EdgeSE3ProjectXYZToBody (:235-283)
        return;
      }
    }
    if (nEdges == 0) vbNotIncludedMP[i] = true;
    else included.push_back(pMP);
  }
  pk.vPointMPs = included;
  std::sort(pk.vPointMPs.begin(), pk.vPointMPs.end(), [](MapPoint* a, MapPoint* b) { return a->mnId < b->mnId; });
  std::map<MapPoint*, int> pointIndex;
  for (size_t j = 0; j < pk.vPointMPs.size(); ++j) {
    pointIndex[pk.vPointMPs[j]] = (int)j;
    const Eigen::Vector3d X = pk.vPointMPs[j]->GetWorldPos().cast<double>();   // :141
    pk.points.push_back(X[0]); pk.points.push_back(X[1]); pk.points.push_back(X[2]);
  }
  for (const Ed& e : edges) {
    pk.edge_pose.push_back(e.pose);
    pk.edge_point.push_back(pointIndex.at(e.mp));
    pk.edge_kind.push_back(e.kind);
    pk.edge_obs.push_back(e.obs[0]); pk.edge_obs.push_back(e.obs[1]); pk.edge_obs.push_back(e.obs[2]);
    pk.edge_info.push_back(e.info);
    pk.vEdgeKF.push_back(e.kf);
    pk.vEdgeMP.push_back(e.mp);
  }
}

void Optimizer::GlobalBundleAdjustemnt(Map* pMap, int nIterations, bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust) {
  std::vector<KeyFrame*> vpKFs = pMap->GetAllKeyFrames();
  std::vector<MapPoint*> vpMP = pMap->GetAllMapPoints();
  BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust);
}

void Optimizer::BundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, int nIterations,
                                 bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust) {
  std::vector<bool> vbNotIncludedMP;
  LbaPack pk;
  PackBundleAdjustment(vpKFs, vpMP, pk, vbNotIncludedMP);
  Map* pMap = vpKFs[0]->GetMap();
  if (pk.unsupported) {
    std::fprintf(stderr, "BA: %s is not supported by the MI355X path yet; map left untouched\n", pk.unsupported);
    return;
  }
  osh_lba_ctx* ctx = HostSolverContext();
  if (!ctx) return;
  osh_lba_problem prob;
  pk.fill(prob);
  // const float thHuber2D = sqrt(5.99); const float thHuber3D = sqrt(7.815) (:130-131); no kernel at all unless bRobust
  prob.huber_mono = bRobust ? (double)(float)std::sqrt(5.99) : std::numeric_limits<double>::infinity();
  prob.huber_stereo = bRobust ? (double)(float)std::sqrt(7.815) : std::numeric_limits<double>::infinity();
  prob.lambda_init = 0.0;
  prob.max_iterations = nIterations;                       // optimizer.optimize(nIterations) (:299)
  prob.stop_flag = reinterpret_cast<const volatile unsigned char*>(pbStopFlag);   // setForceStopFlag (:80-81)
  std::vector<double> out_pose((size_t)pk.n_free * 7), out_pts(pk.points.size());
  osh_lba_result res;
  res.pose_qt = out_pose.data(); res.points = out_pts.data(); res.edge_chi2 = nullptr; res.edge_depth_pos = nullptr;
  if (osh_lba_solve(ctx, 1, &prob, &res) != OSH_OK) {
    std::fprintf(stderr, "BA: device solve failed (%s); map left untouched\n", osh_last_error());
    return;
  }
  // keyframes (:303-379).  The statistics block for keyframes that moved by more than 1 m (:323-377) has no side effect.
  for (size_t i = 0; i < pk.vPoseKFs.size(); ++i) {
    KeyFrame* pKF = pk.vPoseKFs[i];
    const double* qt = ((int)i < pk.n_free) ? &out_pose[i * 7] : &pk.pose_qt[i * 7];
    const Sophus::SE3f T(Eigen::Quaterniond(qt[3], qt[0], qt[1], qt[2]).cast<float>(), Eigen::Vector3d(qt[4], qt[5], qt[6]).cast<float>());
    if (nLoopKF == pMap->GetOriginKF()->mnId) {
      pKF->SetPose(T);
    } else {
      pKF->mTcwGBA = T;
      pKF->mnBAGlobalForKF = nLoopKF;
    }
  }
  // points (:381-391)
  for (size_t j = 0; j < pk.vPointMPs.size(); ++j) {
    MapPoint* pMP = pk.vPointMPs[j];
    const Eigen::Vector3f X = Eigen::Vector3d(out_pts[3 * j], out_pts[3 * j + 1], out_pts[3 * j + 2]).cast<float>();
    if (nLoopKF == pMap->GetOriginKF()->mnId) {
      pMP->SetWorldPos(X);
      pMP->UpdateNormalAndDepth();
    } else {
      pMP->mPosGBA = X;
      pMP->mnBAGlobalForKF = nLoopKF;
    }
  }
}

}  // namespace ORB_SLAM3
