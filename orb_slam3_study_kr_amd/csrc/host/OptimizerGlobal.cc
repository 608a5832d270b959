// OptimizerGlobal.cc -- ORB_SLAM3::Optimizer::BundleAdjustment / GlobalBundleAdjustemnt on MI355X.
//
// Same vertices, edge types and BlockSolver_6_3 as the local bundle adjustment; only the graph
// selection differs: every keyframe and map point handed in, the map's initial keyframe fixed, optional Huber kernel,
// ONE optimizer.optimize(nIterations) and no outlier pass (src/Optimizer.cc:61-392).  The device path is the one of
// Optimizer.cc; a map of more than ~240 keyframes has its reduced camera system factored in global memory (csrc/big_solve.h,
// up to 4000 keyframes: the system is held dense).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <map>
#include <mutex>
#include <set>

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
  struct Ed { KeyFrame* kf; MapPoint* mp; int pose; uint8_t kind; double obs[3]; double info; int right; };
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
        if (!stereo && !pk.mono_camera(pKF->mpCamera, pKF->fx, pKF->fy, pKF->cx, pKF->cy)) return;
        Ed e;
        e.kf = pKF; e.mp = pMP; e.pose = pit->second; e.kind = stereo ? OSH_EDGE_STEREO : OSH_EDGE_MONO;
        e.obs[0] = kpUn.pt.x; e.obs[1] = kpUn.pt.y; e.obs[2] = stereo ? kp_ur : -1.0;
        e.info = pKF->mvInvLevelSigma2[kpUn.octave];
        e.right = -1;
        edges.push_back(e);
      }
      // (the reference compares the UNSHIFTED index with mvKeysRight.size() here, :232: kept as it is)
      if (pKF->mpCamera2 && std::get<1>(ob.second) != -1 && std::get<1>(ob.second) < (int)pKF->mvKeysRight.size()) {
        // EdgeSE3ProjectXYZToBody (:235-283): right-camera observation; the (float) Huber delta is thHuber2D as for the left edge
        if (!pk.rig_camera(pKF)) return;
        if (std::get<1>(ob.second) < pKF->NLeft) { pk.unsupported = "right-camera index below NLeft"; return; }   // (read unchecked at :236)
        Ed e;
        e.kf = pKF; e.mp = pMP; e.pose = pit->second; e.kind = OSH_EDGE_BODY; e.right = std::get<1>(ob.second);
        const cv::KeyPoint& kp = pKF->mvKeysRight[e.right - pKF->NLeft];
        e.obs[0] = kp.pt.x; e.obs[1] = kp.pt.y; e.obs[2] = -1.0;
        e.info = pKF->mvInvLevelSigma2[kp.octave];
        edges.push_back(e);
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
  pk.camera_models_ok();
}

// Failure at the boundary must be SAFE for the reference caller: LoopClosing::RunGlobalBundleAdjustment
// (src/LoopClosing.cc:2330-2386) starts from the origin keyframes' mTcwGBA, propagates `Tchildc * pKF->mTcwGBA` down the
// spanning tree to every keyframe whose mnBAGlobalForKF != nLoopKF and then calls SetPose(mTcwGBA) / SetWorldPos(mPosGBA).
// When the device path cannot produce a result, the IDENTITY result is written (mTcwGBA = current pose, mPosGBA = current
// position, both stamped with nLoopKF), so that propagation is a no-op instead of reading stale or default members.  In the
// origin case (nLoopKF == origin keyframe) the reference writes poses directly and leaving them as they are is the no-op.
static void WriteIdentityGBA(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, Map* pMap,
                             const unsigned long nLoopKF) {
  if (nLoopKF == pMap->GetOriginKF()->mnId) return;
  for (KeyFrame* pKF : vpKFs) {
    if (!pKF || pKF->isBad()) continue;
    pKF->mTcwGBA = pKF->GetPose();
    pKF->mnBAGlobalForKF = nLoopKF;
  }
  for (MapPoint* pMP : vpMP) {
    if (!pMP || pMP->isBad()) continue;
    pMP->mPosGBA = pMP->GetWorldPos();
    pMP->mnBAGlobalForKF = nLoopKF;
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
    std::fprintf(stderr, "BA: %s is not supported by the MI355X path yet; identity result written\n", pk.unsupported);
    WriteIdentityGBA(vpKFs, vpMP, pMap, nLoopKF);
    return;
  }
  osh_lba_ctx* ctx = HostSolverContext();
  if (!ctx) { WriteIdentityGBA(vpKFs, vpMP, pMap, nLoopKF); return; }
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
    std::fprintf(stderr, "BA: device solve failed (%s); identity result written\n", osh_last_error());
    WriteIdentityGBA(vpKFs, vpMP, pMap, nLoopKF);
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

// ------------------------------------------------------------------------------------------------------------------
// Welding bundle adjustment of a map merge (src/Optimizer.cc:3506-3955, called from LoopClosing::MergeLocal).
void PackWeldingBA(KeyFrame* pMainKF, const std::vector<KeyFrame*>& vpAdjustKF, const std::vector<KeyFrame*>& vpFixedKF, LbaPack& pk,
                   std::vector<MapPoint*>& vpMPs) {
  pk = LbaPack();
  vpMPs.clear();
  Map* pCurrentMap = pMainKF->GetMap();
  long unsigned int maxKFid = 0;
  std::vector<KeyFrame*> vFree, vFixed;
  auto collect = [&](KeyFrame* pKFi) {   // the keyframe's map points, each once (:3551-3563, 3589-3606)
    const std::set<MapPoint*> spViewMPs = pKFi->GetMapPoints();
    for (MapPoint* pMPi : spViewMPs) {
      if (!pMPi) continue;
      if (!pMPi->isBad() && pMPi->GetMap() == pCurrentMap && pMPi->mnBALocalForMerge != pMainKF->mnId) {
        vpMPs.push_back(pMPi);
        pMPi->mnBALocalForMerge = pMainKF->mnId;
      }
    }
  };
  for (KeyFrame* pKFi : vpFixedKF) {
    if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;   // "ERROR LBA: KF is bad or is not in the current map"
    pKFi->mnBALocalForMerge = pMainKF->mnId;
    vFixed.push_back(pKFi);
    if (pKFi->mnId > maxKFid) maxKFid = pKFi->mnId;
    collect(pKFi);
  }
  for (KeyFrame* pKFi : vpAdjustKF) {
    if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;
    pKFi->mnBALocalForMerge = pMainKF->mnId;
    vFree.push_back(pKFi);
    if (pKFi->mnId > maxKFid) maxKFid = pKFi->mnId;
    collect(pKFi);
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
    const Eigen::Quaterniond q = Tcw.unit_quaternion().cast<double>();
    const Eigen::Vector3d t = Tcw.translation().cast<double>();
    const double qt[7] = {q.x(), q.y(), q.z(), q.w(), t[0], t[1], t[2]};
    pk.pose_qt.insert(pk.pose_qt.end(), qt, qt + 7);
    const double cam[5] = {pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mbf};
    pk.pose_cam.insert(pk.pose_cam.end(), cam, cam + 5);
  }
  for (MapPoint* pMPi : vpMPs) if (!pMPi->isBad()) pk.vPointMPs.push_back(pMPi);
  std::sort(pk.vPointMPs.begin(), pk.vPointMPs.end(), [](MapPoint* a, MapPoint* b) { return a->mnId < b->mnId; });
  std::map<MapPoint*, int> pointIndex;
  for (size_t j = 0; j < pk.vPointMPs.size(); ++j) {
    pointIndex[pk.vPointMPs[j]] = (int)j;
    const Eigen::Vector3d X = pk.vPointMPs[j]->GetWorldPos().cast<double>();
    pk.points.push_back(X[0]); pk.points.push_back(X[1]); pk.points.push_back(X[2]);
  }
  // edges (:3631-3705): left observation only, mono if mvuRight < 0, stereo otherwise; always a Huber kernel
  for (MapPoint* pMPi : vpMPs) {
    if (pMPi->isBad()) continue;
    const std::map<KeyFrame*, std::tuple<int, int>> observations = pMPi->GetObservations();
    for (const auto& ob : observations) {
      KeyFrame* pKF = ob.first;
      const int leftIndex = std::get<0>(ob.second);
      if (pKF->isBad() || pKF->mnId > maxKFid || pKF->mnBALocalForMerge != pMainKF->mnId || !pKF->GetMapPoint(leftIndex)) continue;
      const cv::KeyPoint& kpUn = pKF->mvKeysUn[leftIndex];
      const float kp_ur = pKF->mvuRight[leftIndex];
      const bool stereo = !(kp_ur < 0);
      if (!stereo && !pk.mono_camera(pKF->mpCamera, pKF->fx, pKF->fy, pKF->cx, pKF->cy)) return;
      pk.edge_pose.push_back(poseIndex.at(pKF));
      pk.edge_point.push_back(pointIndex.at(pMPi));
      pk.edge_kind.push_back(stereo ? OSH_EDGE_STEREO : OSH_EDGE_MONO);
      pk.edge_obs.push_back(kpUn.pt.x); pk.edge_obs.push_back(kpUn.pt.y); pk.edge_obs.push_back(stereo ? kp_ur : -1.0);
      pk.edge_info.push_back(pKF->mvInvLevelSigma2[kpUn.octave]);
      pk.vEdgeKF.push_back(pKF);
      pk.vEdgeMP.push_back(pMPi);
    }
  }
  pk.camera_models_ok();
}

void Optimizer::LocalBundleAdjustment(KeyFrame* pMainKF, std::vector<KeyFrame*> vpAdjustKF, std::vector<KeyFrame*> vpFixedKF, bool* pbStopFlag) {
  LbaPack pk;
  std::vector<MapPoint*> vpMPs;
  PackWeldingBA(pMainKF, vpAdjustKF, vpFixedKF, pk, vpMPs);
  if (pk.unsupported) {
    std::fprintf(stderr, "[BA]: %s is not supported by the MI355X path yet; map left untouched\n", pk.unsupported);
    return;
  }
  if (pbStopFlag && *pbStopFlag) return;   // :3707-3709
  osh_lba_ctx* ctx = HostSolverContext();
  if (!ctx) return;
  const size_t E = pk.edge_pose.size();
  const double thMono = 5.991, thStereo = 7.815;
  // ---- first optimisation: Huber kernels, 5 iterations (:3711-3712)
  osh_lba_problem prob;
  pk.fill(prob);
  prob.huber_mono = (double)(float)std::sqrt(5.99);     // const float thHuber2D = sqrt(5.99) (:3625)
  prob.huber_stereo = (double)(float)std::sqrt(7.815);
  prob.lambda_init = 0.0;
  prob.max_iterations = 5;
  prob.stop_flag = reinterpret_cast<const volatile unsigned char*>(pbStopFlag);
  std::vector<double> pose1((size_t)pk.n_free * 7), pts1(pk.points.size()), chi1(E);
  std::vector<uint8_t> depth1(E);
  osh_lba_result res;
  res.pose_qt = pose1.data(); res.points = pts1.data(); res.edge_chi2 = chi1.data(); res.edge_depth_pos = depth1.data();
  if (osh_lba_solve(ctx, 1, &prob, &res) != OSH_OK) {
    std::fprintf(stderr, "[BA]: device solve failed (%s); map left untouched\n", osh_last_error());
    return;
  }
  std::vector<double> chiF = chi1, poseF = pose1, ptsF = pts1;
  std::vector<uint8_t> depthF = depth1;
  const bool bDoMore = !(pbStopFlag && *pbStopFlag);   // :3714-3718
  if (bDoMore) {
    // outliers of the first optimisation leave the active set (setLevel(1)), every kernel is dropped (:3724-3753),
    // then optimize(10) from the current estimates (:3757-3758)
    std::vector<uint8_t> level1(E, 0);
    LbaPack p2;
    p2.n_free = pk.n_free; p2.n_fixed = pk.n_fixed;
    p2.vPointMPs = pk.vPointMPs;
    p2.pose_qt = pk.pose_qt;
    std::copy(pose1.begin(), pose1.end(), p2.pose_qt.begin());
    p2.pose_cam = pk.pose_cam;
    p2.points = pts1;
    std::vector<size_t> keep;
    for (size_t e = 0; e < E; ++e) {
      const double th = pk.edge_kind[e] == OSH_EDGE_MONO ? thMono : thStereo;
      if (!pk.vEdgeMP[e]->isBad() && (chi1[e] > th || !depth1[e])) { level1[e] = 1; continue; }
      keep.push_back(e);
      p2.edge_pose.push_back(pk.edge_pose[e]); p2.edge_point.push_back(pk.edge_point[e]); p2.edge_kind.push_back(pk.edge_kind[e]);
      for (int k = 0; k < 3; ++k) p2.edge_obs.push_back(pk.edge_obs[3 * e + k]);
      p2.edge_info.push_back(pk.edge_info[e]);
    }
    osh_lba_problem prob2;
    p2.fill(prob2);
    prob2.huber_mono = prob2.huber_stereo = std::numeric_limits<double>::infinity();   // e->setRobustKernel(0)
    prob2.lambda_init = 0.0;
    prob2.max_iterations = 10;
    prob2.stop_flag = reinterpret_cast<const volatile unsigned char*>(pbStopFlag);
    std::vector<double> pose2((size_t)pk.n_free * 7), pts2(pk.points.size()), chi2(keep.size());
    std::vector<uint8_t> depth2(keep.size());
    osh_lba_result res2;
    res2.pose_qt = pose2.data(); res2.points = pts2.data(); res2.edge_chi2 = chi2.data(); res2.edge_depth_pos = depth2.data();
    if (osh_lba_solve(ctx, 1, &prob2, &res2) != OSH_OK) {
      std::fprintf(stderr, "[BA]: device solve failed (%s); map left untouched\n", osh_last_error());
      return;
    }
    poseF = pose2; ptsF = pts2;
    for (size_t x = 0; x < keep.size(); ++x) { chiF[keep[x]] = chi2[x]; depthF[keep[x]] = depth2[x]; }
    // a demoted edge keeps the error of the first optimisation (it is never evaluated again), but isDepthPositive()
    // reads the final estimates (EdgeSE3ProjectXYZ::isDepthPositive, include/OptimizableTypes.h:99-103)
    for (size_t e = 0; e < E; ++e) {
      if (!level1[e]) continue;
      const int ip = pk.edge_pose[e], il = pk.edge_point[e];
      const double* qt = (ip < pk.n_free) ? &poseF[(size_t)ip * 7] : &pk.pose_qt[(size_t)ip * 7];
      const double* X = &ptsF[3 * (size_t)il];
      // third row of q * X + t with Eigen's two-cross-product form (se3quat.h:217-221)
      const double uv0 = 2 * (qt[1] * X[2] - qt[2] * X[1]), uv1 = 2 * (qt[2] * X[0] - qt[0] * X[2]), uv2 = 2 * (qt[0] * X[1] - qt[1] * X[0]);
      const double z = X[2] + qt[3] * uv2 + (qt[0] * uv1 - qt[1] * uv0) + qt[6];
      depthF[e] = z > 0.0 ? 1 : 0;
    }
  }
  // ---- outlier observations (:3761-3804): mono edges first, then stereo
  std::vector<std::pair<KeyFrame*, MapPoint*>> vToErase;
  for (int pass = 0; pass < 2; ++pass) {
    const int kind = pass == 0 ? OSH_EDGE_MONO : OSH_EDGE_STEREO;
    const double th = pass == 0 ? thMono : thStereo;
    for (size_t e = 0; e < E; ++e) {
      if (pk.edge_kind[e] != kind) continue;
      MapPoint* pMP = pk.vEdgeMP[e];
      if (pMP->isBad()) continue;
      if (chiF[e] > th || !depthF[e]) vToErase.push_back(std::make_pair(pk.vEdgeKF[e], pMP));
    }
  }
  std::unique_lock<std::mutex> lock(pMainKF->GetMap()->mMutexMapUpdate);   // :3809
  for (auto& er : vToErase) {
    er.first->EraseMapPointMatch(er.second);
    er.second->EraseObservation(er.first);
  }
  // ---- recover optimised data (:3843-3943)
  std::map<KeyFrame*, int> poseIndex;
  for (size_t i = 0; i < pk.vPoseKFs.size(); ++i) poseIndex[pk.vPoseKFs[i]] = (int)i;
  for (KeyFrame* pKFi : vpAdjustKF) {
    if (pKFi->isBad()) continue;
    const auto it = poseIndex.find(pKFi);
    if (it == poseIndex.end()) continue;   // not in the current map: never a vertex
    const double* qt = &poseF[(size_t)it->second * 7];
    pKFi->SetPose(Sophus::SE3f(Eigen::Quaterniond(qt[3], qt[0], qt[1], qt[2]).cast<float>(), Eigen::Vector3d(qt[4], qt[5], qt[6]).cast<float>()));
  }
  std::map<MapPoint*, int> pointIndex;
  for (size_t j = 0; j < pk.vPointMPs.size(); ++j) pointIndex[pk.vPointMPs[j]] = (int)j;
  for (MapPoint* pMPi : vpMPs) {
    if (pMPi->isBad()) continue;
    const auto it = pointIndex.find(pMPi);
    if (it == pointIndex.end()) continue;
    const size_t j = (size_t)it->second;
    pMPi->SetWorldPos(Eigen::Vector3d(ptsF[3 * j], ptsF[3 * j + 1], ptsF[3 * j + 2]).cast<float>());
    pMPi->UpdateNormalAndDepth();
  }
}

}  // namespace ORB_SLAM3
