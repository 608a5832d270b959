// Optimizer.cc -- ORB_SLAM3::Optimizer::LocalBundleAdjustment on MI355X.
//
// Host side of the drop-in: walks the KeyFrame/MapPoint graph exactly like the reference
// (src/Optimizer.cc:1116-1498), packs it into the flat arrays of osh_lba_problem, lets the HIP
// path run optimizer.optimize(10) (src/Optimizer.cc:1410-1411) and applies the reference's outlier
// test, observation erasure and write-back (:1413-1497).  No g2o, no Eigen.
#include "Optimizer.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <mutex>
#include <set>

#include "host_pack.h"
#include "orbslam3_hip.h"

namespace ORB_SLAM3 {

namespace {

// One solver context (HIP stream + reusable device buffers) per calling thread, destroyed when the thread exits: the reference
// runs every GlobalBundleAdjustment in a freshly created std::thread (LoopClosing), so a context that outlived its thread would
// leak a stream, pinned memory and the largest device buffers of the process per loop closure.
struct ThreadCtx {
  osh_lba_ctx* ctx = nullptr;
  ~ThreadCtx() { if (ctx) osh_lba_destroy(ctx); }
};

osh_lba_ctx* thread_ctx() {
  static thread_local ThreadCtx holder;
  if (!holder.ctx) {
    const char* dev = std::getenv("ORBSLAM3_HIP_DEVICE");
    if (osh_lba_create(dev ? std::atoi(dev) : 0, &holder.ctx) != OSH_OK) {
      std::fprintf(stderr, "LM-LBA: cannot create the HIP solver context: %s\n", osh_last_error());
      holder.ctx = nullptr;
    }
  }
  return holder.ctx;
}

}  // namespace

osh_lba_ctx* HostSolverContext() { return thread_ctx(); }

// Steps 1-6 of src/Optimizer.cc:1118-1404: select the window and flatten it.
bool PackLocalBA(KeyFrame* pKF, Map* pMap, LbaPack& pk) {
  pk = LbaPack();
  // 1. local keyframes: pKF + its covisibles (:1118-1132)
  pk.lLocalKeyFrames.push_back(pKF);
  pKF->mnBALocalForKF = pKF->mnId;
  Map* pCurrentMap = pKF->GetMap();
  const std::vector<KeyFrame*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
  for (KeyFrame* pKFi : vNeighKFs) {
    pKFi->mnBALocalForKF = pKF->mnId;
    if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) pk.lLocalKeyFrames.push_back(pKFi);
  }
  // 2. local map points (:1134-1160)
  pk.num_fixedKF = 0;
  for (KeyFrame* pKFi : pk.lLocalKeyFrames) {
    if (pKFi->mnId == pMap->GetInitKFid()) pk.num_fixedKF = 1;
    const std::vector<MapPoint*> vpMPs = pKFi->GetMapPointMatches();
    for (MapPoint* pMP : vpMPs) {
      if (!pMP) continue;
      if (pMP->isBad() || pMP->GetMap() != pCurrentMap) continue;
      if (pMP->mnBALocalForKF != pKF->mnId) {
        pk.lLocalMapPoints.push_back(pMP);
        pMP->mnBALocalForKF = pKF->mnId;
      }
    }
  }
  // 3. fixed keyframes: observers of local points that are not local (:1162-1179)
  for (MapPoint* pMP : pk.lLocalMapPoints) {
    const std::map<KeyFrame*, std::tuple<int, int>> observations = pMP->GetObservations();
    for (const auto& ob : observations) {
      KeyFrame* pKFi = ob.first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
        pKFi->mnBAFixedForKF = pKF->mnId;
        if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) pk.lFixedCameras.push_back(pKFi);
      }
    }
  }
  pk.num_fixedKF += (int)pk.lFixedCameras.size();
  if (pk.num_fixedKF == 0) return false;  // caller prints the reference's message and returns (:1182-1186)

  // 4./5. vertices.  Hessian order = ascending vertex id among the non-fixed poses, then the points
  // (g2o/core/sparse_optimizer.cpp:166-190); the map's initial keyframe is a fixed vertex (:1220).
  std::vector<KeyFrame*> vFree, vFixed;
  for (KeyFrame* pKFi : pk.lLocalKeyFrames) (pKFi->mnId == pMap->GetInitKFid() ? vFixed : vFree).push_back(pKFi);
  std::sort(vFree.begin(), vFree.end(), [](KeyFrame* a, KeyFrame* b) { return a->mnId < b->mnId; });
  for (KeyFrame* pKFi : pk.lFixedCameras) vFixed.push_back(pKFi);
  pk.vPoseKFs = vFree;
  pk.vPoseKFs.insert(pk.vPoseKFs.end(), vFixed.begin(), vFixed.end());
  pk.n_free = (int)vFree.size();
  pk.n_fixed = (int)vFixed.size();
  std::map<KeyFrame*, int> poseIndex;
  for (size_t i = 0; i < pk.vPoseKFs.size(); ++i) poseIndex[pk.vPoseKFs[i]] = (int)i;
  for (KeyFrame* pKFi : pk.vPoseKFs) {
    const Sophus::SE3f Tcw = pKFi->GetPose();
    const Eigen::Quaterniond q = Tcw.unit_quaternion().cast<double>();   // :1217-1218 float -> double
    const Eigen::Vector3d t = Tcw.translation().cast<double>();
    const double qt[7] = {q.x(), q.y(), q.z(), q.w(), t[0], t[1], t[2]};
    pk.pose_qt.insert(pk.pose_qt.end(), qt, qt + 7);
    const double cam[5] = {pKFi->fx, pKFi->fy, pKFi->cx, pKFi->cy, pKFi->mbf};  // :1352-1356
    pk.pose_cam.insert(pk.pose_cam.end(), cam, cam + 5);
  }
  pk.vPointMPs.assign(pk.lLocalMapPoints.begin(), pk.lLocalMapPoints.end());
  std::sort(pk.vPointMPs.begin(), pk.vPointMPs.end(), [](MapPoint* a, MapPoint* b) { return a->mnId < b->mnId; });
  std::map<MapPoint*, int> pointIndex;
  for (size_t j = 0; j < pk.vPointMPs.size(); ++j) {
    pointIndex[pk.vPointMPs[j]] = (int)j;
    const Eigen::Vector3d X = pk.vPointMPs[j]->GetWorldPos().cast<double>();  // :1286
    pk.points.push_back(X[0]); pk.points.push_back(X[1]); pk.points.push_back(X[2]);
  }
  // 6. edges in g2o insertion order: landmark list order x observation-map order (:1293-1402)
  for (MapPoint* pMP : pk.lLocalMapPoints) {
    const std::map<KeyFrame*, std::tuple<int, int>> observations = pMP->GetObservations();
    for (const auto& ob : observations) {
      KeyFrame* pKFi = ob.first;
      if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;
      const int leftIndex = std::get<0>(ob.second);
      if (leftIndex != -1) {
        const cv::KeyPoint& kpUn = pKFi->mvKeysUn[leftIndex];
        const float kp_ur = pKFi->mvuRight[leftIndex];
        const bool stereo = kp_ur >= 0;   // mono if mvuRight < 0 (:1305), stereo otherwise (:1332)
        if (!stereo) {
          // the mono edge projects through pKFi->mpCamera (:1323); the device keeps ONE intrinsics row per keyframe
          if (!pk.mono_camera(pKFi->mpCamera, pKFi->fx, pKFi->fy, pKFi->cx, pKFi->cy)) return true;
        }
        pk.edge_pose.push_back(poseIndex.at(pKFi));
        pk.edge_point.push_back(pointIndex.at(pMP));
        pk.edge_kind.push_back(stereo ? OSH_EDGE_STEREO : OSH_EDGE_MONO);
        pk.edge_obs.push_back(kpUn.pt.x); pk.edge_obs.push_back(kpUn.pt.y); pk.edge_obs.push_back(stereo ? kp_ur : -1.0);
        const float& invSigma2 = pKFi->mvInvLevelSigma2[kpUn.octave];
        pk.edge_info.push_back(invSigma2);
        pk.vEdgeKF.push_back(pKFi);
        pk.vEdgeMP.push_back(pMP);
      }
      if (pKFi->mpCamera2 && std::get<1>(ob.second) != -1) {
        // EdgeSE3ProjectXYZToBody (:1365-1399): the right-camera observation, a second edge on the pair's Hessian block
        if (!pk.add_body_edge(pKFi, pMP, std::get<1>(ob.second), poseIndex.at(pKFi), pointIndex.at(pMP))) return true;
      }
    }
  }
  pk.camera_models_ok();
  return true;
}

void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF,
                                      int& num_MPs, int& num_edges) {
  (void)num_MPs;  // never assigned by the reference
  LbaPack pk;
  const bool haveFixed = PackLocalBA(pKF, pMap, pk);
  num_fixedKF = pk.num_fixedKF;
  if (!haveFixed) {
    std::fprintf(stderr, "LM-LBA: There are 0 fixed KF in the optimizations, LBA aborted\n");
    return;
  }
  Map* pCurrentMap = pKF->GetMap();
  // DEBUG LBA sets (:1209-1210,1225,1242)
  pCurrentMap->msOptKFs.clear();
  pCurrentMap->msFixedKFs.clear();
  for (KeyFrame* pKFi : pk.lLocalKeyFrames) pCurrentMap->msOptKFs.insert(pKFi->mnId);
  for (KeyFrame* pKFi : pk.lFixedCameras) pCurrentMap->msFixedKFs.insert(pKFi->mnId);
  num_OptKF = (int)pk.lLocalKeyFrames.size();
  if (pk.unsupported) {
    std::fprintf(stderr, "LM-LBA: %s is not supported by the MI355X path yet; map left untouched\n", pk.unsupported);
    return;
  }
  num_edges = (int)pk.edge_pose.size();
  if (pbStopFlag && *pbStopFlag) return;  // :1406-1408

  osh_lba_ctx* ctx = thread_ctx();
  if (!ctx) return;  // device error: map stays consistent (SURVEY.md 8b "Errors")
  osh_lba_problem prob;
  pk.fill(prob);
  prob.huber_mono = (double)(float)std::sqrt(5.991);    // const float thHuberMono = sqrt(5.991)  (:1275)
  prob.huber_stereo = (double)(float)std::sqrt(7.815);  // (:1276)
  prob.lambda_init = pMap->IsInertial() ? 100.0 : 0.0;  // solver->setUserLambdaInit(100.0) (:1197-1198)
  prob.max_iterations = 10;                             // optimizer.optimize(10) (:1411)
  prob.stop_flag = reinterpret_cast<const volatile unsigned char*>(pbStopFlag);
  std::vector<double> out_pose((size_t)pk.n_free * 7), out_pts(pk.points.size()), out_chi2(pk.edge_pose.size());
  std::vector<uint8_t> out_depth(pk.edge_pose.size());
  osh_lba_result res;
  res.pose_qt = out_pose.data(); res.points = out_pts.data(); res.edge_chi2 = out_chi2.data(); res.edge_depth_pos = out_depth.data();
  if (osh_lba_solve(ctx, 1, &prob, &res) != OSH_OK) {
    std::fprintf(stderr, "LM-LBA: device solve failed (%s); map left untouched\n", osh_last_error());
    return;
  }

  // 9. outlier observations (:1413-1460): mono edges first, then the right-camera (body) edges, then stereo, each in insertion order
  std::vector<std::pair<KeyFrame*, MapPoint*>> vToErase;
  vToErase.reserve(pk.edge_pose.size());
  for (int pass = 0; pass < 3; ++pass) {
    const int kind = pass == 0 ? OSH_EDGE_MONO : (pass == 1 ? OSH_EDGE_BODY : OSH_EDGE_STEREO);
    const double th = pass == 2 ? 7.815 : 5.991;
    for (size_t e = 0; e < pk.edge_kind.size(); ++e) {
      if (pk.edge_kind[e] != kind) continue;
      MapPoint* pMP = pk.vEdgeMP[e];
      if (pMP->isBad()) continue;
      if (out_chi2[e] > th || !out_depth[e]) vToErase.push_back(std::make_pair(pk.vEdgeKF[e], pMP));
    }
  }
  // 10. commit under the map mutex (:1464-1475)
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
  for (auto& er : vToErase) {
    er.first->EraseMapPointMatch(er.second);
    er.second->EraseObservation(er.first);
  }
  // 11. recover optimised data (:1477-1497); poses of the map's initial keyframe were fixed and stay as they are
  for (KeyFrame* pKFi : pk.lLocalKeyFrames) {
    int idx = -1;
    for (int i = 0; i < (int)pk.vPoseKFs.size(); ++i) if (pk.vPoseKFs[i] == pKFi) { idx = i; break; }
    const double* qt = (idx < pk.n_free) ? &out_pose[(size_t)idx * 7] : &pk.pose_qt[(size_t)idx * 7];
    Sophus::SE3f Tiw(Eigen::Quaterniond(qt[3], qt[0], qt[1], qt[2]).cast<float>(), Eigen::Vector3d(qt[4], qt[5], qt[6]).cast<float>());
    pKFi->SetPose(Tiw);
  }
  for (MapPoint* pMP : pk.lLocalMapPoints) {
    int j = -1;
    {
      auto it = std::lower_bound(pk.vPointMPs.begin(), pk.vPointMPs.end(), pMP, [](MapPoint* a, MapPoint* b) { return a->mnId < b->mnId; });
      j = (int)(it - pk.vPointMPs.begin());
    }
    pMP->SetWorldPos(Eigen::Vector3d(out_pts[3 * (size_t)j], out_pts[3 * (size_t)j + 1], out_pts[3 * (size_t)j + 2]).cast<float>());
    pMP->UpdateNormalAndDepth();
  }
  pMap->IncreaseChangeIndex();
}

}  // namespace ORB_SLAM3
