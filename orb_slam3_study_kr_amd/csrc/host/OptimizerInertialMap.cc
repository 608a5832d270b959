// OptimizerInertialMap.cc -- ORB_SLAM3::Optimizer::FullInertialBA and MergeInertialBA on MI355X (host side).
//
// Both are the LocalInertialBA problem with another choice of keyframes (src/Optimizer.cc:393-814, 3956-4498): every vertex they
// create is a VertexPose with or without (velocity, gyro bias, acc bias) vertices, every edge an EdgeMono / EdgeStereo / EdgeInertial /
// EdgeGyroRW / EdgeAccRW, the solver a BlockSolverX + Levenberg with a user lambda and ONE optimize(its).  They run in the same
// one-launch kernel (osh_liba_solve).  A keyframe whose velocity / bias vertices have no active edge (g2o leaves such vertices out of
// the active set, Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:218-243: a !bImu keyframe, a covisible keyframe of the merge) is packed
// as a 15-dof keyframe without inertial links: its last nine columns carry the damping term only, their increment is exactly zero
// and the values it is written back with are the ones it came with.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <list>
#include <map>
#include <mutex>
#include <set>

#include "Optimizer.h"
#include "host_pack.h"
#include "orbslam3_hip.h"

namespace ORB_SLAM3 {
bool InvertDense(int n, const double* A, double* inv);   // OptimizerInertial.cc

namespace {
// ImuCamPose(KeyFrame*) (src/G2oTypes.cc:25-71) of every keyframe of pk.vPoseKFs, calibration of the first one
void PackKeyframeStates(LibaPack& pk) {
  for (size_t i = 0; i < pk.vPoseKFs.size(); ++i) {
    KeyFrame* k = pk.vPoseKFs[i];
    const Eigen::Matrix3f Rcw = k->GetRotation(), Rwb = k->GetImuRotation();
    const Eigen::Vector3f tcw = k->GetTranslation(), twb = k->GetImuPosition();
    for (int a = 0; a < 9; ++a) { pk.pose_Rcw.push_back((double)Rcw.v[a]); pk.pose_Rwb.push_back((double)Rwb.v[a]); }
    for (int a = 0; a < 3; ++a) { pk.pose_tcw.push_back((double)tcw(a)); pk.pose_twb.push_back((double)twb(a)); }
    if ((int)i < pk.n_opt + pk.n_fixed_imu) {
      const Eigen::Vector3f v = k->GetVelocity(), bg = k->GetGyroBias(), ba = k->GetAccBias();
      for (int a = 0; a < 3; ++a) { pk.vel.push_back((double)v(a)); pk.bias_g.push_back((double)bg(a)); pk.bias_a.push_back((double)ba(a)); }
    }
  }
  KeyFrame* pKF = pk.vPoseKFs.front();
  const IMU::Calib& cal = pKF->mImuCalib;
  const Eigen::Matrix3f Rcb = cal.mTcb.rotationMatrix();
  for (int a = 0; a < 9; ++a) pk.Rcb[a] = (double)Rcb.v[a];
  for (int a = 0; a < 3; ++a) { pk.tcb[a] = (double)cal.mTcb.translation()(a); pk.tbc[a] = (double)cal.mTbc.translation()(a); }
  pk.cam[0] = pKF->fx; pk.cam[1] = pKF->fy; pk.cam[2] = pKF->cx; pk.cam[3] = pKF->cy; pk.cam[4] = pKF->mbf;
}

// EdgeInertial + EdgeGyroRW + EdgeAccRW between pKFi->mPrevKF and pKFi (src/Optimizer.cc:523-568, 4226-4257): Huber on the inertial
// edge, plain information (no down-weighting of the oldest link here)
void PackLink(LibaPack& pk, KeyFrame* pKFi, int prev, int cur) {
  IMU::Preintegrated* P = pKFi->mpImuPreintegrated;
  pk.link_prev.push_back(prev);
  pk.link_cur.push_back(cur);
  float rec[OSH_PREINT_FLOATS];
  std::memset(rec, 0, sizeof(rec));
  rec[0] = P->dT;
  for (int a = 0; a < 9; ++a) { rec[1 + a] = P->dR.v[a]; rec[16 + a] = P->JRg.v[a]; rec[25 + a] = P->JVg.v[a]; rec[34 + a] = P->JVa.v[a]; rec[43 + a] = P->JPg.v[a]; rec[52 + a] = P->JPa.v[a]; }
  for (int a = 0; a < 3; ++a) { rec[10 + a] = P->dV(a); rec[13 + a] = P->dP(a); }
  rec[61] = P->b.bax; rec[62] = P->b.bay; rec[63] = P->b.baz; rec[64] = P->b.bwx; rec[65] = P->b.bwy; rec[66] = P->b.bwz;
  pk.link_preint.insert(pk.link_preint.end(), rec, rec + OSH_PREINT_FLOATS);
  double info[81];
  InertialInformation(P->C, info);
  pk.link_info.insert(pk.link_info.end(), info, info + 81);
  pk.link_robust.push_back(1);
  for (int which = 0; which < 2; ++which) {
    double Cb[9], inv[9];
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) Cb[a * 3 + c] = (double)P->C(9 + 3 * which + a, 9 + 3 * which + c);
    InvertDense(3, Cb, inv);
    (which == 0 ? pk.link_info_g : pk.link_info_a).insert((which == 0 ? pk.link_info_g : pk.link_info_a).end(), inv, inv + 9);
  }
}

// one camera model per problem (ImuCamPose::Project goes through pKFi->mpCamera, src/G2oTypes.cc:166-171)
bool NoteFisheye(LibaPack& pk, KeyFrame* pKFi) {
  GeometricCamera* c = pKFi->mpCamera;
  if (c->getParameter(0) != (float)pk.cam[0] || c->getParameter(1) != (float)pk.cam[1] || c->getParameter(2) != (float)pk.cam[2] || c->getParameter(3) != (float)pk.cam[3]) {
    pk.unsupported = "monocular observation through a camera that is not the map's own model"; return false;
  }
  for (int k = 0; k < 4; ++k) {
    if (pk.has_kb8 && pk.kb8[k] != (double)c->getParameter(4 + k)) { pk.unsupported = "keyframes with different KannalaBrandt8 coefficients"; return false; }
    pk.kb8[k] = c->getParameter(4 + k);
  }
  pk.has_kb8 = true;
  return true;
}

void PushEdge(LibaPack& pk, int pose, int point, uint8_t kind, float u, float v, float ur, float invSigma2, KeyFrame* pKFi, MapPoint* pMP) {
  pk.edge_pose.push_back(pose);
  pk.edge_point.push_back(point);
  pk.edge_kind.push_back(kind);
  pk.edge_obs.push_back(u); pk.edge_obs.push_back(v); pk.edge_obs.push_back(kind == OSH_EDGE_STEREO ? ur : -1.0);
  pk.edge_info.push_back(invSigma2);
  pk.vEdgeKF.push_back(pKFi);
  pk.vEdgeMP.push_back(pMP);
}

bool NoteRig(LibaPack& pk, KeyFrame* pKFi) {
  if (pKFi->mpCamera->GetType() != GeometricCamera::CAM_FISHEYE || pKFi->mpCamera2->GetType() != GeometricCamera::CAM_FISHEYE) {
    pk.unsupported = "right-camera observation of a rig that is not a KannalaBrandt8 pair"; return false;
  }
  if (!NoteFisheye(pk, pKFi)) return false;
  double c2[8], T[12];
  for (int k = 0; k < 8; ++k) c2[k] = pKFi->mpCamera2->getParameter(k);
  const Sophus::SE3f Trl = pKFi->GetRelativePoseTrl();
  const Eigen::Matrix3f Rrl = Trl.rotationMatrix();
  for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) T[a * 4 + b] = (double)Rrl(a, b); T[a * 4 + 3] = (double)Trl.translation()(a); }
  if (pk.has_rig) {
    for (int k = 0; k < 8; ++k) if (pk.cam2[k] != c2[k]) { pk.unsupported = "keyframes with different right cameras"; return false; }
    for (int k = 0; k < 12; ++k) if (pk.trl[k] != T[k]) { pk.unsupported = "keyframes with different left-to-right transforms"; return false; }
  }
  std::copy(c2, c2 + 8, pk.cam2); std::copy(T, T + 12, pk.trl);
  pk.has_rig = true;
  return true;
}

// drops the points no edge refers to (g2o never activates them) and renumbers edge_point; `kept` = the surviving points in order
void DropUnobservedPoints(LibaPack& pk, std::vector<MapPoint*>& all) {
  std::vector<int> count(all.size(), 0), remap(all.size(), -1);
  for (int32_t j : pk.edge_point) ++count[j];
  pk.vPointMPs.clear(); pk.points.clear();
  for (size_t j = 0; j < all.size(); ++j)
    if (count[j]) {
      remap[j] = (int)pk.vPointMPs.size();
      pk.vPointMPs.push_back(all[j]);
      const Eigen::Vector3d X = all[j]->GetWorldPos().cast<double>();
      pk.points.push_back(X[0]); pk.points.push_back(X[1]); pk.points.push_back(X[2]);
    }
  for (int32_t& j : pk.edge_point) j = remap[j];
}

struct LibaOutput {
  std::vector<double> Rcw, tcw, Rwb, twb, v, bg, ba, pts, chi;
  std::vector<uint8_t> dep;
  osh_liba_result res;
  LibaOutput(int N, int L, int E) : Rcw((size_t)N * 9), tcw((size_t)N * 3), Rwb((size_t)N * 9), twb((size_t)N * 3), v((size_t)N * 3), bg((size_t)N * 3), ba((size_t)N * 3),
                                    pts((size_t)L * 3), chi(E), dep(E) {
    res.pose_Rcw = Rcw.data(); res.pose_tcw = tcw.data(); res.pose_Rwb = Rwb.data(); res.pose_twb = twb.data();
    res.vel = v.data(); res.bias_g = bg.data(); res.bias_a = ba.data(); res.points = pts.data(); res.edge_chi2 = chi.data(); res.edge_depth_pos = dep.data();
  }
  Sophus::SE3f pose(int i) const {
    Eigen::Matrix3f R; Eigen::Vector3f t;
    for (int a = 0; a < 9; ++a) R.v[a] = (float)Rcw[(size_t)i * 9 + a];
    for (int a = 0; a < 3; ++a) t(a) = (float)tcw[(size_t)i * 3 + a];
    return Sophus::SE3f(R, t);
  }
  Eigen::Vector3f velocity(int i) const { return Eigen::Vector3f((float)v[(size_t)i * 3], (float)v[(size_t)i * 3 + 1], (float)v[(size_t)i * 3 + 2]); }
  IMU::Bias bias(int i) const { return IMU::Bias(ba[(size_t)i * 3], ba[(size_t)i * 3 + 1], ba[(size_t)i * 3 + 2], bg[(size_t)i * 3], bg[(size_t)i * 3 + 1], bg[(size_t)i * 3 + 2]); }
  Eigen::Vector3f point(int j) const { return Eigen::Vector3d(pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2]).cast<float>(); }
};
}  // namespace

// ------------------------------------------------------------------------------------------------
// FullInertialBA: vertices :417-470, inertial links :480-579, points and visual edges :604-727
// ------------------------------------------------------------------------------------------------
bool PackFullInertialBA(Map* pMap, int its, bool bFixLocal, bool bInit, float priorG, float priorA, LibaPack& pk, std::vector<KeyFrame*>& vpIdle,
                        std::vector<MapPoint*>& vpAllMPs, int* sharedBiasSlot) {
  pk = LibaPack();
  vpIdle.clear();
  if (sharedBiasSlot) *sharedBiasSlot = -1;
  if (bFixLocal) { pk.unsupported = "bFixLocal (keyframes of the local window fixed) is not on the device path; the reference never passes it"; return false; }
  const long unsigned int maxKFid = pMap->GetMaxKFid();
  const std::vector<KeyFrame*> vpKFs = pMap->GetAllKeyFrames();
  vpAllMPs = pMap->GetAllMapPoints();
  std::vector<KeyFrame*> vKF;
  for (KeyFrame* k : vpKFs) if (k->mnId <= maxKFid) vKF.push_back(k);
  if (vKF.empty()) return false;
  std::sort(vKF.begin(), vKF.end(), [](KeyFrame* a, KeyFrame* b) { return a->mnId < b->mnId; });   // Hessian order = vertex id = mnId
  // which keyframes take part in an edge at all (the others are vertices without active edges: g2o leaves them untouched)
  std::set<KeyFrame*> sKF(vKF.begin(), vKF.end()), sActive;
  auto link_of = [&](KeyFrame* pKFi) -> bool {   // :488-503
    if (!pKFi->mPrevKF || pKFi->isBad() || pKFi->mPrevKF->mnId > maxKFid) return false;
    return pKFi->bImu && pKFi->mPrevKF->bImu && pKFi->mpImuPreintegrated && sKF.count(pKFi->mPrevKF);
  };
  for (KeyFrame* k : vKF) if (link_of(k)) { sActive.insert(k); sActive.insert(k->mPrevKF); }
  for (MapPoint* pMP : vpAllMPs)
    for (const auto& ob : pMP->GetObservations())
      if (ob.first->mnId <= maxKFid && !ob.first->isBad() && sKF.count(ob.first)) sActive.insert(ob.first);
  for (KeyFrame* k : vKF) (sActive.count(k) ? pk.vPoseKFs : vpIdle).push_back(k);
  if (pk.vPoseKFs.empty()) return false;
  pk.n_opt = (int)pk.vPoseKFs.size(); pk.n_fixed_imu = 0; pk.n_fixed = 0; pk.opt_it = its;
  pk.vpOptimizableKFs = pk.vPoseKFs;
  std::map<KeyFrame*, int> poseIndex;
  for (size_t i = 0; i < pk.vPoseKFs.size(); ++i) poseIndex[pk.vPoseKFs[i]] = (int)i;
  PackKeyframeStates(pk);
  for (KeyFrame* pKFi : vpKFs) {   // the reference's order (the map's), :481
    if (!pKFi->mPrevKF) { std::printf("NOT INERTIAL LINK TO PREVIOUS FRAME!\n"); continue; }
    if (pKFi->mnId > maxKFid) continue;
    if (pKFi->isBad() || pKFi->mPrevKF->mnId > maxKFid) continue;
    if (!(pKFi->bImu && pKFi->mPrevKF->bImu)) { std::printf("%lu or %lu no imu\n", pKFi->mnId, pKFi->mPrevKF->mnId); continue; }
    if (!pKFi->mpImuPreintegrated || !sKF.count(pKFi->mPrevKF)) continue;
    pKFi->mpImuPreintegrated->SetNewBias(pKFi->mPrevKF->GetImuBias());   // :504
    PackLink(pk, pKFi, poseIndex.at(pKFi->mPrevKF), poseIndex.at(pKFi));
  }
  if (bInit) {
    // ONE gyro and ONE accelerometer bias vertex for the whole map, created from the last keyframe of the map's list (:452-462); every
    // EdgeInertial hangs on them (:514-518), there are no random-walk edges (:551), EdgePriorAcc / EdgePriorGyro with prior value 0 hold
    // them (:581-601).  The pair lives in the bias slot of a keyframe no link ends at (osh_liba_problem.link_bias); the priors are the
    // random-walk edges of a link without inertial information from a fixed, virtual keyframe whose biases are the prior value.
    KeyFrame* pIncKF = nullptr;
    for (KeyFrame* k : vpKFs) if (k->mnId <= maxKFid) pIncKF = k;
    int slot = -1;
    for (int i = 0; i < pk.n_opt && slot < 0; ++i)
      if (std::find(pk.link_cur.begin(), pk.link_cur.end(), i) == pk.link_cur.end()) slot = i;
    if (slot < 0) { pk.unsupported = "bInit: every keyframe ends an inertial link (a closed chain of mPrevKF)"; return true; }
    const Eigen::Vector3f bg = pIncKF->GetGyroBias(), ba = pIncKF->GetAccBias();
    for (int a = 0; a < 3; ++a) { pk.bias_g[(size_t)slot * 3 + a] = (double)bg(a); pk.bias_a[(size_t)slot * 3 + a] = (double)ba(a); }
    const size_t NLr = pk.link_prev.size();
    std::fill(pk.link_info_g.begin(), pk.link_info_g.end(), 0.0);
    std::fill(pk.link_info_a.begin(), pk.link_info_a.end(), 0.0);
    pk.link_bias.assign(NLr, slot);
    const int N = pk.n_opt;
    pk.n_fixed_imu = 1;   // the virtual keyframe: the slot keyframe's pose (any would do), zero velocity, biases = bprior = 0 (:586)
    for (int a = 0; a < 9; ++a) { pk.pose_Rcw.push_back(pk.pose_Rcw[(size_t)slot * 9 + a]); pk.pose_Rwb.push_back(pk.pose_Rwb[(size_t)slot * 9 + a]); }
    for (int a = 0; a < 3; ++a) { pk.pose_tcw.push_back(pk.pose_tcw[(size_t)slot * 3 + a]); pk.pose_twb.push_back(pk.pose_twb[(size_t)slot * 3 + a]); }
    for (int a = 0; a < 3; ++a) { pk.vel.push_back(0.0); pk.bias_g.push_back(0.0); pk.bias_a.push_back(0.0); }
    pk.link_prev.push_back(N); pk.link_cur.push_back(slot); pk.link_bias.push_back(N);
    float rec[OSH_PREINT_FLOATS];
    std::memset(rec, 0, sizeof(rec));
    rec[1] = rec[5] = rec[9] = 1.f;   // dR = I, dT = 0: a finite residual, dropped by the zero information
    pk.link_preint.insert(pk.link_preint.end(), rec, rec + OSH_PREINT_FLOATS);
    pk.link_info.insert(pk.link_info.end(), 81, 0.0);
    pk.link_robust.push_back(0);
    const double infoPriorG = priorG, infoPriorA = priorA;   // :589,595
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) { pk.link_info_g.push_back(a == c ? infoPriorG : 0.0); pk.link_info_a.push_back(a == c ? infoPriorA : 0.0); }
    if (sharedBiasSlot) *sharedBiasSlot = slot;
  }
  // points in map order; vertex id = mnId + 5 maxKFid + 1, so the Hessian order is ascending mnId
  std::vector<MapPoint*> vMP(vpAllMPs);
  std::sort(vMP.begin(), vMP.end(), [](MapPoint* a, MapPoint* b) { return a->mnId < b->mnId; });
  std::map<MapPoint*, int> pointIndex;
  for (size_t j = 0; j < vMP.size(); ++j) pointIndex[vMP[j]] = (int)j;
  for (MapPoint* pMP : vpAllMPs) {
    const std::map<KeyFrame*, std::tuple<int, int>> observations = pMP->GetObservations();
    const int j = pointIndex.at(pMP);
    for (const auto& ob : observations) {
      KeyFrame* pKFi = ob.first;
      if (pKFi->mnId > maxKFid || pKFi->isBad()) continue;
      auto itk = poseIndex.find(pKFi);
      if (itk == poseIndex.end()) continue;
      const int leftIndex = std::get<0>(ob.second);
      if (leftIndex != -1) {   // mono (:631-655) or stereo (:656-682); information = invSigma2 of the octave, no uncertainty factor here
        const cv::KeyPoint kpUn = pKFi->mvKeysUn[leftIndex];
        const float kp_ur = pKFi->mvuRight[leftIndex];
        const bool stereo = !(kp_ur < 0);
        if (!stereo && pKFi->mpCamera && pKFi->mpCamera->GetType() == GeometricCamera::CAM_FISHEYE && !NoteFisheye(pk, pKFi)) return true;
        PushEdge(pk, itk->second, j, stereo ? OSH_EDGE_STEREO : OSH_EDGE_MONO, kpUn.pt.x, kpUn.pt.y, kp_ur, pKFi->mvInvLevelSigma2[kpUn.octave], pKFi, pMP);
      }
      if (pKFi->mpCamera2) {   // :684-716: the index is compared with mvKeysRight.size() BEFORE NLeft is taken off, as the reference has it
        int rightIndex = std::get<1>(ob.second);
        if (rightIndex != -1 && rightIndex < (int)pKFi->mvKeysRight.size()) {
          rightIndex -= pKFi->NLeft;
          if (rightIndex < 0) { pk.unsupported = "right-camera index below NLeft"; return true; }
          if (!NoteRig(pk, pKFi)) return true;
          const cv::KeyPoint kpUn = pKFi->mvKeysRight[rightIndex];
          PushEdge(pk, itk->second, j, OSH_EDGE_RIGHT, kpUn.pt.x, kpUn.pt.y, -1.f, pKFi->mvInvLevelSigma2[kpUn.octave], pKFi, pMP);
        }
      }
    }
  }
  DropUnobservedPoints(pk, vMP);   // bAllFixed stays true without an edge: the vertex is removed (:719-725)
  if (pk.has_kb8)
    for (uint8_t k : pk.edge_kind) if (k == OSH_EDGE_STEREO) { pk.unsupported = "rectified-stereo observation in a KannalaBrandt8 map"; return true; }
  return !pk.edge_pose.empty() || !pk.link_prev.empty();
}

void Optimizer::FullInertialBA(Map* pMap, int its, const bool bFixLocal, const unsigned long nLoopId, bool* pbStopFlag, bool bInit, float priorG,
                               float priorA, Eigen::VectorXd* vSingVal, bool* bHess) {
  (void)vSingVal; (void)bHess;
  LibaPack pk;
  std::vector<KeyFrame*> vpIdle;
  std::vector<MapPoint*> vpAllMPs;
  int slot = -1;
  const bool packed = PackFullInertialBA(pMap, its, bFixLocal, bInit, priorG, priorA, pk, vpIdle, vpAllMPs, &slot);
  if (!packed || pk.unsupported) {
    std::fprintf(stderr, "FullInertialBA: %s; map left untouched\n", pk.unsupported ? pk.unsupported : "nothing to optimise");
    return;
  }
  if (pbStopFlag && *pbStopFlag) return;   // :729-731 (inside optimize() the flag ends the run between iterations; one launch here)
  osh_lba_ctx* ctx = HostSolverContext();
  if (!ctx) return;
  osh_liba_problem prob;
  pk.fill(prob);
  prob.huber_mono = (double)(float)std::sqrt(5.991);     // :592
  prob.huber_stereo = (double)(float)std::sqrt(7.815);   // :593
  prob.huber_inertial = std::sqrt(16.92);                // :541
  prob.lambda_init = 1e-5;                               // setUserLambdaInit(1e-5) :408
  prob.max_iterations = its;
  const int N = pk.n_opt, L = (int)pk.vPointMPs.size(), E = (int)pk.edge_pose.size();
  LibaOutput out(N, L, E);
  if (osh_liba_solve(ctx, 1, &prob, &out.res) != OSH_OK) {
    std::fprintf(stderr, "FullInertialBA: device solve failed (%s); map left untouched\n", osh_last_error());
    return;
  }
  // recover optimised data (:738-811): into the live map, or beside it for the loop closer (nLoopId != 0)
  auto write_kf = [&](KeyFrame* pKFi, const Sophus::SE3f& Tcw, const Eigen::Vector3f& v, const IMU::Bias& b) {
    if (nLoopId == 0) pKFi->SetPose(Tcw);
    else { pKFi->mTcwGBA = Tcw; pKFi->mnBAGlobalForKF = nLoopId; }
    if (pKFi->bImu) {
      if (nLoopId == 0) { pKFi->SetVelocity(v); pKFi->SetNewBias(b); }
      else { pKFi->mVwbGBA = v; pKFi->mBiasGBA = b; }
    }
  };
  for (int i = 0; i < N; ++i) {
    KeyFrame* pKFi = pk.vPoseKFs[i];
    const bool linked = std::find(pk.link_prev.begin(), pk.link_prev.end(), i) != pk.link_prev.end() || std::find(pk.link_cur.begin(), pk.link_cur.end(), i) != pk.link_cur.end();
    // velocity / bias vertices without an edge keep their float values (the device's increment there is exactly zero as well)
    // with bInit every keyframe is handed the one optimised bias pair (:779-789)
    write_kf(pKFi, out.pose(i), linked ? out.velocity(i) : pKFi->GetVelocity(), slot >= 0 ? out.bias(slot) : linked ? out.bias(i) : pKFi->GetImuBias());
  }
  for (KeyFrame* pKFi : vpIdle)   // vertices no edge touches: the estimate the vertex was created with
    write_kf(pKFi, Sophus::SE3f(pKFi->GetRotation(), pKFi->GetTranslation()), pKFi->GetVelocity(), slot >= 0 ? out.bias(slot) : pKFi->GetImuBias());
  for (int j = 0; j < L; ++j) {
    MapPoint* pMP = pk.vPointMPs[j];
    if (nLoopId == 0) { pMP->SetWorldPos(out.point(j)); pMP->UpdateNormalAndDepth(); }
    else { pMP->mPosGBA = out.point(j); pMP->mnBAGlobalForKF = nLoopId; }
  }
  pMap->IncreaseChangeIndex();
}

// ------------------------------------------------------------------------------------------------
// MergeInertialBA: keyframe selection :3958-4114, vertices :4126-4197, links :4203-4262, visual edges :4290-4382
// ------------------------------------------------------------------------------------------------
bool PackMergeInertialBA(KeyFrame* pCurrKF, KeyFrame* pMergeKF, LibaPack& pk, std::vector<KeyFrame*>& vpCovKFs) {
  pk = LibaPack();
  vpCovKFs.clear();
  const int Nd = 6;
  const unsigned long maxKFid = pCurrKF->mnId;
  std::vector<KeyFrame*>& vpOptimizableKFs = pk.vpOptimizableKFs;
  vpOptimizableKFs.reserve(2 * Nd);
  const int maxCovKF = 30;
  std::vector<KeyFrame*>& vpOptimizableCovKFs = vpCovKFs;
  vpOptimizableCovKFs.reserve(maxCovKF);
  // the current keyframe and its predecessors (:3972-3984); the one before them joins the covisible (pose-only) set (:3988-3998)
  vpOptimizableKFs.push_back(pCurrKF);
  pCurrKF->mnBALocalForKF = pCurrKF->mnId;
  for (int i = 1; i < Nd; i++) {
    if (!vpOptimizableKFs.back()->mPrevKF) break;
    vpOptimizableKFs.push_back(vpOptimizableKFs.back()->mPrevKF);
    vpOptimizableKFs.back()->mnBALocalForKF = pCurrKF->mnId;
  }
  if (vpOptimizableKFs.back()->mPrevKF) {
    vpOptimizableCovKFs.push_back(vpOptimizableKFs.back()->mPrevKF);
    vpOptimizableKFs.back()->mPrevKF->mnBALocalForKF = pCurrKF->mnId;
  } else {
    vpOptimizableCovKFs.push_back(vpOptimizableKFs.back());
    vpOptimizableKFs.pop_back();
  }
  // the merge keyframe, two of its predecessors, one fixed keyframe before them, then its successors up to 2 Nd (:4001-4049)
  vpOptimizableKFs.push_back(pMergeKF);
  pMergeKF->mnBALocalForKF = pCurrKF->mnId;
  for (int i = 1; i < (Nd / 2); i++) {
    if (!vpOptimizableKFs.back()->mPrevKF) break;
    vpOptimizableKFs.push_back(vpOptimizableKFs.back()->mPrevKF);
    vpOptimizableKFs.back()->mnBALocalForKF = pCurrKF->mnId;
  }
  if (vpOptimizableKFs.back()->mPrevKF) {
    pk.lFixedKeyFrames.push_back(vpOptimizableKFs.back()->mPrevKF);
    vpOptimizableKFs.back()->mPrevKF->mnBAFixedForKF = pCurrKF->mnId;
  } else {
    vpOptimizableKFs.back()->mnBALocalForKF = 0;
    vpOptimizableKFs.back()->mnBAFixedForKF = pCurrKF->mnId;
    pk.lFixedKeyFrames.push_back(vpOptimizableKFs.back());
    vpOptimizableKFs.pop_back();
  }
  if (pMergeKF->mNextKF) {
    vpOptimizableKFs.push_back(pMergeKF->mNextKF);
    vpOptimizableKFs.back()->mnBALocalForKF = pCurrKF->mnId;
  }
  while ((int)vpOptimizableKFs.size() < 2 * Nd) {
    if (!vpOptimizableKFs.back()->mNextKF) break;
    vpOptimizableKFs.push_back(vpOptimizableKFs.back()->mNextKF);
    vpOptimizableKFs.back()->mnBALocalForKF = pCurrKF->mnId;
  }
  const int N = (int)vpOptimizableKFs.size();
  // points of the temporal keyframes with the number of those keyframes that see them (:4054-4078)
  std::map<MapPoint*, int> mLocalObs;
  for (int i = 0; i < N; i++)
    for (MapPoint* pMP : vpOptimizableKFs[i]->GetMapPointMatches())
      if (pMP && !pMP->isBad()) {
        if (pMP->mnBALocalForKF != pCurrKF->mnId) { mLocalObs[pMP] = 1; pk.lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pCurrKF->mnId; }
        else mLocalObs[pMP]++;
      }
  // covisible keyframes: the first unmarked observer of each of the 30 first points in sortByVal order (:4080-4114)
  std::vector<std::pair<MapPoint*, int>> pairs;
  pairs.reserve(mLocalObs.size());
  for (auto itr = mLocalObs.begin(); itr != mLocalObs.end(); ++itr) pairs.push_back(*itr);
  std::sort(pairs.begin(), pairs.end(), [](const std::pair<MapPoint*, int>& a, const std::pair<MapPoint*, int>& b) { return a.second < b.second; });
  int i = 0;
  for (auto lit = pairs.begin(); lit != pairs.end(); ++lit, ++i) {
    const std::map<KeyFrame*, std::tuple<int, int>> observations = lit->first->GetObservations();
    if (i >= maxCovKF) break;
    for (const auto& ob : observations) {
      KeyFrame* pKFi = ob.first;
      if (pKFi->mnBALocalForKF != pCurrKF->mnId && pKFi->mnBAFixedForKF != pCurrKF->mnId) {
        pKFi->mnBALocalForKF = pCurrKF->mnId;
        if (!pKFi->isBad()) { vpOptimizableCovKFs.push_back(pKFi); break; }
      }
    }
  }
  if (N == 0) { pk.unsupported = "no temporal keyframe"; return false; }
  // problem order: every optimisable keyframe in Hessian order (ascending id), then the fixed one
  std::vector<KeyFrame*> vOpt(vpOptimizableKFs.begin(), vpOptimizableKFs.end());
  vOpt.insert(vOpt.end(), vpOptimizableCovKFs.begin(), vpOptimizableCovKFs.end());
  std::sort(vOpt.begin(), vOpt.end(), [](KeyFrame* a, KeyFrame* b) { return a->mnId < b->mnId; });
  if (std::adjacent_find(vOpt.begin(), vOpt.end()) != vOpt.end()) { pk.unsupported = "the two temporal chains overlap"; return false; }
  KeyFrame* pFixed = pk.lFixedKeyFrames.front();
  bool fixedLinked = false;
  for (KeyFrame* o : vpOptimizableKFs) if (o->mPrevKF == pFixed && o->bImu && pFixed->bImu && o->mpImuPreintegrated) fixedLinked = true;
  pk.vPoseKFs = vOpt;
  pk.vPoseKFs.push_back(pFixed);
  pk.n_opt = (int)vOpt.size(); pk.n_fixed_imu = fixedLinked ? 1 : 0; pk.n_fixed = fixedLinked ? 0 : 1; pk.opt_it = 8;
  std::map<KeyFrame*, int> poseIndex;
  for (size_t k = 0; k < pk.vPoseKFs.size(); ++k) poseIndex[pk.vPoseKFs[k]] = (int)k;
  PackKeyframeStates(pk);
  for (int k = 0; k < N; k++) {   // :4203-4262
    KeyFrame* pKFi = vpOptimizableKFs[k];
    if (!pKFi->mPrevKF) { std::printf("NOT INERTIAL LINK TO PREVIOUS FRAME!!!!\n"); continue; }
    if (!(pKFi->bImu && pKFi->mPrevKF->bImu && pKFi->mpImuPreintegrated)) { std::printf("ERROR building inertial edge\n"); continue; }
    pKFi->mpImuPreintegrated->SetNewBias(pKFi->mPrevKF->GetImuBias());
    auto itp = poseIndex.find(pKFi->mPrevKF);
    if (itp == poseIndex.end()) { std::fprintf(stderr, "Error: inertial edge to a keyframe without vertices\n"); continue; }   // :4224-4228
    PackLink(pk, pKFi, itp->second, poseIndex.at(pKFi));
  }
  // points and visual edges (:4290-4382): an observation's keyframe must carry the window's mark AND a vertex
  std::vector<MapPoint*> vMP(pk.lLocalMapPoints.begin(), pk.lLocalMapPoints.end());
  std::sort(vMP.begin(), vMP.end(), [](MapPoint* a, MapPoint* b) { return a->mnId < b->mnId; });
  std::map<MapPoint*, int> pointIndex;
  for (size_t j = 0; j < vMP.size(); ++j) pointIndex[vMP[j]] = (int)j;
  for (MapPoint* pMP : pk.lLocalMapPoints) {
    const std::map<KeyFrame*, std::tuple<int, int>> observations = pMP->GetObservations();
    for (const auto& ob : observations) {
      KeyFrame* pKFi = ob.first;
      if (!pKFi) continue;
      if (pKFi->mnBALocalForKF != pCurrKF->mnId && pKFi->mnBAFixedForKF != pCurrKF->mnId) continue;
      if (pKFi->mnId > maxKFid) continue;
      auto itk = poseIndex.find(pKFi);
      if (itk == poseIndex.end() || pKFi->isBad()) continue;
      const int leftIndex = std::get<0>(ob.second);
      if (leftIndex < 0) { pk.unsupported = "observation without a left keypoint (the reference indexes mvKeysUn with it)"; return true; }
      const cv::KeyPoint& kpUn = pKFi->mvKeysUn[leftIndex];
      const float kp_ur = pKFi->mvuRight[leftIndex];
      const bool stereo = !(kp_ur < 0);
      if (!stereo && pKFi->mpCamera && pKFi->mpCamera->GetType() == GeometricCamera::CAM_FISHEYE && !NoteFisheye(pk, pKFi)) return true;
      PushEdge(pk, itk->second, pointIndex.at(pMP), stereo ? OSH_EDGE_STEREO : OSH_EDGE_MONO, kpUn.pt.x, kpUn.pt.y, kp_ur, pKFi->mvInvLevelSigma2[kpUn.octave], pKFi, pMP);
    }
  }
  DropUnobservedPoints(pk, vMP);
  if (pk.has_kb8)
    for (uint8_t k : pk.edge_kind) if (k == OSH_EDGE_STEREO) { pk.unsupported = "rectified-stereo observation in a KannalaBrandt8 map"; return true; }
  return true;
}

void Optimizer::MergeInertialBA(KeyFrame* pCurrKF, KeyFrame* pMergeKF, bool* pbStopFlag, Map* pMap, LoopClosing::KeyFrameAndPose& corrPoses) {
  LibaPack pk;
  std::vector<KeyFrame*> vpCovKFs;
  const bool packed = PackMergeInertialBA(pCurrKF, pMergeKF, pk, vpCovKFs);
  if (!packed || pk.unsupported) {
    std::fprintf(stderr, "MergeInertialBA: %s; map left untouched\n", pk.unsupported ? pk.unsupported : "nothing to optimise");
    return;
  }
  if (pbStopFlag && *pbStopFlag) return;   // :4386-4388
  osh_lba_ctx* ctx = HostSolverContext();
  if (!ctx) return;
  osh_liba_problem prob;
  pk.fill(prob);
  prob.huber_mono = (double)(float)std::sqrt(5.991);     // :4283
  prob.huber_stereo = (double)(float)std::sqrt(7.815);   // :4285
  prob.huber_inertial = std::sqrt(16.92);                // :4244
  prob.lambda_init = 1e3;                                // setUserLambdaInit(1e3) :4121
  prob.max_iterations = 8;                               // optimizer.optimize(8) :4391
  const int N = pk.n_opt, L = (int)pk.vPointMPs.size(), E = (int)pk.edge_pose.size();
  LibaOutput out(N, L, E);
  if (osh_liba_solve(ctx, 1, &prob, &out.res) != OSH_OK) {
    std::fprintf(stderr, "MergeInertialBA: device solve failed (%s); map left untouched\n", osh_last_error());
    return;
  }
  // outliers (:4393-4427): chi2 against the float thresholds, no depth test; monocular edges first
  const float chi2Mono2 = 5.991f, chi2Stereo2 = 7.815f;
  std::vector<std::pair<KeyFrame*, MapPoint*>> vToErase;
  for (int pass = 0; pass < 2; ++pass)
    for (int e = 0; e < E; ++e) {
      if ((pk.edge_kind[e] == OSH_EDGE_STEREO) != (pass == 1)) continue;
      MapPoint* pMP = pk.vEdgeMP[e];
      if (pMP->isBad()) continue;
      if (out.chi[e] > (pass == 0 ? chi2Mono2 : chi2Stereo2)) vToErase.push_back(std::make_pair(pk.vEdgeKF[e], pMP));
    }
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
  for (auto& er : vToErase) { er.first->EraseMapPointMatch(er.second); er.second->EraseObservation(er.first); }
  // recover optimised data (:4444-4495): temporal keyframes, then covisible ones; every corrected pose also goes into corrPoses
  std::map<KeyFrame*, int> poseIndex;
  for (int i = 0; i < N; ++i) poseIndex[pk.vPoseKFs[i]] = i;
  auto write_kf = [&](KeyFrame* pKFi) {
    const int i = poseIndex.at(pKFi);
    pKFi->SetPose(out.pose(i));
    const Sophus::SE3d Tiw = pKFi->GetPose().cast<double>();
    corrPoses[pKFi] = g2o::Sim3(Tiw.unit_quaternion(), Tiw.translation(), 1.0);
    if (pKFi->bImu) {
      const bool linked = std::find(pk.link_prev.begin(), pk.link_prev.end(), i) != pk.link_prev.end() || std::find(pk.link_cur.begin(), pk.link_cur.end(), i) != pk.link_cur.end();
      pKFi->SetVelocity(linked ? out.velocity(i) : pKFi->GetVelocity());
      pKFi->SetNewBias(linked ? out.bias(i) : pKFi->GetImuBias());
    }
  };
  for (KeyFrame* pKFi : pk.vpOptimizableKFs) write_kf(pKFi);
  for (KeyFrame* pKFi : vpCovKFs) write_kf(pKFi);
  std::map<MapPoint*, int> pointIndex;
  for (int j = 0; j < L; ++j) pointIndex[pk.vPointMPs[j]] = j;
  for (MapPoint* pMP : pk.lLocalMapPoints) {
    auto it = pointIndex.find(pMP);
    pMP->SetWorldPos(it == pointIndex.end() ? pMP->GetWorldPos() : out.point(it->second));   // a point without an edge keeps its position
    pMP->UpdateNormalAndDepth();
  }
  pMap->IncreaseChangeIndex();
}

}  // namespace ORB_SLAM3
