// OptimizerInertial.cc -- ORB_SLAM3::Optimizer::LocalInertialBA on MI355X (host side).
//
// Graph walk, information matrices, outlier rules, divergence test and write-back of the reference
// (src/Optimizer.cc:2387-2964); the optimisation itself (:2843-2848) runs in the persistent HIP kernel behind
// osh_liba_solve.  The 9x9 EdgeInertial information (src/G2oTypes.cc:492-511: inverse, symmetrise, clamp the
// eigenvalues below 1e-12) is computed here once per link, as the reference does in the edge constructor.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <mutex>

#include "Optimizer.h"
#include "host_pack.h"
#include "orbslam3_hip.h"

namespace ORB_SLAM3 {

// ------------------------------------------------------------------------------------------------
// small dense helpers (double)
// ------------------------------------------------------------------------------------------------
bool InvertDense(int n, const double* A, double* inv) {   // Gauss-Jordan with partial pivoting (Eigen: PartialPivLU)
  std::vector<double> M((size_t)n * 2 * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { M[(size_t)i * 2 * n + j] = A[i * n + j]; M[(size_t)i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r) if (std::fabs(M[(size_t)r * 2 * n + c]) > std::fabs(M[(size_t)piv * 2 * n + c])) piv = r;
    if (M[(size_t)piv * 2 * n + c] == 0.0) return false;
    if (piv != c) for (int j = 0; j < 2 * n; ++j) std::swap(M[(size_t)c * 2 * n + j], M[(size_t)piv * 2 * n + j]);
    const double d = M[(size_t)c * 2 * n + c];
    for (int j = 0; j < 2 * n; ++j) M[(size_t)c * 2 * n + j] /= d;
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = M[(size_t)r * 2 * n + c];
      if (f != 0.0) for (int j = 0; j < 2 * n; ++j) M[(size_t)r * 2 * n + j] -= f * M[(size_t)c * 2 * n + j];
    }
  }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) inv[i * n + j] = M[(size_t)i * 2 * n + n + j];
  return true;
}

// cyclic Jacobi eigen-decomposition of a symmetric matrix: A = V diag(w) V^T
void SymmetricEigen(int n, const double* A, double* w, double* V) {
  std::vector<double> a(A, A + (size_t)n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) { diag += a[(size_t)i * n + i] * a[(size_t)i * n + i]; for (int j = i + 1; j < n; ++j) off += a[(size_t)i * n + j] * a[(size_t)i * n + j]; }
    if (off <= 1e-32 * diag) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double theta = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
          a[(size_t)k * n + p] = c * akp - s * akq; a[(size_t)k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
          a[(size_t)p * n + k] = c * apk - s * aqk; a[(size_t)q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = a[(size_t)i * n + i];
}

// EdgeInertial::EdgeInertial information (src/G2oTypes.cc:500-508)
void InertialInformation(const Eigen::Matrix<float, 15, 15>& C, double* info81) {
  double Cd[81], inv[81], w[9], V[81];
  for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) Cd[i * 9 + j] = (double)C(i, j);
  InvertDense(9, Cd, inv);
  for (int i = 0; i < 9; ++i) for (int j = i; j < 9; ++j) { const double s = (inv[i * 9 + j] + inv[j * 9 + i]) / 2; inv[i * 9 + j] = inv[j * 9 + i] = s; }
  SymmetricEigen(9, inv, w, V);
  for (int i = 0; i < 9; ++i) if (w[i] < 1e-12) w[i] = 0;
  for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) { double s = 0; for (int k = 0; k < 9; ++k) s += V[i * 9 + k] * w[k] * V[j * 9 + k]; info81[i * 9 + j] = s; }
}

// ------------------------------------------------------------------------------------------------
// steps of src/Optimizer.cc:2389-2506 (window selection) and :2531-2832 (vertices / edges) -> flat arrays
// ------------------------------------------------------------------------------------------------
bool PackLocalInertialBA(KeyFrame* pKF, Map* pMap, bool bLarge, bool bRecInit, LibaPack& pk) {
  pk = LibaPack();
  Map* pCurrentMap = pKF->GetMap();
  int maxOpt = 10;
  pk.opt_it = 10;
  if (bLarge) { maxOpt = 25; pk.opt_it = 4; }
  const int Nd = std::min((int)pCurrentMap->KeyFramesInMap() - 2, maxOpt);
  std::vector<KeyFrame*>& vpOptimizableKFs = pk.vpOptimizableKFs;
  vpOptimizableKFs.reserve(std::max(Nd, 1));
  vpOptimizableKFs.push_back(pKF);
  pKF->mnBALocalForKF = pKF->mnId;
  for (int i = 1; i < Nd; i++) {
    if (!vpOptimizableKFs.back()->mPrevKF) break;
    vpOptimizableKFs.push_back(vpOptimizableKFs.back()->mPrevKF);
    vpOptimizableKFs.back()->mnBALocalForKF = pKF->mnId;
  }
  int N = (int)vpOptimizableKFs.size();
  // optimisable points seen by the temporal window (:2421-2436)
  for (int i = 0; i < N; i++)
    for (MapPoint* pMP : vpOptimizableKFs[i]->GetMapPointMatches())
      if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) { pk.lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pKF->mnId; }
  // fixed keyframe: the one before the window, else the oldest window keyframe itself (:2439-2451)
  if (vpOptimizableKFs.back()->mPrevKF) {
    pk.lFixedKeyFrames.push_back(vpOptimizableKFs.back()->mPrevKF);
    vpOptimizableKFs.back()->mPrevKF->mnBAFixedForKF = pKF->mnId;
  } else {
    vpOptimizableKFs.back()->mnBALocalForKF = 0;
    vpOptimizableKFs.back()->mnBAFixedForKF = pKF->mnId;
    pk.lFixedKeyFrames.push_back(vpOptimizableKFs.back());
    vpOptimizableKFs.pop_back();
  }
  // maxCovKF = 0: no extra optimisable visual keyframes (:2454-2481)
  // fixed observers: the FIRST not-yet-seen observer of every local point, at most 200 (:2484-2506)
  const size_t maxFixKF = 200;
  for (MapPoint* pMP : pk.lLocalMapPoints) {
    const std::map<KeyFrame*, std::tuple<int, int>> observations = pMP->GetObservations();
    for (const auto& ob : observations) {
      KeyFrame* pKFi = ob.first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
        pKFi->mnBAFixedForKF = pKF->mnId;
        if (!pKFi->isBad()) { pk.lFixedKeyFrames.push_back(pKFi); break; }
      }
    }
    if (pk.lFixedKeyFrames.size() >= maxFixKF) break;
  }
  N = (int)vpOptimizableKFs.size();
  if (N == 0) { pk.unsupported = "empty temporal window"; return false; }
  // ---- pose order: temporal keyframes ascending id (Hessian order), then the fixed ones; IMU-carrying fixed first
  std::vector<KeyFrame*> vOpt(vpOptimizableKFs.begin(), vpOptimizableKFs.end());
  std::sort(vOpt.begin(), vOpt.end(), [](KeyFrame* a, KeyFrame* b) { return a->mnId < b->mnId; });
  std::vector<KeyFrame*> vFixImu, vFix;
  for (KeyFrame* k : pk.lFixedKeyFrames) {
    bool linked = false;   // only the keyframe just before the window takes part in an inertial edge
    for (KeyFrame* o : vOpt) if (o->mPrevKF == k) linked = true;
    ((k->bImu && linked) ? vFixImu : vFix).push_back(k);
  }
  if (vFixImu.size() > 1) { pk.unsupported = "more than one fixed inertial predecessor"; return false; }
  pk.vPoseKFs = vOpt;
  pk.vPoseKFs.insert(pk.vPoseKFs.end(), vFixImu.begin(), vFixImu.end());
  pk.vPoseKFs.insert(pk.vPoseKFs.end(), vFix.begin(), vFix.end());
  pk.n_opt = (int)vOpt.size(); pk.n_fixed_imu = (int)vFixImu.size(); pk.n_fixed = (int)vFix.size();
  std::map<KeyFrame*, int> poseIndex;
  for (size_t i = 0; i < pk.vPoseKFs.size(); ++i) poseIndex[pk.vPoseKFs[i]] = (int)i;
  for (KeyFrame* k : vOpt) if (!k->bImu) { pk.unsupported = "temporal keyframe without IMU"; return false; }
  // ImuCamPose(KeyFrame*) (src/G2oTypes.cc:25-71): float members widened to double
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
  {
    const IMU::Calib& cal = pKF->mImuCalib;
    const Eigen::Matrix3f Rcb = cal.mTcb.rotationMatrix();
    for (int a = 0; a < 9; ++a) pk.Rcb[a] = (double)Rcb.v[a];
    for (int a = 0; a < 3; ++a) { pk.tcb[a] = (double)cal.mTcb.translation()(a); pk.tbc[a] = (double)cal.mTbc.translation()(a); }
    pk.cam[0] = pKF->fx; pk.cam[1] = pKF->fy; pk.cam[2] = pKF->cx; pk.cam[3] = pKF->cy; pk.cam[4] = pKF->mbf;
  }
  // ---- inertial links (:2600-2667), in the reference's newest-first order i = 0..N-1
  for (int i = 0; i < N; i++) {
    KeyFrame* pKFi = vpOptimizableKFs[i];
    if (!pKFi->mPrevKF) { std::printf("NOT INERTIAL LINK TO PREVIOUS FRAME!!!!\n"); continue; }
    if (!(pKFi->bImu && pKFi->mPrevKF->bImu && pKFi->mpImuPreintegrated)) { std::printf("ERROR building inertial edge\n"); continue; }
    pKFi->mpImuPreintegrated->SetNewBias(pKFi->mPrevKF->GetImuBias());
    auto itp = poseIndex.find(pKFi->mPrevKF);
    if (itp == poseIndex.end() || itp->second >= pk.n_opt + pk.n_fixed_imu) continue;   // vertex missing (:2625-2629)
    IMU::Preintegrated* P = pKFi->mpImuPreintegrated;
    pk.link_prev.push_back(itp->second);
    pk.link_cur.push_back(poseIndex.at(pKFi));
    float rec[OSH_PREINT_FLOATS];
    std::memset(rec, 0, sizeof(rec));
    rec[0] = P->dT;
    for (int a = 0; a < 9; ++a) { rec[1 + a] = P->dR.v[a]; rec[16 + a] = P->JRg.v[a]; rec[25 + a] = P->JVg.v[a]; rec[34 + a] = P->JVa.v[a]; rec[43 + a] = P->JPg.v[a]; rec[52 + a] = P->JPa.v[a]; }
    for (int a = 0; a < 3; ++a) { rec[10 + a] = P->dV(a); rec[13 + a] = P->dP(a); }
    rec[61] = P->b.bax; rec[62] = P->b.bay; rec[63] = P->b.baz; rec[64] = P->b.bwx; rec[65] = P->b.bwy; rec[66] = P->b.bwz;
    pk.link_preint.insert(pk.link_preint.end(), rec, rec + OSH_PREINT_FLOATS);
    double info[81];
    InertialInformation(P->C, info);
    const bool robust = (i == N - 1) || bRecInit;
    if (i == N - 1) for (double& x : info) x *= 1e-2;   // the link to the fixed keyframe is down-weighted (:2644-2645)
    pk.link_info.insert(pk.link_info.end(), info, info + 81);
    pk.link_robust.push_back(robust ? 1 : 0);
    for (int which = 0; which < 2; ++which) {
      double Cb[9], inv[9];
      for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) Cb[a * 3 + c] = (double)P->C(9 + 3 * which + a, 9 + 3 * which + c);
      InvertDense(3, Cb, inv);
      (which == 0 ? pk.link_info_g : pk.link_info_a).insert((which == 0 ? pk.link_info_g : pk.link_info_a).end(), inv, inv + 9);
    }
  }
  // ---- points and visual edges (:2694-2832)
  pk.vPointMPs.assign(pk.lLocalMapPoints.begin(), pk.lLocalMapPoints.end());
  std::sort(pk.vPointMPs.begin(), pk.vPointMPs.end(), [](MapPoint* a, MapPoint* b) { return a->mnId < b->mnId; });
  std::map<MapPoint*, int> pointIndex;
  for (size_t j = 0; j < pk.vPointMPs.size(); ++j) {
    pointIndex[pk.vPointMPs[j]] = (int)j;
    const Eigen::Vector3d X = pk.vPointMPs[j]->GetWorldPos().cast<double>();
    pk.points.push_back(X[0]); pk.points.push_back(X[1]); pk.points.push_back(X[2]);
  }
  for (MapPoint* pMP : pk.lLocalMapPoints) {
    const std::map<KeyFrame*, std::tuple<int, int>> observations = pMP->GetObservations();
    for (const auto& ob : observations) {
      KeyFrame* pKFi = ob.first;
      if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) continue;
      if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;
      auto itk = poseIndex.find(pKFi);
      if (itk == poseIndex.end()) continue;   // marked fixed but never added (the `break` above / the 200 cap)
      const int leftIndex = std::get<0>(ob.second);
      cv::KeyPoint kpUn;   // declared per observation like the reference's (:2732): default-constructed unless there is a left observation
      // one KannalaBrandt8 per window: EdgeMono projects through pKFi->mpCamera (ImuCamPose::Project, src/G2oTypes.cc:166-171)
      auto window_fisheye = [&]() -> bool {
        GeometricCamera* c = pKFi->mpCamera;
        if (c->getParameter(0) != pKF->fx || c->getParameter(1) != pKF->fy || c->getParameter(2) != pKF->cx || c->getParameter(3) != pKF->cy) {
          pk.unsupported = "monocular observation through a camera that is not the window's own model"; return false;
        }
        for (int k = 0; k < 4; ++k) {
          if (pk.has_kb8 && pk.kb8[k] != (double)c->getParameter(4 + k)) { pk.unsupported = "keyframes with different KannalaBrandt8 coefficients"; return false; }
          pk.kb8[k] = c->getParameter(4 + k);
        }
        pk.has_kb8 = true;
        return true;
      };
      if (leftIndex != -1) {
        kpUn = pKFi->mvKeysUn[leftIndex];
        const float kp_ur = pKFi->mvuRight[leftIndex];
        const bool stereo = !(kp_ur < 0);
        if (!stereo && pKFi->mpCamera && pKFi->mpCamera->GetType() == GeometricCamera::CAM_FISHEYE && !window_fisheye()) return true;
        Eigen::Matrix<double, 2, 1> obs2(kpUn.pt.x, kpUn.pt.y);
        const float unc2 = pKFi->mpCamera->uncertainty2(obs2);
        const float invSigma2 = pKFi->mvInvLevelSigma2[kpUn.octave] / unc2;   // :2741, float division
        pk.edge_pose.push_back(itk->second);
        pk.edge_point.push_back(pointIndex.at(pMP));
        pk.edge_kind.push_back(stereo ? OSH_EDGE_STEREO : OSH_EDGE_MONO);
        pk.edge_obs.push_back(kpUn.pt.x); pk.edge_obs.push_back(kpUn.pt.y); pk.edge_obs.push_back(stereo ? kp_ur : -1.0);
        pk.edge_info.push_back(invSigma2);
        pk.vEdgeKF.push_back(pKFi);
        pk.vEdgeMP.push_back(pMP);
      }
      // monocular right observation (:2798-2835): EdgeMono(1) on camera 1 of the keyframe's ImuCamPose
      if (pKFi->mpCamera2) {
        int rightIndex = std::get<1>(ob.second);
        if (rightIndex != -1) {
          rightIndex -= pKFi->NLeft;
          // (the reference reads mvKeysRight[rightIndex] without a check, :2803; an observation tuple whose right slot lies below NLeft
          // would read out of bounds: declined like the same case of FullInertialBA)
          if (rightIndex < 0 || rightIndex >= (int)pKFi->mvKeysRight.size()) { pk.unsupported = "right-camera index outside mvKeysRight"; return true; }
          if (pKFi->mpCamera->GetType() != GeometricCamera::CAM_FISHEYE || pKFi->mpCamera2->GetType() != GeometricCamera::CAM_FISHEYE) {
            pk.unsupported = "right-camera observation of a rig that is not a KannalaBrandt8 pair"; return true;
          }
          if (!window_fisheye()) return true;
          double c2[8], T[12];
          for (int k = 0; k < 8; ++k) c2[k] = pKFi->mpCamera2->getParameter(k);
          const Sophus::SE3f Trl = pKFi->GetRelativePoseTrl();       // ImuCamPose: Trl.matrix().cast<double>() (src/G2oTypes.cc:58)
          const Eigen::Matrix3f Rrl = Trl.rotationMatrix();
          for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) T[a * 4 + b] = (double)Rrl(a, b); T[a * 4 + 3] = (double)Trl.translation()(a); }
          if (pk.has_rig) {
            for (int k = 0; k < 8; ++k) if (pk.cam2[k] != c2[k]) { pk.unsupported = "keyframes with different right cameras"; return true; }
            for (int k = 0; k < 12; ++k) if (pk.trl[k] != T[k]) { pk.unsupported = "keyframes with different left-to-right transforms"; return true; }
          }
          std::copy(c2, c2 + 8, pk.cam2); std::copy(T, T + 12, pk.trl);
          pk.has_rig = true;
          const cv::KeyPoint kp = pKFi->mvKeysRight[rightIndex];
          Eigen::Matrix<double, 2, 1> obs2(kp.pt.x, kp.pt.y);
          const float unc2 = pKFi->mpCamera->uncertainty2(obs2);                  // the LEFT camera's, as the reference has it (:2819)
          const float invSigma2 = pKFi->mvInvLevelSigma2[kpUn.octave] / unc2;     // kpUn: the left keypoint variable (:2821)
          pk.edge_pose.push_back(itk->second);
          pk.edge_point.push_back(pointIndex.at(pMP));
          pk.edge_kind.push_back(OSH_EDGE_RIGHT);
          pk.edge_obs.push_back(kp.pt.x); pk.edge_obs.push_back(kp.pt.y); pk.edge_obs.push_back(-1.0);
          pk.edge_info.push_back(invSigma2);
          pk.vEdgeKF.push_back(pKFi);
          pk.vEdgeMP.push_back(pMP);
        }
      }
    }
  }
  if (pk.has_kb8)
    for (uint8_t k : pk.edge_kind) if (k == OSH_EDGE_STEREO) { pk.unsupported = "rectified-stereo observation in a KannalaBrandt8 window"; return true; }
  return true;
}

void Optimizer::LocalInertialBA(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs,
                                int& num_edges, bool bLarge, bool bRecInit) {
  (void)num_fixedKF; (void)num_OptKF; (void)num_MPs; (void)num_edges;   // never assigned by the reference
  (void)pbStopFlag;   // attached only AFTER optimize() in the reference (:2849-2850): it cannot interrupt this call
  LibaPack pk;
  const bool packed = PackLocalInertialBA(pKF, pMap, bLarge, bRecInit, pk);
  auto reset_marks = [&]() { for (KeyFrame* k : pk.lFixedKeyFrames) k->mnBAFixedForKF = 0; };
  if (!packed || pk.unsupported) {
    std::fprintf(stderr, "LocalInertialBA: %s; map left untouched\n", pk.unsupported ? pk.unsupported : "nothing to optimise");
    return;
  }
  osh_lba_ctx* ctx = HostSolverContext();
  if (!ctx) return;
  osh_liba_problem prob;
  pk.fill(prob);
  prob.huber_mono = (double)(float)std::sqrt(5.991);     // :2694
  prob.huber_stereo = (double)(float)std::sqrt(7.815);   // :2696
  prob.huber_inertial = std::sqrt(16.92);                // rki->setDelta(sqrt(16.92)) :2646
  prob.lambda_init = bLarge ? 1e-2 : 1e0;                // :2517-2528
  prob.max_iterations = pk.opt_it;
  const int N = pk.n_opt, L = (int)pk.vPointMPs.size(), E = (int)pk.edge_pose.size();
  std::vector<double> oRcw((size_t)N * 9), otcw((size_t)N * 3), oRwb((size_t)N * 9), otwb((size_t)N * 3), ov((size_t)N * 3),
      obg((size_t)N * 3), oba((size_t)N * 3), opts((size_t)L * 3), ochi(E);
  std::vector<uint8_t> odep(E);
  osh_liba_result res;
  res.pose_Rcw = oRcw.data(); res.pose_tcw = otcw.data(); res.pose_Rwb = oRwb.data(); res.pose_twb = otwb.data();
  res.vel = ov.data(); res.bias_g = obg.data(); res.bias_a = oba.data(); res.points = opts.data(); res.edge_chi2 = ochi.data(); res.edge_depth_pos = odep.data();
  if (osh_liba_solve(ctx, 1, &prob, &res) != OSH_OK) {
    std::fprintf(stderr, "LocalInertialBA: device solve failed (%s); map left untouched\n", osh_last_error());
    return;
  }
  const float err = (float)res.chi2_initial, err_end = (float)res.chi2_final;   // `float err = optimizer.activeRobustChi2()` :2845,2848
  // inlier check (:2855-2888): float thresholds; close points get 1.5x; stereo edges have no depth test
  const float chi2Mono2 = 5.991f, chi2Stereo2 = 7.815f;
  std::vector<std::pair<KeyFrame*, MapPoint*>> vToErase;
  for (int pass = 0; pass < 2; ++pass)
    for (int e = 0; e < E; ++e) {
      if ((pk.edge_kind[e] == OSH_EDGE_STEREO) != (pass == 1)) continue;   // vpEdgesMono holds EdgeMono(0) and EdgeMono(1) in insertion order
      MapPoint* pMP = pk.vEdgeMP[e];
      if (pass == 0) {
        const bool bClose = pMP->mTrackDepth < 10.f;
        if (pMP->isBad()) continue;
        if ((ochi[e] > chi2Mono2 && !bClose) || (ochi[e] > 1.5f * chi2Mono2 && bClose) || !odep[e]) vToErase.push_back(std::make_pair(pk.vEdgeKF[e], pMP));
      } else {
        if (pMP->isBad()) continue;
        if (ochi[e] > chi2Stereo2) vToErase.push_back(std::make_pair(pk.vEdgeKF[e], pMP));
      }
    }
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
  if ((2 * err < err_end || std::isnan(err) || std::isnan(err_end)) && !bLarge) {   // :2897-2901
    std::printf("FAIL LOCAL-INERTIAL BA!!!!\n");
    return;
  }
  for (auto& er : vToErase) { er.first->EraseMapPointMatch(er.second); er.second->EraseObservation(er.first); }
  reset_marks();
  // recover optimised data (:2909-2963)
  for (KeyFrame* pKFi : pk.vpOptimizableKFs) {
    int idx = 0;
    while (pk.vPoseKFs[idx] != pKFi) ++idx;
    Eigen::Matrix3f R; Eigen::Vector3f t;
    for (int a = 0; a < 9; ++a) R.v[a] = (float)oRcw[(size_t)idx * 9 + a];
    for (int a = 0; a < 3; ++a) t(a) = (float)otcw[(size_t)idx * 3 + a];
    pKFi->SetPose(Sophus::SE3f(R, t));
    pKFi->mnBALocalForKF = 0;
    if (pKFi->bImu) {
      pKFi->SetVelocity(Eigen::Vector3f((float)ov[(size_t)idx * 3], (float)ov[(size_t)idx * 3 + 1], (float)ov[(size_t)idx * 3 + 2]));
      pKFi->SetNewBias(IMU::Bias(oba[(size_t)idx * 3], oba[(size_t)idx * 3 + 1], oba[(size_t)idx * 3 + 2], obg[(size_t)idx * 3], obg[(size_t)idx * 3 + 1],
                                 obg[(size_t)idx * 3 + 2]));
    }
  }
  for (MapPoint* pMP : pk.lLocalMapPoints) {
    const int j = (int)(std::lower_bound(pk.vPointMPs.begin(), pk.vPointMPs.end(), pMP, [](MapPoint* a, MapPoint* b) { return a->mnId < b->mnId; }) - pk.vPointMPs.begin());
    pMP->SetWorldPos(Eigen::Vector3d(opts[3 * (size_t)j], opts[3 * (size_t)j + 1], opts[3 * (size_t)j + 2]).cast<float>());
    pMP->UpdateNormalAndDepth();
  }
  pMap->IncreaseChangeIndex();
}

}  // namespace ORB_SLAM3
