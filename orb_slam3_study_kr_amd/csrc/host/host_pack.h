// host_pack.h -- the flattened local-BA window produced by PackLocalBA (csrc/host/Optimizer.cc).
#pragma once
#include <cmath>
#include <cstdint>
#include <list>
#include <vector>
#include "CameraModels/GeometricCamera.h"
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
  bool has_kb8 = false;       // the keyframes' camera is a KannalaBrandt8 (monocular fisheye)
  double kb8[4] = {0, 0, 0, 0};
  int n_pinhole_mono = 0;
  bool has_rig = false;       // fisheye stereo rig: right-camera (body) edges through cam2 / trl
  double cam2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double trl[7] = {0, 0, 0, 1, 0, 0, 0};
  // Right camera of a keyframe with a right-camera observation (e->pCamera = pKFi->mpCamera2, e->mTrl = GetRelativePoseTrl(),
  // src/Optimizer.cc:1389-1392): one KannalaBrandt8 model and one Trl shared by the whole window (a rig is rigid).
  bool rig_camera(KeyFrame* pKF) {
    GeometricCamera* cam = pKF->mpCamera2;
    if (!cam || cam->GetType() != GeometricCamera::CAM_FISHEYE) { unsupported = "right camera that is not a KannalaBrandt8"; return false; }
    double c[8], t[7];
    for (int i = 0; i < 8; ++i) c[i] = cam->getParameter(i);
    const Sophus::SE3f Trl = pKF->GetRelativePoseTrl();
    const Eigen::Quaterniond q = Trl.unit_quaternion().cast<double>();
    const Eigen::Vector3d tt = Trl.translation().cast<double>();
    t[0] = q.x(); t[1] = q.y(); t[2] = q.z(); t[3] = q.w(); t[4] = tt[0]; t[5] = tt[1]; t[6] = tt[2];
    if (has_rig) {
      for (int i = 0; i < 8; ++i) if (c[i] != cam2[i]) { unsupported = "keyframes with different right cameras in one window"; return false; }
      for (int i = 0; i < 7; ++i) if (t[i] != trl[i]) { unsupported = "keyframes with different Trl in one window"; return false; }
    }
    has_rig = true;
    for (int i = 0; i < 8; ++i) cam2[i] = c[i];
    for (int i = 0; i < 7; ++i) trl[i] = t[i];
    return true;
  }
  // the right-camera edge of (pKF, pMP): appended right after the pair's left edge, as the reference inserts it
  bool add_body_edge(KeyFrame* pKF, MapPoint* pMP, int rightIndex, int pose, int point) {
    if (!rig_camera(pKF)) return false;
    rightIndex -= pKF->NLeft;                                   // :1369
    if (rightIndex < 0 || rightIndex >= (int)pKF->mvKeysRight.size()) { unsupported = "right-camera index outside mvKeysRight"; return false; }
    const cv::KeyPoint& kp = pKF->mvKeysRight[rightIndex];      // :1372
    edge_pose.push_back(pose);
    edge_point.push_back(point);
    edge_kind.push_back(OSH_EDGE_BODY);
    edge_obs.push_back(kp.pt.x); edge_obs.push_back(kp.pt.y); edge_obs.push_back(-1.0);
    edge_info.push_back(pKF->mvInvLevelSigma2[kp.octave]);      // :1380-1381
    vEdgeKF.push_back(pKF);
    vEdgeMP.push_back(pMP);
    return true;
  }
  // Camera of a monocular observation (the edge projects through pKF->mpCamera, src/Optimizer.cc:1323): the keyframe's own
  // pinhole model, or one KannalaBrandt8 model shared by the whole window.  Anything else sets `unsupported`.
  bool mono_camera(GeometricCamera* cam, float fx, float fy, float cx, float cy) {
    if (!cam || cam->getParameter(0) != fx || cam->getParameter(1) != fy || cam->getParameter(2) != cx || cam->getParameter(3) != cy) {
      unsupported = "monocular observation through a camera that is not the keyframe's own model";
      return false;
    }
    if (cam->GetType() == GeometricCamera::CAM_PINHOLE) { ++n_pinhole_mono; }
    else if (cam->GetType() == GeometricCamera::CAM_FISHEYE) {
      double k[4];
      for (int i = 0; i < 4; ++i) k[i] = cam->getParameter(4 + i);
      if (has_kb8 && (k[0] != kb8[0] || k[1] != kb8[1] || k[2] != kb8[2] || k[3] != kb8[3])) {
        unsupported = "keyframes with different KannalaBrandt8 coefficients in one window";
        return false;
      }
      has_kb8 = true;
      for (int i = 0; i < 4; ++i) kb8[i] = k[i];
    } else { unsupported = "unknown camera model"; return false; }
    if (has_kb8 && n_pinhole_mono > 0) { unsupported = "pinhole and KannalaBrandt8 monocular observations in one window"; return false; }
    return true;
  }
  // a fisheye window is monocular on the device (no rectified-stereo edges next to KannalaBrandt8 ones)
  bool camera_models_ok() {
    if (has_kb8)
      for (uint8_t k : edge_kind) if (k == OSH_EDGE_STEREO) { unsupported = "rectified-stereo observation in a KannalaBrandt8 window"; return false; }
    if (has_rig && !has_kb8) { unsupported = "right-camera observations without a KannalaBrandt8 left camera"; return false; }
    return true;
  }
  void fill(osh_lba_problem& p) const {
    p.n_free = n_free; p.n_fixed = n_fixed; p.n_points = (int32_t)vPointMPs.size(); p.n_edges = (int32_t)edge_pose.size();
    p.pose_qt = pose_qt.data(); p.pose_cam = pose_cam.data(); p.points = points.data();
    p.edge_pose = edge_pose.data(); p.edge_point = edge_point.data(); p.edge_kind = edge_kind.data();
    p.edge_obs = edge_obs.data(); p.edge_info = edge_info.data();
    p.huber_mono = p.huber_stereo = 0; p.lambda_init = 0; p.max_iterations = 10; p.stop_flag = nullptr;
    p.kb8 = has_kb8 ? kb8 : nullptr;
    p.cam2 = has_rig ? cam2 : nullptr; p.trl = has_rig ? trl : nullptr;
  }
};
struct LibaPack {
  std::vector<KeyFrame*> vpOptimizableKFs;              // newest first, as the reference builds it (:2402-2417)
  std::list<KeyFrame*> lFixedKeyFrames;
  std::list<MapPoint*> lLocalMapPoints;
  std::vector<KeyFrame*> vPoseKFs;                       // problem order: temporal (ascending id), fixed predecessor, fixed observers
  std::vector<MapPoint*> vPointMPs;
  std::vector<KeyFrame*> vEdgeKF;
  std::vector<MapPoint*> vEdgeMP;
  int n_opt = 0, n_fixed_imu = 0, n_fixed = 0, opt_it = 10;
  const char* unsupported = nullptr;
  std::vector<double> pose_Rcw, pose_tcw, pose_Rwb, pose_twb, vel, bias_g, bias_a, points, edge_obs, edge_info, link_info, link_info_g, link_info_a;
  double Rcb[9], tcb[3], tbc[3], cam[5];
  bool has_kb8 = false;       // the window's camera is a KannalaBrandt8 (monocular fisheye)
  double kb8[4] = {0, 0, 0, 0};
  bool has_rig = false;       // fisheye stereo rig: OSH_EDGE_RIGHT edges (EdgeMono(1)) through cam2 / trl
  double cam2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, trl[12] = {0};
  std::vector<int32_t> edge_pose, edge_point, link_prev, link_cur;
  std::vector<int32_t> link_bias;   // empty, or per link the keyframe that stores the edge's bias vertices (FullInertialBA with bInit)
  std::vector<uint8_t> edge_kind, link_robust;
  std::vector<float> link_preint;
  void fill(osh_liba_problem& p) const {
    p.n_opt = n_opt; p.n_fixed_imu = n_fixed_imu; p.n_fixed = n_fixed;
    p.n_points = (int32_t)vPointMPs.size(); p.n_edges = (int32_t)edge_pose.size(); p.n_links = (int32_t)link_prev.size();
    p.pose_Rcw = pose_Rcw.data(); p.pose_tcw = pose_tcw.data(); p.pose_Rwb = pose_Rwb.data(); p.pose_twb = pose_twb.data();
    p.Rcb = Rcb; p.tcb = tcb; p.tbc = tbc; p.cam = cam; p.vel = vel.data(); p.bias_g = bias_g.data(); p.bias_a = bias_a.data();
    p.points = points.data(); p.edge_pose = edge_pose.data(); p.edge_point = edge_point.data(); p.edge_kind = edge_kind.data();
    p.edge_obs = edge_obs.data(); p.edge_info = edge_info.data(); p.link_prev = link_prev.data(); p.link_cur = link_cur.data();
    p.link_preint = link_preint.data(); p.link_info = link_info.data(); p.link_info_g = link_info_g.data(); p.link_info_a = link_info_a.data();
    p.link_robust = link_robust.data();
    p.huber_mono = p.huber_stereo = p.huber_inertial = 0; p.lambda_init = 1.0; p.max_iterations = opt_it;
    p.kb8 = has_kb8 ? kb8 : nullptr;
    p.cam2 = has_rig ? cam2 : nullptr; p.trl = has_rig ? trl : nullptr;
    p.link_bias = link_bias.empty() ? nullptr : link_bias.data();
  }
};
// Flat problem of Optimizer::PoseInertialOptimizationLastKeyFrame (mode 0) / LastFrame (mode 1), src/Optimizer.cc:4499-5299
struct PoseiPack {
  int mode = 0;
  bool rec_init = false;
  const char* unsupported = nullptr;
  int n_mono = 0, n_stereo = 0;                 // nInitialMonoCorrespondences / nInitialStereoCorrespondences
  std::vector<int> index;                       // keypoint of every edge
  std::vector<double> points, edge_obs, edge_info;
  std::vector<uint8_t> edge_kind, edge_close;
  double Rcw[9], tcw[3], Rwb[9], twb[3], vel[3], bias_g[3], bias_a[3];
  double prev_Rwb[9], prev_twb[3], prev_vel[3], prev_bias_g[3], prev_bias_a[3];
  double Rcb[9], tcb[3], tbc[3], cam[5];
  bool has_kb8 = false, has_rig = false;
  double kb8[4] = {0, 0, 0, 0}, cam2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, trl[12] = {0};
  float preint[OSH_PREINT_FLOATS];
  double info_inertial[81], info_g[9], info_a[9];
  double prior_Rwb[9], prior_twb[3], prior_vel[3], prior_bg[3], prior_ba[3], prior_H[225];
  void fill(osh_posei_problem& p) const {
    p.mode = mode; p.n_edges = (int32_t)index.size(); p.rec_init = rec_init ? 1 : 0;
    p.Rcw = Rcw; p.tcw = tcw; p.Rwb = Rwb; p.twb = twb; p.vel = vel; p.bias_g = bias_g; p.bias_a = bias_a;
    p.prev_Rwb = prev_Rwb; p.prev_twb = prev_twb; p.prev_vel = prev_vel; p.prev_bias_g = prev_bias_g; p.prev_bias_a = prev_bias_a;
    p.Rcb = Rcb; p.tcb = tcb; p.tbc = tbc; p.cam = cam;
    p.kb8 = has_kb8 ? kb8 : nullptr; p.cam2 = has_rig ? cam2 : nullptr; p.trl = has_rig ? trl : nullptr;
    p.preint = preint; p.info_inertial = info_inertial; p.info_g = info_g; p.info_a = info_a;
    const bool pr = mode == 1;
    p.prior_Rwb = pr ? prior_Rwb : nullptr; p.prior_twb = pr ? prior_twb : nullptr; p.prior_vel = pr ? prior_vel : nullptr;
    p.prior_bg = pr ? prior_bg : nullptr; p.prior_ba = pr ? prior_ba : nullptr; p.prior_H = pr ? prior_H : nullptr;
    p.points = points.data(); p.edge_kind = edge_kind.data(); p.edge_obs = edge_obs.data(); p.edge_info = edge_info.data();
    p.edge_close = edge_close.data();
    p.huber_mono = (double)(float)std::sqrt(5.991);     // const float thHuberMono = sqrt(5.991) (:4552)
    p.huber_stereo = (double)(float)std::sqrt(7.815);
    p.huber_prior = 5.0;                                 // rkp->setDelta(5) (:5117)
    const float m0[4] = {12.f, 7.5f, 5.991f, 5.991f}, m1[4] = {5.991f, 5.991f, 5.991f, 5.991f}, st[4] = {15.6f, 9.8f, 7.815f, 7.815f};
    for (int k = 0; k < 4; ++k) { p.chi2_mono[k] = mode == 0 ? m0[k] : m1[k]; p.chi2_stereo[k] = st[k]; p.iterations[k] = 10; }   // :4714-4716 / :5121-5123
  }
};
class Frame;
bool PackPoseInertial(Frame* pFrame, bool bRecInit, int mode, PoseiPack& pk);
bool PackLocalInertialBA(KeyFrame* pKF, Map* pMap, bool bLarge, bool bRecInit, LibaPack& pk);
// Optimizer::FullInertialBA / MergeInertialBA as the same flat problem (OptimizerInertialMap.cc).  vpIdle: keyframes no edge touches;
// vpCovKFs: the merge's covisible keyframes in the reference's order
// *sharedBiasSlot: with bInit the keyframe (pose index) whose bias slot holds the one optimised bias pair, else -1
bool PackFullInertialBA(Map* pMap, int its, bool bFixLocal, bool bInit, float priorG, float priorA, LibaPack& pk, std::vector<KeyFrame*>& vpIdle,
                        std::vector<MapPoint*>& vpAllMPs, int* sharedBiasSlot);
bool PackMergeInertialBA(KeyFrame* pCurrKF, KeyFrame* pMergeKF, LibaPack& pk, std::vector<KeyFrame*>& vpCovKFs);
void InertialInformation(const Eigen::Matrix<float, 15, 15>& C, double* info81);
osh_lba_ctx* HostSolverContext();   // one solver context per calling thread (Optimizer.cc)

// Steps 1-6 of Optimizer::LocalBundleAdjustment; false when the window has no fixed keyframe.
bool PackLocalBA(KeyFrame* pKF, Map* pMap, LbaPack& pk);
// Vertex / edge construction of the welding Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, ...) (src/Optimizer.cc:3524-3705);
// vpMPs = the map points of the problem in the reference's insertion order.
void PackWeldingBA(KeyFrame* pMainKF, const std::vector<KeyFrame*>& vpAdjustKF, const std::vector<KeyFrame*>& vpFixedKF, LbaPack& pk,
                   std::vector<MapPoint*>& vpMPs);
// Vertex / edge construction of Optimizer::BundleAdjustment (src/Optimizer.cc:112-300); vbNotIncludedMP as there.
void PackBundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, LbaPack& pk,
                          std::vector<bool>& vbNotIncludedMP);
}  // namespace ORB_SLAM3
