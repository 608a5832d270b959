// OptimizerPose.cc -- ORB_SLAM3::Optimizer::PoseOptimization on MI355X.
//
// Host side: the edge construction of src/Optimizer.cc:815-1017 flattened into an osh_pose_problem, the device runs the
// four optimise / classify rounds (:1019-1108, csrc/pose_device.hip), the host writes mvbOutlier and the pose back
// (:1110-1114).  Monocular (pinhole or KannalaBrandt8), rectified-stereo / RGB-D and fisheye-stereo frames (pFrame->mpCamera2,
// right keypoints as EdgeSE3ProjectXYZOnlyPoseToBody, :933-1008).
#include <cmath>
#include <cstdio>
#include <mutex>
#include <vector>

#include "Optimizer.h"
#include "host_pack.h"
#include "orbslam3_hip.h"

namespace ORB_SLAM3 {

int Optimizer::PoseOptimization(Frame* pFrame) {
  int nInitialCorrespondences = 0;
  const Sophus::SE3f Tcw = pFrame->GetPose();
  const Eigen::Quaterniond q = Tcw.unit_quaternion().cast<double>();   // :833-834 float -> double
  const Eigen::Vector3d t = Tcw.translation().cast<double>();
  const double pose_qt[7] = {q.x(), q.y(), q.z(), q.w(), t[0], t[1], t[2]};
  const double cam[5] = {pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy, pFrame->mbf};   // :928-932
  const int N = pFrame->N;
  std::vector<double> points, obs, info;
  std::vector<uint8_t> kind;
  bool has_kb8 = false, has_rig = false;
  double kb8[4] = {0, 0, 0, 0}, cam2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, trl[7] = {0, 0, 0, 1, 0, 0, 0};
  int n_stereo = 0;
  std::vector<int> index;   // keypoint of every edge (vnIndexEdgeMono / vnIndexEdgeStereo merged, edge order = keypoint order)
  points.reserve((size_t)N * 3); obs.reserve((size_t)N * 3); info.reserve(N); kind.reserve(N); index.reserve(N);
  {
    std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);
    for (int i = 0; i < N; i++) {
      MapPoint* pMP = pFrame->mvpMapPoints[i];
      if (!pMP) continue;
      nInitialCorrespondences++;
      pFrame->mvbOutlier[i] = false;
      if (pFrame->mpCamera2) {
        // SLAM with respect to a rigid body (:933-1008): left keypoints [0, Nleft) through mpCamera on the raw keypoints (mvKeys),
        // right keypoints through Trl and mpCamera2
        const bool right = i >= pFrame->Nleft;
        GeometricCamera* c = pFrame->mpCamera;
        if (!c || c->GetType() != GeometricCamera::CAM_FISHEYE || pFrame->mpCamera2->GetType() != GeometricCamera::CAM_FISHEYE ||
            c->getParameter(0) != pFrame->fx || c->getParameter(1) != pFrame->fy || c->getParameter(2) != pFrame->cx || c->getParameter(3) != pFrame->cy) {
          std::fprintf(stderr, "PoseOptimization: a two-camera frame that is not a KannalaBrandt8 pair; not supported\n");
          return 0;
        }
        has_kb8 = true; has_rig = true;
        for (int k = 0; k < 4; ++k) kb8[k] = c->getParameter(4 + k);
        const cv::KeyPoint& kp = right ? pFrame->mvKeysRight[i - pFrame->Nleft] : pFrame->mvKeys[i];
        const Eigen::Vector3d Xr = pMP->GetWorldPos().cast<double>();
        points.push_back(Xr[0]); points.push_back(Xr[1]); points.push_back(Xr[2]);
        obs.push_back(kp.pt.x); obs.push_back(kp.pt.y); obs.push_back(-1.0);
        info.push_back(pFrame->mvInvLevelSigma2[kp.octave]);
        kind.push_back(right ? OSH_EDGE_BODY : OSH_EDGE_MONO);
        index.push_back(i);
        continue;
      }
      const cv::KeyPoint& kpUn = pFrame->mvKeysUn[i];
      const float kp_ur = pFrame->mvuRight[i];
      const bool stereo = !(kp_ur < 0);   // mono if mvuRight < 0 (:871), stereo otherwise
      if (!stereo) {
        // the mono edge projects through pFrame->mpCamera (:897): the frame's pinhole model or its KannalaBrandt8 (fisheye) model
        GeometricCamera* c = pFrame->mpCamera;
        const bool fisheye = c && c->GetType() == GeometricCamera::CAM_FISHEYE;
        if (!c || (!fisheye && c->GetType() != GeometricCamera::CAM_PINHOLE) || c->getParameter(0) != pFrame->fx ||
            c->getParameter(1) != pFrame->fy || c->getParameter(2) != pFrame->cx || c->getParameter(3) != pFrame->cy) {
          std::fprintf(stderr, "PoseOptimization: monocular observation through a camera that is not the frame's own model; not supported yet\n");
          return 0;
        }
        if (fisheye) { has_kb8 = true; for (int k = 0; k < 4; ++k) kb8[k] = c->getParameter(4 + k); }
      } else n_stereo++;
      const Eigen::Vector3d Xw = pMP->GetWorldPos().cast<double>();
      points.push_back(Xw[0]); points.push_back(Xw[1]); points.push_back(Xw[2]);
      obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(stereo ? kp_ur : -1.0);
      info.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
      kind.push_back(stereo ? OSH_EDGE_STEREO : OSH_EDGE_MONO);
      index.push_back(i);
    }
  }
  if (nInitialCorrespondences < 3) return 0;   // :1012-1013
  if (has_kb8 && n_stereo > 0) {
    std::fprintf(stderr, "PoseOptimization: rectified-stereo observations in a KannalaBrandt8 frame are not supported by the MI355X path yet\n");
    return 0;
  }

  osh_lba_ctx* ctx = HostSolverContext();
  if (!ctx) return 0;
  osh_pose_problem prob;
  prob.n_edges = (int32_t)index.size();
  prob.pose_qt = pose_qt; prob.cam = cam; prob.points = points.data(); prob.edge_kind = kind.data();
  prob.edge_obs = obs.data(); prob.edge_info = info.data();
  if (has_rig) {
    for (int k = 0; k < 8; ++k) cam2[k] = pFrame->mpCamera2->getParameter(k);
    const Sophus::SE3f Trl = pFrame->GetRelativePoseTrl();   // e->mTrl = g2o::SE3Quat(Trl.unit_quaternion().cast<double>(), Trl.translation().cast<double>()) (:997)
    const Eigen::Quaterniond ql = Trl.unit_quaternion().cast<double>();
    const Eigen::Vector3d tl = Trl.translation().cast<double>();
    trl[0] = ql.x(); trl[1] = ql.y(); trl[2] = ql.z(); trl[3] = ql.w(); trl[4] = tl[0]; trl[5] = tl[1]; trl[6] = tl[2];
  }
  prob.kb8 = has_kb8 ? kb8 : nullptr;
  prob.cam2 = has_rig ? cam2 : nullptr; prob.trl = has_rig ? trl : nullptr;
  prob.huber_mono = (double)(float)std::sqrt(5.991);     // const float deltaMono = sqrt(5.991) (:858)
  prob.huber_stereo = (double)(float)std::sqrt(7.815);   // (:859)
  for (int k = 0; k < 4; ++k) { prob.chi2_mono[k] = 5.991f; prob.chi2_stereo[k] = 7.815f; prob.iterations[k] = 10; }   // :1016-1018
  std::vector<uint8_t> outlier(index.size());
  osh_pose_result res;
  res.outlier = outlier.data(); res.edge_chi2 = nullptr;
  if (osh_pose_optimize(ctx, 1, &prob, &res) != OSH_OK) {
    std::fprintf(stderr, "PoseOptimization: device solve failed (%s); frame left untouched\n", osh_last_error());
    return 0;
  }
  for (size_t e = 0; e < index.size(); ++e) pFrame->mvbOutlier[index[e]] = outlier[e] != 0;
  const double* qt = res.pose_qt;
  const Sophus::SE3f pose(Eigen::Quaterniond(qt[3], qt[0], qt[1], qt[2]).cast<float>(), Eigen::Vector3d(qt[4], qt[5], qt[6]).cast<float>());
  pFrame->SetPose(pose);   // :1111-1112
  return nInitialCorrespondences - res.n_bad;
}

}  // namespace ORB_SLAM3
