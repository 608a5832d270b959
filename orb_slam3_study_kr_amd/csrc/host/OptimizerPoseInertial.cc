// OptimizerPoseInertial.cc -- ORB_SLAM3::Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame on MI355X
// (src/Optimizer.cc:4499-5299).
//
// Host side: the vertex / edge construction of the reference flattened into an osh_posei_problem (visual edges in the order
// of the keypoints, kinds as the reference picks them: monocular for mvuRight < 0 or a left fisheye keypoint, rectified stereo
// otherwise, EdgeMonoOnlyPose(Xw, 1) for a right fisheye keypoint), the device runs the four Gauss-Newton rounds
// (csrc/posei_device.hip), the host writes mvbOutlier, the IMU pose / velocity / bias and the frame's ConstraintPoseImu back
// (Optimizer::Marginalize of the previous frame's block in the LastFrame variant).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "G2oTypes.h"
#include "KeyFrame.h"
#include "Optimizer.h"
#include "host_pack.h"
#include "orbslam3_hip.h"

namespace ORB_SLAM3 {

namespace {

void pack_preintegration(const IMU::Preintegrated* P, float* rec) {
  std::memset(rec, 0, sizeof(float) * OSH_PREINT_FLOATS);
  rec[0] = P->dT;
  for (int a = 0; a < 9; ++a) { rec[1 + a] = P->dR(a / 3, a % 3); rec[16 + a] = P->JRg(a / 3, a % 3); rec[25 + a] = P->JVg(a / 3, a % 3); rec[34 + a] = P->JVa(a / 3, a % 3); rec[43 + a] = P->JPg(a / 3, a % 3); rec[52 + a] = P->JPa(a / 3, a % 3); }
  for (int a = 0; a < 3; ++a) { rec[10 + a] = P->dV(a); rec[13 + a] = P->dP(a); }
  rec[61] = P->b.bax; rec[62] = P->b.bay; rec[63] = P->b.baz; rec[64] = P->b.bwx; rec[65] = P->b.bwy; rec[66] = P->b.bwz;
}

void invert3(const Eigen::Matrix<float, 15, 15>& C, int o, double* inv) {
  double m[9];
  for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) m[a * 3 + c] = (double)C(o + a, o + c);
  const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  inv[0] = (m[4] * m[8] - m[5] * m[7]) / det; inv[1] = (m[2] * m[7] - m[1] * m[8]) / det; inv[2] = (m[1] * m[5] - m[2] * m[4]) / det;
  inv[3] = (m[5] * m[6] - m[3] * m[8]) / det; inv[4] = (m[0] * m[8] - m[2] * m[6]) / det; inv[5] = (m[2] * m[3] - m[0] * m[5]) / det;
  inv[6] = (m[3] * m[7] - m[4] * m[6]) / det; inv[7] = (m[1] * m[6] - m[0] * m[7]) / det; inv[8] = (m[0] * m[4] - m[1] * m[3]) / det;
}

// symmetric eigen-decomposition (cyclic Jacobi): A -> eigenvalues w, eigenvectors in the columns of V
void sym_eig(int n, double* A, double* w, double* V) {
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        if (A[p * n + q] == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2 * A[p * n + q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
        const double c = 1 / std::sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < n; ++k) { const double akp = A[k * n + p], akq = A[k * n + q]; A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq; }
        for (int k = 0; k < n; ++k) { const double apk = A[p * n + k], aqk = A[q * n + k]; A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk; }
        for (int k = 0; k < n; ++k) { const double vkp = V[k * n + p], vkq = V[k * n + q]; V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq; }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}

// Optimizer::Marginalize(H, 0, 14).block<15,15>(15,15) of the 30x30 Hessian (src/Optimizer.cc:2967-3050, :5293-5294): the Schur
// complement of the previous frame's block with its pseudo-inverse (JacobiSVD there, singular values above 1e-6 kept; the block
// is symmetric, so its singular values are the absolute eigenvalues)
void marginalize_previous(const double* H30, double* out15) {
  double Hb[225], w[15], V[225], inv[225];
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) Hb[i * 15 + j] = 0.5 * (H30[i * 30 + j] + H30[j * 30 + i]);
  sym_eig(15, Hb, w, V);
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) {
    double t = 0;
    for (int k = 0; k < 15; ++k) if (std::fabs(w[k]) > 1e-6) t += V[i * 15 + k] * V[j * 15 + k] / w[k];
    inv[i * 15 + j] = t;
  }
  for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) {
    double acc = H30[(15 + i) * 30 + 15 + j];
    for (int k = 0; k < 15; ++k) { double t = 0; for (int m = 0; m < 15; ++m) t += inv[k * 15 + m] * H30[m * 30 + 15 + j]; acc -= H30[(15 + i) * 30 + k] * t; }
    out15[i * 15 + j] = acc;
  }
}

}  // namespace

// The problem of one call (exposed so that the test harness can read back the arrays the device was given)
bool PackPoseInertial(Frame* pFrame, bool bRecInit, int mode, PoseiPack& pk) {
  pk = PoseiPack();
  pk.mode = mode; pk.rec_init = bRecInit;
  const int N = pFrame->N, Nleft = pFrame->Nleft;
  const bool bRight = (Nleft != -1);
  pk.index.reserve(N);
  {
    std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);
    for (int i = 0; i < N; i++) {
      MapPoint* pMP = pFrame->mvpMapPoints[i];
      if (!pMP) continue;
      cv::KeyPoint kpUn;
      int kind;
      double ur = -1.0;
      if ((!bRight && pFrame->mvuRight[i] < 0) || i < Nleft) {          // left monocular observation (:4565-4597)
        kpUn = (i < Nleft) ? pFrame->mvKeys[i] : pFrame->mvKeysUn[i];
        kind = OSH_EDGE_MONO;
        pk.n_mono++;
      } else if (!bRight) {                                              // stereo observation (:4599-4628)
        kpUn = pFrame->mvKeysUn[i];
        ur = pFrame->mvuRight[i];
        kind = OSH_EDGE_STEREO;
        pk.n_stereo++;
      } else {                                                           // right monocular observation (:4631-4661): i >= Nleft
        kpUn = pFrame->mvKeysRight[i - Nleft];
        kind = OSH_EDGE_RIGHT;
        pk.n_mono++;
      }
      pFrame->mvbOutlier[i] = false;
      GeometricCamera* c = pFrame->mpCamera;
      const bool fisheye = c && c->GetType() == GeometricCamera::CAM_FISHEYE;
      if (kind != OSH_EDGE_STEREO) {
        if (!c || (!fisheye && c->GetType() != GeometricCamera::CAM_PINHOLE) || c->getParameter(0) != pFrame->fx || c->getParameter(1) != pFrame->fy ||
            c->getParameter(2) != pFrame->cx || c->getParameter(3) != pFrame->cy) { pk.unsupported = "monocular observation through a camera that is not the frame's own model"; return false; }
        if (fisheye) { pk.has_kb8 = true; for (int k = 0; k < 4; ++k) pk.kb8[k] = c->getParameter(4 + k); }
      }
      Eigen::Matrix<double, 2, 1> obs2(kpUn.pt.x, kpUn.pt.y);
      const float unc2 = pFrame->mpCamera->uncertainty2(obs2);
      const float invSigma2 = pFrame->mvInvLevelSigma2[kpUn.octave] / unc2;
      const Eigen::Vector3d Xw = pMP->GetWorldPos().cast<double>();
      pk.points.push_back(Xw[0]); pk.points.push_back(Xw[1]); pk.points.push_back(Xw[2]);
      pk.edge_obs.push_back(kpUn.pt.x); pk.edge_obs.push_back(kpUn.pt.y); pk.edge_obs.push_back(ur);
      pk.edge_info.push_back(invSigma2);
      pk.edge_kind.push_back((uint8_t)kind);
      pk.edge_close.push_back(pMP->mTrackDepth < 10.f ? 1 : 0);
      pk.index.push_back(i);
    }
  }
  if (pk.has_kb8 && pk.n_stereo > 0) { pk.unsupported = "rectified-stereo observations in a KannalaBrandt8 frame"; return false; }
  if (bRight) {
    if (!pFrame->mpCamera2 || pFrame->mpCamera2->GetType() != GeometricCamera::CAM_FISHEYE || !pk.has_kb8) { pk.unsupported = "a two-camera frame that is not a KannalaBrandt8 pair"; return false; }
    for (int k = 0; k < 8; ++k) pk.cam2[k] = pFrame->mpCamera2->getParameter(k);
    const Sophus::SE3f Trl = pFrame->GetRelativePoseTrl();               // ImuCamPose(Frame*): Trl.matrix().cast<double>() (src/G2oTypes.cc:104)
    const Eigen::Matrix3f Rrl = Trl.rotationMatrix();
    for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) pk.trl[a * 4 + b] = (double)Rrl(a, b); pk.trl[a * 4 + 3] = (double)Trl.translation()(a); }
    pk.has_rig = true;
  }
  // current frame: VertexPose / Velocity / GyroBias / AccBias (pFrame) (:4520-4539)
  {
    const Eigen::Matrix3f Rcw = pFrame->GetPose().rotationMatrix(), Rwb = pFrame->GetImuRotation();
    const Eigen::Vector3f tcw = pFrame->GetPose().translation(), twb = pFrame->GetImuPosition(), v = pFrame->GetVelocity();
    for (int a = 0; a < 9; ++a) { pk.Rcw[a] = (double)Rcw(a / 3, a % 3); pk.Rwb[a] = (double)Rwb(a / 3, a % 3); }
    for (int a = 0; a < 3; ++a) { pk.tcw[a] = (double)tcw(a); pk.twb[a] = (double)twb(a); pk.vel[a] = (double)v(a); }
    pk.bias_g[0] = pFrame->mImuBias.bwx; pk.bias_g[1] = pFrame->mImuBias.bwy; pk.bias_g[2] = pFrame->mImuBias.bwz;
    pk.bias_a[0] = pFrame->mImuBias.bax; pk.bias_a[1] = pFrame->mImuBias.bay; pk.bias_a[2] = pFrame->mImuBias.baz;
    const IMU::Calib& cal = pFrame->mImuCalib;
    const Eigen::Matrix3f Rcb = cal.mTcb.rotationMatrix();
    for (int a = 0; a < 9; ++a) pk.Rcb[a] = (double)Rcb(a / 3, a % 3);
    for (int a = 0; a < 3; ++a) { pk.tcb[a] = (double)cal.mTcb.translation()(a); pk.tbc[a] = (double)cal.mTbc.translation()(a); }
    pk.cam[0] = pFrame->fx; pk.cam[1] = pFrame->fy; pk.cam[2] = pFrame->cx; pk.cam[3] = pFrame->cy; pk.cam[4] = pFrame->mbf;
  }
  const IMU::Preintegrated* Plink = nullptr;
  if (mode == 0) {                                                        // the last keyframe, fixed (:4666-4686)
    KeyFrame* pKF = pFrame->mpLastKeyFrame;
    if (!pKF || !pFrame->mpImuPreintegrated) { pk.unsupported = "no last keyframe / preintegration"; return false; }
    const Eigen::Matrix3f Rwb = pKF->GetImuRotation();
    const Eigen::Vector3f twb = pKF->GetImuPosition(), v = pKF->GetVelocity(), bg = pKF->GetGyroBias(), ba = pKF->GetAccBias();
    for (int a = 0; a < 9; ++a) pk.prev_Rwb[a] = (double)Rwb(a / 3, a % 3);
    for (int a = 0; a < 3; ++a) { pk.prev_twb[a] = (double)twb(a); pk.prev_vel[a] = (double)v(a); pk.prev_bias_g[a] = (double)bg(a); pk.prev_bias_a[a] = (double)ba(a); }
    Plink = pFrame->mpImuPreintegrated;
  } else {                                                                // the previous frame, free, with its prior (:5068-5118)
    Frame* pFp = pFrame->mpPrevFrame;
    if (!pFp || !pFrame->mpImuPreintegratedFrame || !pFrame->mpImuPreintegrated) { pk.unsupported = "no previous frame / preintegration"; return false; }
    if (!pFp->mpcpi) { pk.unsupported = "pFp->mpcpi does not exist"; return false; }
    const Eigen::Matrix3f Rwb = pFp->GetImuRotation();
    const Eigen::Vector3f twb = pFp->GetImuPosition(), v = pFp->GetVelocity();
    for (int a = 0; a < 9; ++a) pk.prev_Rwb[a] = (double)Rwb(a / 3, a % 3);
    for (int a = 0; a < 3; ++a) { pk.prev_twb[a] = (double)twb(a); pk.prev_vel[a] = (double)v(a); }
    pk.prev_bias_g[0] = pFp->mImuBias.bwx; pk.prev_bias_g[1] = pFp->mImuBias.bwy; pk.prev_bias_g[2] = pFp->mImuBias.bwz;
    pk.prev_bias_a[0] = pFp->mImuBias.bax; pk.prev_bias_a[1] = pFp->mImuBias.bay; pk.prev_bias_a[2] = pFp->mImuBias.baz;
    const ConstraintPoseImu* c = pFp->mpcpi;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) pk.prior_Rwb[a * 3 + b] = c->Rwb(a, b);
    for (int a = 0; a < 3; ++a) { pk.prior_twb[a] = c->twb(a); pk.prior_vel[a] = c->vwb(a); pk.prior_bg[a] = c->bg(a); pk.prior_ba[a] = c->ba(a); }
    for (int a = 0; a < 15; ++a) for (int b = 0; b < 15; ++b) pk.prior_H[a * 15 + b] = c->H(a, b);
    Plink = pFrame->mpImuPreintegratedFrame;
  }
  pack_preintegration(Plink, pk.preint);
  InertialInformation(Plink->C, pk.info_inertial);
  // the random-walk informations come from mpImuPreintegrated in BOTH variants (:4702, 4710 / :5092, 5100)
  invert3(pFrame->mpImuPreintegrated->C, 9, pk.info_g);
  invert3(pFrame->mpImuPreintegrated->C, 12, pk.info_a);
  return true;
}

static int PoseInertialOptimization(Frame* pFrame, bool bRecInit, int mode, const char* name) {
  PoseiPack pk;
  if (!PackPoseInertial(pFrame, bRecInit, mode, pk)) {
    std::fprintf(stderr, "%s: %s; frame left untouched\n", name, pk.unsupported ? pk.unsupported : "cannot build the problem");
    return 0;
  }
  osh_lba_ctx* ctx = HostSolverContext();
  if (!ctx) return 0;
  osh_posei_problem prob;
  pk.fill(prob);
  std::vector<uint8_t> outlier(pk.index.size());
  osh_posei_result res;
  res.outlier = outlier.data(); res.edge_chi2 = nullptr;
  if (osh_posei_optimize(ctx, 1, &prob, &res) != OSH_OK) {
    std::fprintf(stderr, "%s: device solve failed (%s); frame left untouched\n", name, osh_last_error());
    return 0;
  }
  for (size_t e = 0; e < pk.index.size(); ++e) pFrame->mvbOutlier[pk.index[e]] = outlier[e] != 0;
  // recover optimised pose, velocity and biases (:4851-4855)
  Eigen::Matrix3d Rwb; Eigen::Vector3d twb, vwb, bg, ba;
  for (int a = 0; a < 9; ++a) Rwb(a / 3, a % 3) = res.Rwb[a];
  for (int a = 0; a < 3; ++a) { twb(a) = res.twb[a]; vwb(a) = res.vel[a]; bg(a) = res.bias_g[a]; ba(a) = res.bias_a[a]; }
  pFrame->SetImuPoseVelocity(Rwb.cast<float>(), twb.cast<float>(), vwb.cast<float>());
  pFrame->mImuBias = IMU::Bias((float)ba(0), (float)ba(1), (float)ba(2), (float)bg(0), (float)bg(1), (float)bg(2));
  // the frame's ConstraintPoseImu (:4857-4895 / :5252-5297)
  Matrix15d H;
  if (mode == 0) {
    for (int a = 0; a < 15; ++a) for (int b = 0; b < 15; ++b) H(a, b) = res.H[a * 15 + b];
  } else {
    double Hm[225];
    marginalize_previous(res.H, Hm);
    for (int a = 0; a < 15; ++a) for (int b = 0; b < 15; ++b) H(a, b) = Hm[a * 15 + b];
  }
  pFrame->mpcpi = new ConstraintPoseImu(Rwb, twb, vwb, bg, ba, H);
  if (mode == 1) { delete pFrame->mpPrevFrame->mpcpi; pFrame->mpPrevFrame->mpcpi = nullptr; }   // :5296-5297
  return (pk.n_mono + pk.n_stereo) - res.n_bad;
}

int Optimizer::PoseInertialOptimizationLastKeyFrame(Frame* pFrame, bool bRecInit) {
  return PoseInertialOptimization(pFrame, bRecInit, 0, "PoseInertialOptimizationLastKeyFrame");
}

int Optimizer::PoseInertialOptimizationLastFrame(Frame* pFrame, bool bRecInit) {
  return PoseInertialOptimization(pFrame, bRecInit, 1, "PoseInertialOptimizationLastFrame");
}

}  // namespace ORB_SLAM3
