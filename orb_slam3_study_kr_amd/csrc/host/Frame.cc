// Frame.cc -- Frame::isInFrustum on the device (reference src/Frame.cc:513-587, Nleft == -1 branch) and
// Frame::UpdatePoseMatrices (src/Frame.cc:473-480).
//
// Tracking::SearchLocalPoints (src/Tracking.cc:3411-3432) calls isInFrustum once per local map point; the batch overload
// below projects the whole list in one launch of k_frustum (csrc/orb_device.hip) and writes the same MapPoint fields
// (mbTrackInView, mTrackProjX/Y/XR, mTrackDepth, mnTrackScaleLevel, mTrackViewCos).  The single-point overload keeps the
// reference signature and goes through the same kernel with a batch of one.  There is no CPU fallback.
#include <cstdio>
#include <vector>
#include "Frame.h"
#include "MapPoint.h"
#include "orbslam3_hip.h"

namespace ORB_SLAM3 {

osh_orb_ctx* HostMatcherContext();   // csrc/host/ORBmatcher.cc

void Frame::UpdatePoseMatrices() {
  const Sophus::SE3<float> Twc = mTcw.inverse();
  mRwc = Twc.rotationMatrix();
  mOw = Twc.translation();
  mRcw = mTcw.rotationMatrix();
  mtcw = mTcw.translation();
}

// src/Frame.cc:482-488 through the SE3 product Twc * Tcb (the stand-in Eigen types carry no matrix operators)
Eigen::Vector3f Frame::GetImuPosition() const { return (mTcw.inverse() * mImuCalib.mTcb).translation(); }
Eigen::Matrix3f Frame::GetImuRotation() { return (mTcw.inverse() * mImuCalib.mTcb).rotationMatrix(); }

// src/Frame.cc:458-471
void Frame::SetImuPoseVelocity(const Eigen::Matrix3f& Rwb, const Eigen::Vector3f& twb, const Eigen::Vector3f& Vwb) {
  mVw = Vwb;
  mbHasVelocity = true;
  const Sophus::SE3f Twb(Rwb, twb);
  mTcw = mImuCalib.mTcb * Twb.inverse();
  ++mnPoseSets;
  UpdatePoseMatrices();
}

int Frame::isInFrustum(const std::vector<MapPoint*>& vpMPs, float viewingCosLimit, std::vector<bool>& vbInView) {
  const int n = (int)vpMPs.size();
  vbInView.assign(n, false);
  if (n == 0) return 0;
  if (Nleft != -1 || mpCamera2) {
    std::fprintf(stderr, "Frame::isInFrustum: fisheye-stereo frames (Nleft != -1) are not supported by the MI355X path yet\n");
    return -1;
  }
  osh_orb_ctx* ctx = HostMatcherContext();
  if (!ctx) return -1;
  osh_frustum_frame f;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) f.Rcw[3 * r + c] = mRcw(r, c);
    f.tcw[r] = mtcw(r); f.Ow[r] = mOw(r);
  }
  f.fx = fx; f.fy = fy; f.cx = cx; f.cy = cy; f.bf = mbf;
  f.min_x = mnMinX; f.max_x = mnMaxX; f.min_y = mnMinY; f.max_y = mnMaxY;
  f.log_scale_factor = mfLogScaleFactor; f.n_scale_levels = mnScaleLevels; f.viewing_cos_limit = viewingCosLimit;
  f.fisheye = (mpCamera && mpCamera->GetType() == GeometricCamera::CAM_FISHEYE) ? 1 : 0;   // mpCamera->project(Pc) (:532)
  for (int k = 0; k < 4; ++k) f.kb8[k] = f.fisheye ? mpCamera->getParameter(4 + k) : 0.f;
  std::vector<float> pos((size_t)n * 3), nrm((size_t)n * 3), dmin(n), dmax(n);
  for (int i = 0; i < n; ++i) {
    const Eigen::Vector3f P = vpMPs[i]->GetWorldPos(), Pn = vpMPs[i]->GetNormal();
    for (int k = 0; k < 3; ++k) { pos[3 * (size_t)i + k] = P(k); nrm[3 * (size_t)i + k] = Pn(k); }
    dmin[i] = vpMPs[i]->mfMinDistance; dmax[i] = vpMPs[i]->mfMaxDistance;
  }
  osh_frustum_points pts{n, pos.data(), nrm.data(), dmin.data(), dmax.data()};
  std::vector<uint8_t> stage(n);
  std::vector<float> px(n), py(n), pxr(n), depth(n), vcos(n);
  std::vector<int32_t> level(n);
  osh_frustum_result res{stage.data(), px.data(), py.data(), pxr.data(), depth.data(), vcos.data(), level.data()};
  if (osh_orb_frustum(ctx, &f, &pts, &res) != OSH_OK) {
    std::fprintf(stderr, "Frame::isInFrustum: %s\n", osh_last_error());
    return -1;
  }
  int n_in = 0;
  for (int i = 0; i < n; ++i) {
    MapPoint* pMP = vpMPs[i];
    pMP->mbTrackInView = false;                 // :515
    pMP->mTrackProjX = px[i];                   // -1 (:516-517) or uv (:539-540)
    pMP->mTrackProjY = py[i];
    if (stage[i] == 2) {                        // :563-571
      pMP->mbTrackInView = true;
      pMP->mTrackProjXR = pxr[i];
      pMP->mTrackDepth = depth[i];
      pMP->mnTrackScaleLevel = level[i];
      pMP->mTrackViewCos = vcos[i];
      vbInView[i] = true;
      ++n_in;
    }
  }
  return n_in;
}

bool Frame::isInFrustum(MapPoint* pMP, float viewingCosLimit) {
  std::vector<bool> in;
  return isInFrustum(std::vector<MapPoint*>{pMP}, viewingCosLimit, in) == 1;
}

}  // namespace ORB_SLAM3
