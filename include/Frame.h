/* Frame.h -- members of ORB_SLAM3::Frame used by ORBmatcher::SearchByProjection
 * (reference include/Frame.h:112,144,213-246,250-252,281-290,326-329; src/Frame.cc:397-417,658-736).
 * Minimal test double: the monocular / rectified-stereo layout (Nleft == -1) and the fisheye stereo layout (Nleft != -1: keypoints
 * [0, Nleft) in mvKeys / mGrid, right-camera keypoints in mvKeysRight / mGridRight, descriptor rows [Nleft, N) for the right ones). */
#ifndef FRAME_H
#define FRAME_H
#include <vector>
#include "CameraModels/GeometricCamera.h"
#include "G2oTypes.h"
#include "ImuTypes.h"
#include "MapPoint.h"
#include "orbslam3_compat.h"
#define FRAME_GRID_ROWS 48
#define FRAME_GRID_COLS 64
namespace ORB_SLAM3 {
class KeyFrame;
class Frame {
 public:
  Frame() {}
  // candidate generator (src/Frame.cc:658-722): cells ix-major then iy, insertion order inside a cell
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel = -1,
                                        const int maxLevel = -1, const bool bRight = false) const;
  void AssignFeaturesToGrid();   // src/Frame.cc:397-417 with PosInGrid :726-736
  Sophus::SE3f GetPose() const { return mTcw; }
  void SetPose(const Sophus::SE3f& Tcw) { mTcw = Tcw; ++mnPoseSets; UpdatePoseMatrices(); }   // src/Frame.cc:291-296
  void UpdatePoseMatrices();     // src/Frame.cc:473-480
  // src/Frame.cc:513-587 (Nleft == -1).  The batch form is the loop of Tracking::SearchLocalPoints (src/Tracking.cc:3411-3432)
  // as one device launch: it fills the same MapPoint fields and returns how many points are in view (-1 on a device error).
  bool isInFrustum(MapPoint* pMP, float viewingCosLimit);
  int isInFrustum(const std::vector<MapPoint*>& vpMPs, float viewingCosLimit, std::vector<bool>& vbInView);

  int N = 0;
  int Nleft = -1, Nright = -1;
  float mbf = 0, mb = 0;
  float fx = 0, fy = 0, cx = 0, cy = 0;          // static members in the reference (include/Frame.h:208-213)
  std::vector<float> mvInvLevelSigma2;
  GeometricCamera* mpCamera2 = nullptr;
  int mnPoseSets = 0;                            // test-double bookkeeping
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn, mvKeysRight;
  std::vector<float> mvuRight;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  cv::Mat mDescriptors;
  DBoW2::FeatureVector mFeatVec;   // include/Frame.h:262 (filled by ComputeBoW)
  std::vector<float> mvScaleFactors;
  int mnScaleLevels = 0;
  float mfLogScaleFactor = 0;   // log(mfScaleFactor), src/Frame.cc:75
  GeometricCamera* mpCamera = nullptr;
  std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
  std::vector<std::size_t> mGridRight[FRAME_GRID_COLS][FRAME_GRID_ROWS];   // include/Frame.h:329
  std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;                 // include/Frame.h:301 (stereo fisheye matches)
  Sophus::SE3f mTrl;                                                        // include/Frame.h:297
  Sophus::SE3f GetRelativePoseTrl() const { return mTrl; }                  // src/Frame.cc:1134-1137
  float mnMinX = 0, mnMaxX = 752, mnMinY = 0, mnMaxY = 480;   // static in the reference
  float mfGridElementWidthInv = 64.f / 752.f, mfGridElementHeightInv = 48.f / 480.f;
  // IMU side (include/Frame.h:88-110,189-207,262-275; src/Frame.cc:447-492): what PoseInertialOptimizationLastKeyFrame / LastFrame touch
  Eigen::Vector3f GetImuPosition() const;        // mRwc * mImuCalib.mTcb.translation() + mOw
  Eigen::Matrix3f GetImuRotation();              // mRwc * mImuCalib.mTcb.rotationMatrix()
  Eigen::Vector3f GetVelocity() const { return mVw; }
  void SetVelocity(Eigen::Vector3f Vwb) { mVw = Vwb; mbHasVelocity = true; }
  void SetImuPoseVelocity(const Eigen::Matrix3f& Rwb, const Eigen::Vector3f& twb, const Eigen::Vector3f& Vwb);
  IMU::Calib mImuCalib;
  IMU::Bias mImuBias;
  IMU::Preintegrated* mpImuPreintegrated = nullptr;        // from the last keyframe
  IMU::Preintegrated* mpImuPreintegratedFrame = nullptr;   // from the previous frame
  KeyFrame* mpLastKeyFrame = nullptr;
  Frame* mpPrevFrame = nullptr;
  ConstraintPoseImu* mpcpi = nullptr;
  Eigen::Vector3f mVw;
  bool mbHasVelocity = false;
  long unsigned int mnId = 0;
  Sophus::SE3f mTcw;
  Eigen::Matrix3f mRcw, mRwc;      // include/Frame.h:337-340 (private in the reference)
  Eigen::Vector3f mtcw, mOw;
};
}  // namespace ORB_SLAM3
#endif
