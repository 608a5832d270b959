/* KeyFrame.h -- members of ORB_SLAM3::KeyFrame used by Optimizer::LocalBundleAdjustment
 * (reference include/KeyFrame.h; src/KeyFrame.cc:109-131,237,309,367,681,1108).  Minimal test double. */
#ifndef KEYFRAME_H
#define KEYFRAME_H
#include <set>
#include <vector>
#include "CameraModels/GeometricCamera.h"
#include "ImuTypes.h"
#include "Map.h"
#include "MapPoint.h"
#include "orbslam3_compat.h"
namespace ORB_SLAM3 {
class KeyFrame {
 public:
  KeyFrame(long unsigned int id, Map* pMap) : mnId(id), mpMap(pMap) {}
  void SetPose(const Sophus::SE3f& Tcw);   // src/KeyFrame.cc:109-122 (also refreshes the IMU position Owb)
  Sophus::SE3f GetPose() { return mTcw; }
  // inertial accessors (src/KeyFrame.cc:124-181,809-833)
  void SetVelocity(const Eigen::Vector3f& Vw) { mVw = Vw; mbHasVelocity = true; }
  Eigen::Vector3f GetImuPosition() { return mOwb; }
  Eigen::Matrix3f GetImuRotation() { return (mTwc * mImuCalib.mTcb).rotationMatrix(); }
  Eigen::Matrix3f GetRotation() { return mRcw; }
  Eigen::Vector3f GetTranslation() { return mTcw.translation(); }
  Eigen::Vector3f GetVelocity() { return mVw; }
  void SetNewBias(const IMU::Bias& b) { mImuBias = b; if (mpImuPreintegrated) mpImuPreintegrated->SetNewBias(b); }
  Eigen::Vector3f GetGyroBias() { return Eigen::Vector3f(mImuBias.bwx, mImuBias.bwy, mImuBias.bwz); }
  Eigen::Vector3f GetAccBias() { return Eigen::Vector3f(mImuBias.bax, mImuBias.bay, mImuBias.baz); }
  IMU::Bias GetImuBias() { return mImuBias; }
  std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { return mvpOrderedConnectedKeyFrames; }
  std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
  std::set<MapPoint*> GetMapPoints();                        // src/KeyFrame.cc:336-349: the non-bad matches
  MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }   // src/KeyFrame.cc:373-377
  void EraseMapPointMatch(MapPoint* pMP);
  void EraseMapPointMatch(const int& idx) { mvpMapPoints[idx] = nullptr; }                  // src/KeyFrame.cc:303-307
  void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }           // src/KeyFrame.cc:297-301
  void ReplaceMapPointMatch(const int& idx, MapPoint* pMP) { mvpMapPoints[idx] = pMP; }     // src/KeyFrame.cc:320-323
  Eigen::Vector3f GetCameraCenter() { return mTwc.translation(); }                          // src/KeyFrame.cc:143-146
  Sophus::SE3f GetPoseInverse() { return mTwc; }                                            // src/KeyFrame.cc:133-136
  Sophus::SE3f GetRightPoseInverse() { return mTwc * mTrl.inverse(); }                      // src/KeyFrame.cc:1126-1130
  Sophus::SE3f GetRightPose() { return mTrl * mTcw; }                                       // src/KeyFrame.cc:1120-1124
  Eigen::Vector3f GetRightCameraCenter() { return (mTwc * mTrl.inverse()).translation(); }  // src/KeyFrame.cc:1132-1136 (mTlr = mTrl^-1)
  Sophus::SE3f GetRelativePoseTrl() { return mTrl; }
  bool isBad() { return mbBad; }
  Map* GetMap() { return mpMap; }
  // candidate generator of the Sim3 searches (src/KeyFrame.cc:704-750): the grid is the one of the Frame the keyframe was made from
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const bool bRight = false) const;
  bool IsInImage(const float& x, const float& y) const { return (x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY); }

  long unsigned int mnId;
  long unsigned int mnBALocalForKF = 0, mnBAFixedForKF = 0;
  long unsigned int mnBALocalForMerge = 0;   // include/KeyFrame.h:318
  // global BA results kept beside the live pose until the loop-closing thread applies them (include/KeyFrame.h:369-372)
  Sophus::SE3f mTcwGBA;
  long unsigned int mnBAGlobalForKF = 0;
  Eigen::Vector3f mVwbGBA;                   // include/KeyFrame.h:373-375: what FullInertialBA leaves for the loop closer
  IMU::Bias mBiasGBA;
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
  int N = 0, NLeft = -1;
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn, mvKeysRight;
  DBoW2::FeatureVector mFeatVec;   // include/KeyFrame.h:403 (filled by ComputeBoW)
  std::vector<float> mvuRight;
  std::vector<float> mvInvLevelSigma2;
  std::vector<float> mvLevelSigma2;
  std::vector<float> mvScaleFactors;
  int mnScaleLevels = 0;
  float mfLogScaleFactor = 0;
  cv::Mat mDescriptors;
  int mnGridCols = 0, mnGridRows = 0;
  float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
  int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;   // const int in the reference (include/KeyFrame.h:357-360)
  std::vector<std::vector<std::vector<size_t>>> mGrid;
  GeometricCamera* mpCamera = nullptr;
  GeometricCamera* mpCamera2 = nullptr;
  KeyFrame* mPrevKF = nullptr;
  KeyFrame* mNextKF = nullptr;               // include/KeyFrame.h:419-420
  bool bImu = false;
  IMU::Preintegrated* mpImuPreintegrated = nullptr;
  IMU::Calib mImuCalib;
  // test-double state
  Sophus::SE3f mTcw, mTrl, mTwc;
  Eigen::Matrix3f mRcw;
  Eigen::Vector3f mOwb, mVw;
  IMU::Bias mImuBias;
  bool mbHasVelocity = false;
  std::vector<KeyFrame*> mvpOrderedConnectedKeyFrames;
  std::vector<MapPoint*> mvpMapPoints;
  bool mbBad = false;
  Map* mpMap;
  int mnPoseSets = 0;
};
}  // namespace ORB_SLAM3
#endif
