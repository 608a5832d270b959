/* ORBmatcher.h -- the SearchByProjection entry points of ORB_SLAM3::ORBmatcher on the hot path, with the
 * reference's signatures (include/ORBmatcher.h:36-102). */
#ifndef ORBMATCHER_H
#define ORBMATCHER_H
#include <set>
#include <vector>
#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
#include "orbslam3_compat.h"
namespace ORB_SLAM3 {
class ORBmatcher {
 public:
  ORBmatcher(float nnratio = 0.6, bool checkOri = true);
  // src/ORBmatcher.cc:2058-2074 (kept on the host for single pairs; batches go through osh_orb_*)
  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);
  // src/ORBmatcher.cc:43-213 (TrackLocalMap): device nearest/second-nearest search with the sequential "slot already taken"
  // rule resolved on the device; fisheye stereo frames: two searches + host replay of the stereo-partner claims.
  int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3, const bool bFarPoints = false,
                         const float thFarPoints = 50.0f);
  // src/ORBmatcher.cc:1676-1887 (TrackWithMotionModel)
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
  // src/ORBmatcher.cc:1889-2010 (Relocalization): best candidate only, any occupied slot is skipped, accept <= ORBdist
  int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                         const int ORBdist);
  // src/ORBmatcher.cc:427-532 and :534-646 (loop closing / map merging): map points projected with a Sim3 into a keyframe,
  // level filter inside the candidate loop, any matched slot is skipped, accept bestDist <= TH_LOW * ratioHamming (float)
  int SearchByProjection(KeyFrame* pKF, Sophus::Sim3f& Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched,
                         int th, float ratioHamming = 1.0);
  int SearchByProjection(KeyFrame* pKF, Sophus::Sim3<float>& Scw, const std::vector<MapPoint*>& vpPoints,
                         const std::vector<KeyFrame*>& vpPointsKFs, std::vector<MapPoint*>& vpMatched,
                         std::vector<KeyFrame*>& vpMatchedKF, int th, float ratioHamming = 1.0);
  // src/ORBmatcher.cc:223-420 (TrackReferenceKeyFrame, Relocalization): the features of the keyframe that hold a map point against
  // the frame's features of the same vocabulary node; ratio test, rotation histogram; monocular / stereo and fisheye stereo frames
  int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
  // src/ORBmatcher.cc:765-905 (loop closing / map merging): map points of keyframe 1 against the map points of keyframe 2 of the same node
  int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);
  // Matching for the Map Initialization (only used in the monocular case) (include/ORBmatcher.h:72, src/ORBmatcher.cc:648-763)
  int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);
  // Matching to triangulate new MapPoints. Check Epipolar Constraint. (include/ORBmatcher.h:75-76, src/ORBmatcher.cc:907-1146)
  int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo,
                             const bool bCoarse = false);
  // Search matches between MapPoints seen in KF1 and KF2 transforming by a Sim3 [s12*R12|t12] (include/ORBmatcher.h:78-81,
  // src/ORBmatcher.cc:1457-1674; LoopClosing::DetectCommonRegionsFromBoW after the Sim3 solver)
  int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const Sophus::Sim3f& S12, const float th);
  // Project MapPoints into KeyFrame and search for duplicated MapPoints (include/ORBmatcher.h:87, src/ORBmatcher.cc:1148-1338)
  int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0, const bool bRight = false);
  // Project MapPoints into KeyFrame using a given Sim3 and search for duplicated MapPoints (include/ORBmatcher.h:90, src/ORBmatcher.cc:1340-1455)
  int Fuse(KeyFrame* pKF, Sophus::Sim3f& Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint);
  static const int TH_LOW;
  static const int TH_HIGH;
  static const int HISTO_LENGTH;
 protected:
  float RadiusByViewingCos(const float& viewCos);
  void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);
  float mfNNratio;
  bool mbCheckOrientation;
};
}  // namespace ORB_SLAM3
#endif
