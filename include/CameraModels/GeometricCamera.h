/* GeometricCamera.h -- the members of ORB_SLAM3::GeometricCamera the local-BA boundary touches
 * (reference include/CameraModels/GeometricCamera.h:62-105).  Pinhole only (src/CameraModels/Pinhole.cpp:35-81);
 * KannalaBrandt8 is a "next" row. */
#ifndef CAMERAMODELS_GEOMETRICCAMERA_H
#define CAMERAMODELS_GEOMETRICCAMERA_H
#include <vector>
#include "../orbslam3_compat.h"
namespace ORB_SLAM3 {
class GeometricCamera {
 public:
  GeometricCamera() {}
  explicit GeometricCamera(const std::vector<float>& p) : mvParameters(p) {}
  virtual ~GeometricCamera() {}
  virtual Eigen::Vector2d project(const Eigen::Vector3d& v3D) = 0;
  virtual Eigen::Vector2f project(const Eigen::Vector3f& v3D) = 0;
  virtual float uncertainty2(const Eigen::Matrix<double, 2, 1>& p2D) = 0;
  float getParameter(const int i) { return mvParameters[i]; }
  unsigned int GetType() { return mnType; }
  const static unsigned int CAM_PINHOLE = 0;
  const static unsigned int CAM_FISHEYE = 1;
 protected:
  std::vector<float> mvParameters;
  unsigned int mnType = CAM_PINHOLE;
};
class Pinhole : public GeometricCamera {
 public:
  explicit Pinhole(const std::vector<float>& p) : GeometricCamera(p) { mnType = CAM_PINHOLE; }
  Eigen::Vector2d project(const Eigen::Vector3d& v) override {
    return Eigen::Vector2d(mvParameters[0] * v[0] / v[2] + mvParameters[2], mvParameters[1] * v[1] / v[2] + mvParameters[3]);
  }
  Eigen::Vector2f project(const Eigen::Vector3f& v) override {
    return Eigen::Vector2f(mvParameters[0] * v[0] / v[2] + mvParameters[2], mvParameters[1] * v[1] / v[2] + mvParameters[3]);
  }
  float uncertainty2(const Eigen::Matrix<double, 2, 1>&) override { return 1.0; }
};
}  // namespace ORB_SLAM3
#endif
