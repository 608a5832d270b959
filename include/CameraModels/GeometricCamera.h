/* GeometricCamera.h -- the members of ORB_SLAM3::GeometricCamera the local-BA boundary touches
 * (reference include/CameraModels/GeometricCamera.h:62-105): Pinhole (src/CameraModels/Pinhole.cpp:35-81) and the
 * members of KannalaBrandt8 the packers read (include/CameraModels/KannalaBrandt8.h; parameters fx fy cx cy k1..k4). */
#ifndef CAMERAMODELS_GEOMETRICCAMERA_H
#define CAMERAMODELS_GEOMETRICCAMERA_H
#include <cmath>
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
// Test double of the fisheye model: the packers only read its type and parameters (the projection runs on the device);
// project() is the formula of src/CameraModels/KannalaBrandt8.cpp:45-63 for completeness.
class KannalaBrandt8 : public GeometricCamera {
 public:
  explicit KannalaBrandt8(const std::vector<float>& p) : GeometricCamera(p) { mnType = CAM_FISHEYE; }
  Eigen::Vector2d project(const Eigen::Vector3d& v) override {
    const double x2_plus_y2 = v[0] * v[0] + v[1] * v[1];
    const double theta = atan2f(sqrtf(x2_plus_y2), v[2]), psi = atan2f(v[1], v[0]);
    const double t2 = theta * theta, t3 = theta * t2, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
    const double r = theta + mvParameters[4] * t3 + mvParameters[5] * t5 + mvParameters[6] * t7 + mvParameters[7] * t9;
    return Eigen::Vector2d(mvParameters[0] * r * cos(psi) + mvParameters[2], mvParameters[1] * r * sin(psi) + mvParameters[3]);
  }
  Eigen::Vector2f project(const Eigen::Vector3f& v) override {
    const Eigen::Vector2d d = project(Eigen::Vector3d(v[0], v[1], v[2]));
    return Eigen::Vector2f((float)d[0], (float)d[1]);
  }
  float uncertainty2(const Eigen::Matrix<double, 2, 1>&) override { return 1.0; }
};
}  // namespace ORB_SLAM3
#endif
