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
  // include/CameraModels/GeometricCamera.h:79
  virtual bool epipolarConstrain(GeometricCamera* otherCamera, const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const Eigen::Matrix3f& R12,
                                 const Eigen::Vector3f& t12, const float sigmaLevel, const float unc) = 0;
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
  // src/CameraModels/Pinhole.cpp:107-129: distance of kp2 to the epipolar line of kp1, F12 = K1^-T [t12]x R12 K2^-1
  bool epipolarConstrain(GeometricCamera* pCamera2, const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const Eigen::Matrix3f& R12,
                         const Eigen::Vector3f& t12, const float, const float unc) override {
    const float fx1 = mvParameters[0], fy1 = mvParameters[1], cx1 = mvParameters[2], cy1 = mvParameters[3];
    const float fx2 = pCamera2->getParameter(0), fy2 = pCamera2->getParameter(1), cx2 = pCamera2->getParameter(2), cy2 = pCamera2->getParameter(3);
    const float tx[9] = {0.f, -t12(2), t12(1), t12(2), 0.f, -t12(0), -t12(1), t12(0), 0.f};
    float E[9];                                       // [t12]x R12
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) E[i * 3 + j] = tx[i * 3] * R12(0, j) + tx[i * 3 + 1] * R12(1, j) + tx[i * 3 + 2] * R12(2, j);
    // K^-1 = [1/fx 0 -cx/fx; 0 1/fy -cy/fy; 0 0 1]
    const float k1[9] = {1.f / fx1, 0.f, 0.f, 0.f, 1.f / fy1, 0.f, -cx1 / fx1, -cy1 / fy1, 1.f};   // K1^-T
    const float k2[9] = {1.f / fx2, 0.f, -cx2 / fx2, 0.f, 1.f / fy2, -cy2 / fy2, 0.f, 0.f, 1.f};   // K2^-1
    float A[9], F[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) A[i * 3 + j] = k1[i * 3] * E[j] + k1[i * 3 + 1] * E[3 + j] + k1[i * 3 + 2] * E[6 + j];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) F[i * 3 + j] = A[i * 3] * k2[j] + A[i * 3 + 1] * k2[3 + j] + A[i * 3 + 2] * k2[6 + j];
    const float a = kp1.pt.x * F[0] + kp1.pt.y * F[3] + F[6];
    const float b = kp1.pt.x * F[1] + kp1.pt.y * F[4] + F[7];
    const float c = kp1.pt.x * F[2] + kp1.pt.y * F[5] + F[8];
    const float num = a * kp2.pt.x + b * kp2.pt.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * unc;
  }
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
  // src/CameraModels/KannalaBrandt8.cpp:232-235 triangulates the pair (TriangulateMatches); the test double does not restate it
  bool epipolarConstrain(GeometricCamera*, const cv::KeyPoint&, const cv::KeyPoint&, const Eigen::Matrix3f&, const Eigen::Vector3f&, const float,
                         const float) override { return false; }
};
}  // namespace ORB_SLAM3
#endif
