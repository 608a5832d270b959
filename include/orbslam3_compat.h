/*
 * orbslam3_compat.h -- tiny stand-ins for the third-party types that appear in the signatures
 * of the hot path's interface (Eigen::Vector3f/Quaternionf, Sophus::SE3f, cv::KeyPoint, cv::Mat),
 * so that the host layer of THIS repository compiles in an image without Eigen/Sophus/OpenCV.
 *
 * They expose only the members the local-BA / matcher boundary touches.  Inside a
 * real ORB-SLAM3 tree define ORBSLAM3_HIP_USE_REAL_HEADERS and the genuine headers are used
 * instead; csrc/host/Optimizer.cc and ORBmatcher.cc only rely on the common subset
 * (operator(), x()/y()/z()/w(), cast<T>(), unit_quaternion(), translation(), pt/octave/angle,
 * ptr<T>(row), rows/cols).
 */
#ifndef ORBSLAM3_COMPAT_H
#define ORBSLAM3_COMPAT_H

#ifdef ORBSLAM3_HIP_USE_REAL_HEADERS
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <opencv2/core/core.hpp>
#include <sophus/se3.hpp>
#else

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace Eigen {
template <class T, int R, int C = 1>
struct Matrix {
  T v[R * C];
  Matrix() { for (int i = 0; i < R * C; ++i) v[i] = T(0); }
  Matrix(T a, T b) { static_assert(R * C == 2, "size"); v[0] = a; v[1] = b; }
  Matrix(T a, T b, T c) { static_assert(R * C == 3, "size"); v[0] = a; v[1] = b; v[2] = c; }
  T& operator()(int i) { return v[i]; }
  const T& operator()(int i) const { return v[i]; }
  T& operator[](int i) { return v[i]; }
  const T& operator[](int i) const { return v[i]; }
  T& operator()(int r, int c) { return v[r * C + c]; }
  const T& operator()(int r, int c) const { return v[r * C + c]; }
  template <class U> Matrix<U, R, C> cast() const { Matrix<U, R, C> o; for (int i = 0; i < R * C; ++i) o.v[i] = (U)v[i]; return o; }
};
using Matrix3f = Matrix<float, 3, 3>;
using Matrix3d = Matrix<double, 3, 3>;
using Vector2d = Matrix<double, 2>;
using Vector3d = Matrix<double, 3>;
using Vector3f = Matrix<float, 3>;
using Vector2f = Matrix<float, 2>;

template <class T>
struct Quaternion {
  T qx, qy, qz, qw;
  Quaternion() : qx(0), qy(0), qz(0), qw(1) {}
  Quaternion(T w, T x, T y, T z) : qx(x), qy(y), qz(z), qw(w) {}   // Eigen order: w first
  T x() const { return qx; } T y() const { return qy; } T z() const { return qz; } T w() const { return qw; }
  template <class U> Quaternion<U> cast() const { return Quaternion<U>((U)qw, (U)qx, (U)qy, (U)qz); }
  void normalize() { const T n = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw); qx /= n; qy /= n; qz /= n; qw /= n; }
};
using Quaternionf = Quaternion<float>;
using Quaterniond = Quaternion<double>;
}  // namespace Eigen

namespace Sophus {
// SE3 as unit quaternion + translation (the storage KeyFrame::GetPose / SetPose exchange).
template <class T>
class SE3 {
 public:
  SE3() {}
  SE3(const Eigen::Quaternion<T>& q, const Eigen::Matrix<T, 3>& t) : q_(q), t_(t) { q_.normalize(); }  // SO3 ctor normalises
  // from a rotation matrix (Eigen's matrix -> quaternion conversion)
  SE3(const Eigen::Matrix<T, 3, 3>& R, const Eigen::Matrix<T, 3>& t) : t_(t) {
    T tr = R(0, 0) + R(1, 1) + R(2, 2), x, y, z, w;
    if (tr > T(0)) {
      T s = std::sqrt(tr + T(1)); w = T(0.5) * s; s = T(0.5) / s;
      x = (R(2, 1) - R(1, 2)) * s; y = (R(0, 2) - R(2, 0)) * s; z = (R(1, 0) - R(0, 1)) * s;
    } else {
      int i = 0;
      if (R(1, 1) > R(0, 0)) i = 1;
      if (R(2, 2) > R(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      T s = std::sqrt(R(i, i) - R(j, j) - R(k, k) + T(1));
      T q[3];
      q[i] = T(0.5) * s; s = T(0.5) / s;
      w = (R(k, j) - R(j, k)) * s; q[j] = (R(j, i) + R(i, j)) * s; q[k] = (R(k, i) + R(i, k)) * s;
      x = q[0]; y = q[1]; z = q[2];
    }
    q_ = Eigen::Quaternion<T>(w, x, y, z);
    q_.normalize();
  }
  Eigen::Matrix<T, 3, 3> rotationMatrix() const {
    const T x = q_.x(), y = q_.y(), z = q_.z(), w = q_.w();
    const T tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
    const T txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    Eigen::Matrix<T, 3, 3> R;
    R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
    R(1, 0) = txy + twz; R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1 - (txx + tyy);
    return R;
  }
  SE3 inverse() const {
    SE3 o;
    o.q_ = Eigen::Quaternion<T>(q_.w(), -q_.x(), -q_.y(), -q_.z());
    SE3 rot; rot.q_ = o.q_;
    const Eigen::Matrix<T, 3> mt(-t_(0), -t_(1), -t_(2));
    o.t_ = rot * mt;
    return o;
  }
  SE3 operator*(const SE3& b) const {   // composition
    SE3 o;
    const T aw = q_.w(), ax = q_.x(), ay = q_.y(), az = q_.z(), bw = b.q_.w(), bx = b.q_.x(), by = b.q_.y(), bz = b.q_.z();
    o.q_ = Eigen::Quaternion<T>(aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                                aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx);
    o.t_ = (*this) * b.t_;
    return o;
  }
  const Eigen::Quaternion<T>& unit_quaternion() const { return q_; }
  const Eigen::Matrix<T, 3>& translation() const { return t_; }
  template <class U> SE3<U> cast() const {   // member-wise cast, no renormalisation (Sophus: SE3<U>(so3.cast<U>(), t.cast<U>()))
    SE3<U> o; o.set_raw(q_.template cast<U>(), t_.template cast<U>()); return o;
  }
  void set_raw(const Eigen::Quaternion<T>& q, const Eigen::Matrix<T, 3>& t) { q_ = q; t_ = t; }
  // p_out = R p + t
  Eigen::Matrix<T, 3> operator*(const Eigen::Matrix<T, 3>& p) const {
    const T x = q_.x(), y = q_.y(), z = q_.z(), w = q_.w();
    const T uvx = T(2) * (y * p(2) - z * p(1)), uvy = T(2) * (z * p(0) - x * p(2)), uvz = T(2) * (x * p(1) - y * p(0));
    return Eigen::Matrix<T, 3>(p(0) + w * uvx + (y * uvz - z * uvy) + t_(0), p(1) + w * uvy + (z * uvx - x * uvz) + t_(1),
                               p(2) + w * uvz + (x * uvy - y * uvx) + t_(2));
  }
 private:
  Eigen::Quaternion<T> q_;
  Eigen::Matrix<T, 3> t_;
};
using SE3f = SE3<float>;
using SE3d = SE3<double>;
// Sim3 as rotation + translation + scale; only what ORBmatcher::SearchByProjection(KeyFrame*, Sim3f&, ...) and SearchBySim3 use.
template <class T>
class Sim3 {
 public:
  Sim3() : s_(1) {}
  Sim3(const Eigen::Quaternion<T>& unit_q, const Eigen::Matrix<T, 3>& t, T scale) : rot_(unit_q, Eigen::Matrix<T, 3>(0, 0, 0)), t_(t), s_(scale) {}
  Eigen::Matrix<T, 3, 3> rotationMatrix() const { return rot_.rotationMatrix(); }
  const Eigen::Matrix<T, 3>& translation() const { return t_; }
  T scale() const { return s_; }
  // p_out = s R p + t (what ORBmatcher::SearchBySim3 applies to camera-frame points)
  Eigen::Matrix<T, 3> operator*(const Eigen::Matrix<T, 3>& p) const {
    const Eigen::Matrix<T, 3> r = rot_ * p;
    return Eigen::Matrix<T, 3>(s_ * r(0) + t_(0), s_ * r(1) + t_(1), s_ * r(2) + t_(2));
  }
  // (s R, t)^-1 = (R^T / s, -R^T t / s)
  Sim3 inverse() const {
    const Eigen::Quaternion<T>& q = rot_.unit_quaternion();
    const Eigen::Quaternion<T> qi(q.w(), -q.x(), -q.y(), -q.z());
    const SE3<T> ri(qi, Eigen::Matrix<T, 3>(0, 0, 0));
    const Eigen::Matrix<T, 3> rt = ri * t_;
    return Sim3(qi, Eigen::Matrix<T, 3>(-rt(0) / s_, -rt(1) / s_, -rt(2) / s_), T(1) / s_);
  }
 private:
  SE3<T> rot_;
  Eigen::Matrix<T, 3> t_;
  T s_;
};
using Sim3f = Sim3<float>;
}  // namespace Sophus

// g2o::Sim3 (Thirdparty/g2o/g2o/types/sim3.h:45-80): what Optimizer::MergeInertialBA leaves in LoopClosing::KeyFrameAndPose
namespace g2o {
class Sim3 {
 public:
  Sim3() : s_(1.0) {}
  Sim3(const Eigen::Quaterniond& r, const Eigen::Vector3d& t, double s) : r_(r), t_(t), s_(s) {}
  const Eigen::Quaterniond& rotation() const { return r_; }
  const Eigen::Vector3d& translation() const { return t_; }
  const double& scale() const { return s_; }
 private:
  Eigen::Quaterniond r_;
  Eigen::Vector3d t_;
  double s_;
};
}  // namespace g2o
namespace Eigen {
template <class T> using aligned_allocator = std::allocator<T>;
struct VectorXd { std::vector<double> v; };   // only named by Optimizer::FullInertialBA's unused vSingVal parameter
}  // namespace Eigen

// DBoW2::FeatureVector / BowVector (Thirdparty/DBoW2/DBoW2/FeatureVector.h:23, BowVector.h:48): vocabulary node -> indices of the
// local features that fell into it; ORBmatcher::SearchByBoW walks two of them in step
#include <map>
namespace DBoW2 {
typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector;
typedef std::map<unsigned int, double> BowVector;
}  // namespace DBoW2

namespace cv {
struct Point2f { float x = 0, y = 0; };
struct KeyPoint { Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1; };
// Row-major byte matrix (only CV_8U descriptor matrices N x 32 cross this boundary).
class Mat {
 public:
  int rows = 0, cols = 0;
  Mat() {}
  Mat(int r, int c) : rows(r), cols(c), buf_(new std::vector<uint8_t>((size_t)r * c, 0)), off_(0) {}
  bool empty() const { return rows == 0; }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(buf_->data() + off_ + (size_t)r * cols); }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(buf_->data() + off_ + (size_t)r * cols); }
  Mat row(int r) const { Mat m; m.rows = 1; m.cols = cols; m.buf_ = buf_; m.off_ = off_ + (size_t)r * cols; return m; }
  Mat clone() const { Mat m(rows, cols); if (rows) std::memcpy(m.buf_->data(), buf_->data() + off_, (size_t)rows * cols); return m; }
 private:
  std::shared_ptr<std::vector<uint8_t>> buf_;
  size_t off_ = 0;
};
}  // namespace cv

#ifndef EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#endif
#endif /* ORBSLAM3_HIP_USE_REAL_HEADERS */
#endif
