/* ImuTypes.h -- IMU::Bias / Calib / Preintegrated as used by Optimizer::LocalInertialBA and EdgeInertial
 * (reference include/ImuTypes.h:43-230, src/ImuTypes.cc:147-309,398-410).  All arithmetic is float32, as in the
 * reference.  Own implementation in csrc/host/ImuTypes.cc (no Eigen): Eigen::JacobiSVD's U V^T is computed as the
 * orthogonal polar factor. */
#ifndef IMUTYPES_H
#define IMUTYPES_H
#include "orbslam3_compat.h"
namespace ORB_SLAM3 {
namespace IMU {
const float GRAVITY_VALUE = 9.81;
class Bias {
 public:
  Bias() : bax(0), bay(0), baz(0), bwx(0), bwy(0), bwz(0) {}
  Bias(const float& b_acc_x, const float& b_acc_y, const float& b_acc_z, const float& b_ang_vel_x, const float& b_ang_vel_y,
       const float& b_ang_vel_z)
      : bax(b_acc_x), bay(b_acc_y), baz(b_acc_z), bwx(b_ang_vel_x), bwy(b_ang_vel_y), bwz(b_ang_vel_z) {}
  float bax, bay, baz, bwx, bwy, bwz;
};
class Calib {
 public:
  Calib() {}
  Calib(const Sophus::SE3f& Tbc, const float& ng, const float& na, const float& ngw, const float& naw) { Set(Tbc, ng, na, ngw, naw); }
  void Set(const Sophus::SE3f& sophTbc, const float& ng, const float& na, const float& ngw, const float& naw);
  Sophus::SE3f mTcb, mTbc;
  float Cov[6] = {0, 0, 0, 0, 0, 0}, CovWalk[6] = {0, 0, 0, 0, 0, 0};   // diagonals (Eigen::DiagonalMatrix<float,6> in the reference)
  bool mbIsSet = false;
};
class Preintegrated {
 public:
  Preintegrated(const Bias& b_, const Calib& calib);
  void Initialize(const Bias& b_);
  void IntegrateNewMeasurement(const Eigen::Vector3f& acceleration, const Eigen::Vector3f& angVel, const float& dt);
  void SetNewBias(const Bias& bu_);
  Bias GetDeltaBias(const Bias& b_);
  float dT;
  Eigen::Matrix<float, 15, 15> C;
  float Nga[6], NgaWalk[6];
  Bias b;   // linearisation bias
  Eigen::Matrix3f dR;
  Eigen::Vector3f dV, dP;
  Eigen::Matrix3f JRg, JVg, JVa, JPg, JPa;
  Bias bu;  // updated bias
  float db[6];
};
}  // namespace IMU
}  // namespace ORB_SLAM3
#endif
