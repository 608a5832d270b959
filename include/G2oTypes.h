/* G2oTypes.h -- ConstraintPoseImu, the only type of the reference's include/G2oTypes.h that crosses the boundary of
 * Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame (Frame::mpcpi, reference include/G2oTypes.h:706-730).
 * Minimal test double: same members; the constructor symmetrises H and drops its eigenvalues below 1e-12 like the reference's
 * (SelfAdjointEigenSolver there, cyclic Jacobi rotations here).  A real ORB-SLAM3 tree supplies its own header. */
#ifndef G2OTYPES_H
#define G2OTYPES_H
#include <cmath>
#include "orbslam3_compat.h"
namespace ORB_SLAM3 {
typedef Eigen::Matrix<double, 15, 15> Matrix15d;
class ConstraintPoseImu {
 public:
  ConstraintPoseImu(const Eigen::Matrix3d& Rwb_, const Eigen::Vector3d& twb_, const Eigen::Vector3d& vwb_, const Eigen::Vector3d& bg_,
                    const Eigen::Vector3d& ba_, const Matrix15d& H_)
      : Rwb(Rwb_), twb(twb_), vwb(vwb_), bg(bg_), ba(ba_), H(H_) {
    const int n = 15;
    double A[225], V[225], w[15];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { A[i * n + j] = 0.5 * (H(i, j) + H(j, i)); V[i * n + j] = (i == j); }
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
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i] < 1e-12 ? 0.0 : A[i * n + i];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += V[i * n + k] * w[k] * V[j * n + k]; H(i, j) = s; }
  }
  Eigen::Matrix3d Rwb;
  Eigen::Vector3d twb, vwb, bg, ba;
  Matrix15d H;
};
}  // namespace ORB_SLAM3
#endif
