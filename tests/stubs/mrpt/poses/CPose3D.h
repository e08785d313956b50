#pragma once  // stand-in
#include <mrpt/math/types.h>
namespace mrpt::poses {
class CPose3D { public:
  std::array<double, 3> m_coords{{0, 0, 0}};
  CPose3D() = default; explicit CPose3D(const mrpt::math::TPose3D&) {} explicit CPose3D(const mrpt::math::CMatrixDouble44&) {}
  static CPose3D Identity() { return CPose3D(); }
  const mrpt::math::CMatrixDouble33& getRotationMatrix() const { return R_; }
  double x() const { return m_coords[0]; } double y() const { return m_coords[1]; } double z() const { return m_coords[2]; }
  double yaw() const { return 0; } double pitch() const { return 0; } double roll() const { return 0; }
  CPose3D operator+(const CPose3D&) const { return *this; }
 private: mrpt::math::CMatrixDouble33 R_ = mrpt::math::CMatrixDouble33::Identity(); };
struct CPose3DPDFGaussian { CPose3D mean; mrpt::math::CMatrixDouble66 cov; };
struct CPose3DPDFGaussianInf { CPose3D mean; mrpt::math::CMatrixDouble66 cov_inv; };
}
