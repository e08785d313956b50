#pragma once  // stand-in: the lightweight geometry types
#include <array>
namespace mrpt::math {
struct TPoint3Df { float x = 0, y = 0, z = 0; TPoint3Df() = default; TPoint3Df(float a, float b, float c) : x(a), y(b), z(c) {} };
struct TPoint2Df { float x = 0, y = 0; };
struct TPoint3D { double x = 0, y = 0, z = 0; TPoint3D() = default; TPoint3D(double a, double b, double c) : x(a), y(b), z(c) {} };
using TVector3D = TPoint3D;
struct TPose3D { double x = 0, y = 0, z = 0, yaw = 0, pitch = 0, roll = 0; };
struct TPlane { double coefs[4] = {0, 0, 1, 0}; TPlane() = default; TPlane(const TPoint3D&, const TVector3D&) {} TVector3D getNormalVector() const { return {coefs[0], coefs[1], coefs[2]}; } };
struct TBoundingBoxf { TPoint3Df min, max; };
template <int R, int C> struct CMatrixFixedD { double d[R][C] = {}; double& operator()(int r, int c) { return d[r][c]; } const double& operator()(int r, int c) const { return d[r][c]; }
  static CMatrixFixedD Identity() { CMatrixFixedD m; for (int i = 0; i < (R < C ? R : C); i++) m.d[i][i] = 1; return m; } };
using CMatrixDouble44 = CMatrixFixedD<4, 4>; using CMatrixDouble33 = CMatrixFixedD<3, 3>; using CMatrixDouble66 = CMatrixFixedD<6, 6>;
}
