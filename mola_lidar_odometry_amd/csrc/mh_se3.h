// mh_se3.h -- fp64 SE(3) / small dense linear algebra used by the device-side solver and by the
// host-side API glue.  Conventions are those of mrpt::poses::CPose3D / Lie::SE<3> as used by the
// reference: pose = row-major 3x4 [R|t]; tangent = [v;w]; exp uses the
// V-matrix on the translation; ypr = Rz(yaw)*Ry(pitch)*Rx(roll) (LidarOdometry.cpp:235).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define MH_HD __host__ __device__ inline

namespace mh {

struct Pose {
  double m[12];  // r00 r01 r02 tx | r10 r11 r12 ty | r20 r21 r22 tz
  MH_HD double& R(int i, int j) { return m[i * 4 + j]; }
  MH_HD double R(int i, int j) const { return m[i * 4 + j]; }
  MH_HD double& t(int i) { return m[i * 4 + 3]; }
  MH_HD double t(int i) const { return m[i * 4 + 3]; }
};

MH_HD Pose pose_identity() {
  Pose p;
  for (int i = 0; i < 12; i++) p.m[i] = 0.0;
  p.m[0] = p.m[5] = p.m[10] = 1.0;
  return p;
}

// a (+) b
// (every loop over a pose's entries is unrolled: a runtime-indexed Pose lives in scratch memory on the device -- the serial lane
// of the solve paid 33 scratch round trips per step for compose / inverse before round 4 spelled this out)
MH_HD Pose compose(const Pose& a, const Pose& b) {
  Pose c;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) c.R(i, j) = a.R(i, 0) * b.R(0, j) + a.R(i, 1) * b.R(1, j) + a.R(i, 2) * b.R(2, j);
    c.t(i) = a.R(i, 0) * b.t(0) + a.R(i, 1) * b.t(1) + a.R(i, 2) * b.t(2) + a.t(i);
  }
  return c;
}

MH_HD Pose inverse(const Pose& a) {
  Pose c;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) c.R(i, j) = a.R(j, i);
    c.t(i) = -(a.R(0, i) * a.t(0) + a.R(1, i) * a.t(1) + a.R(2, i) * a.t(2));
  }
  return c;
}

// sin(th)/th, (1-cos th)/th^2, (th-sin th)/th^3 without cancellation
MH_HD void rodrigues_coeffs(double th, double& a, double& b, double& c) {
  const double t2 = th * th;
  if (th < 1e-2) {
    a = 1.0 - t2 * (1.0 / 6.0) * (1.0 - t2 * (1.0 / 20.0) * (1.0 - t2 * (1.0 / 42.0)));
    b = 0.5 - t2 * (1.0 / 24.0) * (1.0 - t2 * (1.0 / 30.0) * (1.0 - t2 * (1.0 / 56.0)));
    c = 1.0 / 6.0 - t2 * (1.0 / 120.0) * (1.0 - t2 * (1.0 / 42.0) * (1.0 - t2 * (1.0 / 72.0)));
  } else {
    const double s = sin(th), sh = sin(0.5 * th);
    a = s / th;
    b = 2.0 * sh * sh / t2;
    c = (th - s) / (t2 * th);
  }
}

// exp([v;w]) = (Rodrigues(w), V(w) v)
MH_HD Pose se3_exp(const double xi[6]) {
  const double wx = xi[3], wy = xi[4], wz = xi[5];
  const double th = sqrt(wx * wx + wy * wy + wz * wz);
  double a, b, c;
  rodrigues_coeffs(th, a, b, c);
  // W and W^2 (symmetric part) written out
  const double xx = wx * wx, yy = wy * wy, zz = wz * wz, xy = wx * wy, xz = wx * wz, yz = wy * wz;
  const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  const double W2[9] = {-(yy + zz), xy, xz, xy, -(xx + zz), yz, xz, yz, -(xx + yy)};
  Pose p;
  double V[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double I = (i == j) ? 1.0 : 0.0;
      p.R(i, j) = I + a * W[i * 3 + j] + b * W2[i * 3 + j];
      V[i * 3 + j] = I + b * W[i * 3 + j] + c * W2[i * 3 + j];
    }
#pragma unroll
  for (int i = 0; i < 3; i++) p.t(i) = V[i * 3] * xi[0] + V[i * 3 + 1] * xi[1] + V[i * 3 + 2] * xi[2];
  return p;
}

MH_HD void so3_log(const Pose& p, double w[3]) {
  double cth = 0.5 * (p.R(0, 0) + p.R(1, 1) + p.R(2, 2) - 1.0);
  cth = cth > 1.0 ? 1.0 : (cth < -1.0 ? -1.0 : cth);
  const double vx = p.R(2, 1) - p.R(1, 2), vy = p.R(0, 2) - p.R(2, 0), vz = p.R(1, 0) - p.R(0, 1);
  const double s2 = sqrt(vx * vx + vy * vy + vz * vz);  // 2 sin(th)
  const double th = atan2(0.5 * s2, cth);
  if (th < 1e-7) {
    const double k = 0.5 * (1.0 + th * th * (1.0 / 6.0));
    w[0] = k * vx; w[1] = k * vy; w[2] = k * vz;
    return;
  }
  if (3.141592653589793 - th > 1e-6) {
    const double k = th / s2;
    w[0] = k * vx; w[1] = k * vy; w[2] = k * vz;
    return;
  }
  // th ~ pi: R + I ~ 2 n n^T.  (Selected with compile-time indices only: `p.R(k, j)` with a runtime k would put the whole
  // pose -- and every pose that flows into this function -- into scratch memory on the device.)
  const double d0 = p.R(0, 0), d1 = p.R(1, 1), d2 = p.R(2, 2);
  const int k = (d0 >= d1 && d0 >= d2) ? 0 : (d1 >= d2 ? 1 : 2);
  const double dk = k == 0 ? d0 : (k == 1 ? d1 : d2);
  const double nk = sqrt(fmax(0.0, 0.5 * (dk + 1.0)));
  const double s01 = 0.25 * (p.R(0, 1) + p.R(1, 0)) / nk, s02 = 0.25 * (p.R(0, 2) + p.R(2, 0)) / nk, s12 = 0.25 * (p.R(1, 2) + p.R(2, 1)) / nk;
  const double n0 = k == 0 ? nk : (k == 1 ? s01 : s02);
  const double n1 = k == 1 ? nk : (k == 0 ? s01 : s12);
  const double n2 = k == 2 ? nk : (k == 0 ? s02 : s12);
  const double dot = n0 * vx + n1 * vy + n2 * vz;
  const double nn = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
  const double s = (dot < 0.0 ? -th : th) / nn;
  w[0] = s * n0; w[1] = s * n1; w[2] = s * n2;
}

MH_HD void se3_log(const Pose& p, double xi[6]) {
  double w[3];
  so3_log(p, w);
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double th = sqrt(th2);
  // V^-1 = I - W/2 + k W^2
  double k;
  if (th < 1e-2)
    k = 1.0 / 12.0 + th2 * (1.0 / 720.0) + th2 * th2 * (1.0 / 30240.0);
  else
    k = (1.0 - 0.5 * th / tan(0.5 * th)) / th2;
  const double xx = w[0] * w[0], yy = w[1] * w[1], zz = w[2] * w[2], xy = w[0] * w[1], xz = w[0] * w[2], yz = w[1] * w[2];
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  const double W2[9] = {-(yy + zz), xy, xz, xy, -(xx + zz), yz, xz, yz, -(xx + yy)};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 3; j++) acc += (((i == j) ? 1.0 : 0.0) - 0.5 * W[i * 3 + j] + k * W2[i * 3 + j]) * p.t(j);
    xi[i] = acc;
  }
  xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

MH_HD Pose pose_from_ypr(const double v[6]) {
  const double cy = cos(v[3]), sy = sin(v[3]), cp = cos(v[4]), sp = sin(v[4]), cr = cos(v[5]), sr = sin(v[5]);
  Pose p;
  p.R(0, 0) = cy * cp; p.R(0, 1) = cy * sp * sr - sy * cr; p.R(0, 2) = cy * sp * cr + sy * sr;
  p.R(1, 0) = sy * cp; p.R(1, 1) = sy * sp * sr + cy * cr; p.R(1, 2) = sy * sp * cr - cy * sr;
  p.R(2, 0) = -sp;     p.R(2, 1) = cp * sr;                p.R(2, 2) = cp * cr;
  p.t(0) = v[0]; p.t(1) = v[1]; p.t(2) = v[2];
  return p;
}

MH_HD void pose_to_ypr(const Pose& p, double v[6]) {
  v[0] = p.t(0); v[1] = p.t(1); v[2] = p.t(2);
  const double c = sqrt(p.R(0, 0) * p.R(0, 0) + p.R(1, 0) * p.R(1, 0));
  v[4] = atan2(-p.R(2, 0), c);
  if (c > 1e-12) {
    v[3] = atan2(p.R(1, 0), p.R(0, 0));
    v[5] = atan2(p.R(2, 1), p.R(2, 2));
  } else {
    v[3] = atan2(-p.R(0, 1), p.R(1, 1));
    v[5] = 0.0;
  }
}

// Symmetric 6x6, packed upper triangle: index of (i<=j)
MH_HD int sym6(int i, int j) {
  if (i > j) { const int t = i; i = j; j = t; }
  return i * 6 - (i * (i - 1)) / 2 + (j - i);
}

// x = H^-1 b by LDL^T with diagonal pivoting (what Eigen's ldlt() does for the reference's
// optimal_tf_gauss_newton); zero pivots give a zero component.  Returns false on non-finite results.
// Written with compile-time indices only so that on the device the whole 6x6 stays in registers -- a runtime-indexed
// array would live in scratch memory.  The pivot swaps are branches around compile-time-indexed swaps (one thread runs
// this: as predicated selects the 15 candidate swaps cost ~800 instructions, a third of the solve's time).
MH_HD void cswap(bool c, double& a, double& b) {
  const double ta = c ? b : a, tb = c ? a : b;
  a = ta;
  b = tb;
}

MH_HD bool ldlt_solve6(const double Hfull[36], const double b[6], double x[6]) {
#pragma clang fp contract(fast)  // one thread's dependent fp64 chain: fused multiply-adds halve it
  double A[6][6], y[6], invd[6];
  int piv[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j < 6; j++) A[i][j] = Hfull[i * 6 + j];
    y[i] = b[i];
  }
  const double tiny = 2.2250738585072014e-308;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    int p = k;
    double best = fabs(A[k][k]);
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      const double v = fabs(A[i][i]);
      if (v > best) { best = v; p = i; }
    }
    piv[k] = p;
#pragma unroll
    for (int i = k + 1; i < 6; i++) {
      if (p == i) {
#pragma unroll
        for (int j = 0; j < 6; j++) cswap(true, A[k][j], A[i][j]);
#pragma unroll
        for (int j = 0; j < 6; j++) cswap(true, A[j][k], A[j][i]);
        cswap(true, y[k], y[i]);
      }
    }
    const double d = A[k][k];
    ok = ok && isfinite(d);
    const bool nz = fabs(d) > tiny;
    const double inv = nz ? 1.0 / d : 0.0;
    invd[k] = inv;
#pragma unroll
    for (int i = k + 1; i < 6; i++) A[i][k] = nz ? A[i][k] * inv : 0.0;
#pragma unroll
    for (int i = k + 1; i < 6; i++)
#pragma unroll
      for (int j = k + 1; j <= i; j++) {
        A[i][j] -= A[i][k] * d * A[j][k];
        A[j][i] = A[i][j];
      }
  }
  // forward, diagonal, backward
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] *= invd[i];  // (0 where the pivot vanished)
#pragma unroll
  for (int i = 5; i >= 0; i--)
#pragma unroll
    for (int j = i + 1; j < 6; j++) y[i] -= A[j][i] * y[j];
  // undo the permutation: x = P^T y, swaps applied in reverse order
#pragma unroll
  for (int k = 5; k >= 0; k--)
#pragma unroll
    for (int i = k + 1; i < 6; i++)
      if (piv[k] == i) cswap(true, y[k], y[i]);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    x[i] = y[i];
    ok = ok && isfinite(y[i]);
  }
  return ok;
}

// The same solve without the pivot search, for the matrices the Gauss-Newton step actually sees: H = J^T W J (+ prior
// information) is symmetric positive definite, and LDL^T of an SPD matrix is stable in any order -- pivoting only decides
// what happens to (near-)singular ones.  Straight-line code, a third of the pivoted routine's dependent fp64 chain (one
// lane runs it between two launches of every ICP iteration).  Returns false -- the caller then takes ldlt_solve6 -- as
// soon as a pivot is not safely positive relative to the largest diagonal entry (rank-deficient or weakly constrained
// geometry: a plane-only scene, a featureless corridor) or anything is not finite; the two routines agree to rounding (a few 1e-16 of |x| times H's condition) on
// what this one accepts.
MH_HD bool ldlt_solve6_spd(const double Hfull[36], const double b[6], double x[6]) {
#pragma clang fp contract(fast)
  double A[6][6], y[6], invd[6];
  double dmax = 0.0;
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) A[i][j] = Hfull[i * 6 + j];
    y[i] = b[i];
    dmax = fmax(dmax, fabs(Hfull[i * 6 + i]));
  }
  const double floor_d = 1e-7 * dmax;  // (condition <= ~1e7: the two routines then agree to ~1e-9 relative even on the weakest direction)
  bool ok = dmax > 0.0 && isfinite(dmax);
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double d = A[k][k];
    ok = ok && (d > floor_d);  // (false for NaN as well)
    const double inv = 1.0 / d;
    invd[k] = inv;
    double l[6];
#pragma unroll
    for (int i = k + 1; i < 6; i++) l[i] = A[i][k] * inv;
#pragma unroll
    for (int i = k + 1; i < 6; i++)
#pragma unroll
      for (int j = k + 1; j <= i; j++) A[i][j] -= l[i] * A[j][k];  // A[j][k] still holds l_jk * d
#pragma unroll
    for (int i = k + 1; i < 6; i++) A[i][k] = l[i];
  }
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] *= invd[i];
#pragma unroll
  for (int i = 5; i >= 0; i--)
#pragma unroll
    for (int j = i + 1; j < 6; j++) y[i] -= A[j][i] * y[j];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    x[i] = y[i];
    ok = ok && isfinite(y[i]);
  }
  return ok;
}

// inverse of an SPD 6x6 via Cholesky (mrpt inverse_LLt in mp2p_icp::covariance); false if not SPD
MH_HD bool chol_inverse6(const double A[36], double Ainv[36]) {
  double L[36];
#pragma unroll
  for (int i = 0; i < 36; i++) L[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) {
      double s = A[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        L[i * 6 + i] = sqrt(s);
      } else {
        L[i * 6 + j] = s / L[j * 6 + j];
      }
    }
#pragma unroll
  for (int c = 0; c < 6; c++) {
    double y[6], x[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k];
      y[i] = s / L[i * 6 + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
      double s = y[i];
#pragma unroll
      for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k];
      x[i] = s / L[i * 6 + i];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) Ainv[i * 6 + c] = x[i];
  }
  return true;
}

}  // namespace mh
