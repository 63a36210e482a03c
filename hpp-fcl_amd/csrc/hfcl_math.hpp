// hfcl_math.hpp -- scalar-generic 3-vector helpers for the device narrow phase.
// Everything here is register-only, branch-free POD arithmetic usable from HIP device
// code (gfx950) and, for the CPU-side validation build (tests/hostsim), from plain g++.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HFCL_HD __host__ __device__ __forceinline__
#define HFCL_D __device__ __forceinline__
#else
#define HFCL_HD inline __attribute__((always_inline))
#define HFCL_D inline __attribute__((always_inline))
#endif

namespace hfcl {

template <typename T>
struct V3 {
  T x, y, z;
};

template <typename T> HFCL_HD V3<T> mk(T x, T y, T z) { V3<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <typename T> HFCL_HD V3<T> operator+(const V3<T>& a, const V3<T>& b) { return mk<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> HFCL_HD V3<T> operator-(const V3<T>& a, const V3<T>& b) { return mk<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> HFCL_HD V3<T> operator-(const V3<T>& a) { return mk<T>(-a.x, -a.y, -a.z); }
template <typename T> HFCL_HD V3<T> operator*(T s, const V3<T>& a) { return mk<T>(s * a.x, s * a.y, s * a.z); }
template <typename T> HFCL_HD V3<T> operator*(const V3<T>& a, T s) { return mk<T>(a.x * s, a.y * s, a.z * s); }
template <typename T> HFCL_HD V3<T> operator/(const V3<T>& a, T s) { return mk<T>(a.x / s, a.y / s, a.z / s); }
template <typename T> HFCL_HD T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> HFCL_HD V3<T> cross(const V3<T>& a, const V3<T>& b) {
  return mk<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename T> HFCL_HD T sqnorm(const V3<T>& a) { return dot(a, a); }

HFCL_HD float hsqrt(float x) { return sqrtf(x); }
HFCL_HD double hsqrt(double x) { return sqrt(x); }
HFCL_HD float habs(float x) { return fabsf(x); }
HFCL_HD double habs(double x) { return fabs(x); }
template <typename T> HFCL_HD T hmax(T a, T b) { return a > b ? a : b; }
template <typename T> HFCL_HD T hmin(T a, T b) { return a < b ? a : b; }

template <typename T> HFCL_HD T norm(const V3<T>& a) { return hsqrt(sqnorm(a)); }
// Eigen normalized(): divide only when squaredNorm() > 0
template <typename T> HFCL_HD V3<T> normalized(const V3<T>& a) {
  T z = sqnorm(a);
  T inv = z > T(0) ? hsqrt(z) : T(1);
  return mk<T>(a.x / inv, a.y / inv, a.z / inv);
}
template <typename T> HFCL_HD T triple(const V3<T>& a, const V3<T>& b, const V3<T>& c) { return dot(a, cross(b, c)); }
template <typename T> HFCL_HD V3<T> sel(bool c, const V3<T>& a, const V3<T>& b) {
  return mk<T>(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z);
}
template <typename T> HFCL_HD T comp(const V3<T>& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

template <typename T> struct Lim;
template <> struct Lim<float> {
  static HFCL_HD float max() { return 3.402823466e+38f; }
  static HFCL_HD float eps() { return 1.192092896e-07f; }
  static HFCL_HD float nan() { return __builtin_nanf(""); }
  // Eigen::NumTraits<float>::dummy_precision()
  static HFCL_HD float dummy() { return 1e-5f; }
  // The reference's 1e-12 dead zones (identity test, Box/Capsule support) are kept at 1e-12 in
  // fp32 as well: there they degenerate to "exactly zero / exactly identity", which is the
  // behaviour closest to the fp64 reference.
  static HFCL_HD float tiny() { return 1e-12f; }
};
template <> struct Lim<double> {
  static HFCL_HD double max() { return 1.7976931348623157e+308; }
  static HFCL_HD double eps() { return 2.2204460492503131e-16; }
  static HFCL_HD double nan() { return __builtin_nan(""); }
  static HFCL_HD double dummy() { return 1e-12; }
  static HFCL_HD double tiny() { return 1e-12; }
};

// Rotation as three rows.  (R v)_i = dot(r[i], v);  (R^T v) = v.x r0 + v.y r1 + v.z r2.
template <typename T>
struct M3 {
  V3<T> r0, r1, r2;
};
template <typename T> HFCL_HD V3<T> mul(const M3<T>& A, const V3<T>& v) { return mk<T>(dot(A.r0, v), dot(A.r1, v), dot(A.r2, v)); }
template <typename T> HFCL_HD V3<T> tmul(const M3<T>& A, const V3<T>& v) {
  return mk<T>(A.r0.x * v.x + A.r1.x * v.y + A.r2.x * v.z, A.r0.y * v.x + A.r1.y * v.y + A.r2.y * v.z,
               A.r0.z * v.x + A.r1.z * v.y + A.r2.z * v.z);
}
template <typename T> HFCL_HD V3<T> col(const M3<T>& A, int c) { return mk<T>(comp(A.r0, c), comp(A.r1, c), comp(A.r2, c)); }
// A^T * B
template <typename T> HFCL_HD M3<T> tmul(const M3<T>& A, const M3<T>& B) {
  M3<T> R;
  R.r0 = mk<T>(A.r0.x * B.r0.x + A.r1.x * B.r1.x + A.r2.x * B.r2.x, A.r0.x * B.r0.y + A.r1.x * B.r1.y + A.r2.x * B.r2.y,
               A.r0.x * B.r0.z + A.r1.x * B.r1.z + A.r2.x * B.r2.z);
  R.r1 = mk<T>(A.r0.y * B.r0.x + A.r1.y * B.r1.x + A.r2.y * B.r2.x, A.r0.y * B.r0.y + A.r1.y * B.r1.y + A.r2.y * B.r2.y,
               A.r0.y * B.r0.z + A.r1.y * B.r1.z + A.r2.y * B.r2.z);
  R.r2 = mk<T>(A.r0.z * B.r0.x + A.r1.z * B.r1.x + A.r2.z * B.r2.x, A.r0.z * B.r0.y + A.r1.z * B.r1.y + A.r2.z * B.r2.y,
               A.r0.z * B.r0.z + A.r1.z * B.r1.z + A.r2.z * B.r2.z);
  return R;
}
template <typename T> HFCL_HD M3<T> mmul(const M3<T>& A, const M3<T>& B) {  // A * B
  M3<T> R;
  R.r0 = mk<T>(A.r0.x * B.r0.x + A.r0.y * B.r1.x + A.r0.z * B.r2.x, A.r0.x * B.r0.y + A.r0.y * B.r1.y + A.r0.z * B.r2.y,
               A.r0.x * B.r0.z + A.r0.y * B.r1.z + A.r0.z * B.r2.z);
  R.r1 = mk<T>(A.r1.x * B.r0.x + A.r1.y * B.r1.x + A.r1.z * B.r2.x, A.r1.x * B.r0.y + A.r1.y * B.r1.y + A.r1.z * B.r2.y,
               A.r1.x * B.r0.z + A.r1.y * B.r1.z + A.r1.z * B.r2.z);
  R.r2 = mk<T>(A.r2.x * B.r0.x + A.r2.y * B.r1.x + A.r2.z * B.r2.x, A.r2.x * B.r0.y + A.r2.y * B.r1.y + A.r2.z * B.r2.y,
               A.r2.x * B.r0.z + A.r2.y * B.r1.z + A.r2.z * B.r2.z);
  return R;
}

// Pose {R, T}
template <typename T>
struct Pose {
  M3<T> R;
  V3<T> t;
};
template <typename T> HFCL_HD V3<T> xform(const Pose<T>& p, const V3<T>& v) { return mul(p.R, v) + p.t; }

// ABI pose (12 doubles: column-major R then T) -> Pose<T>
template <typename T> HFCL_HD Pose<T> pose_from_abi(const double* p) {
  Pose<T> r;
  r.R.r0 = mk<T>(T(p[0]), T(p[3]), T(p[6]));
  r.R.r1 = mk<T>(T(p[1]), T(p[4]), T(p[7]));
  r.R.r2 = mk<T>(T(p[2]), T(p[5]), T(p[8]));
  r.t = mk<T>(T(p[9]), T(p[10]), T(p[11]));
  return r;
}
// compact fp32 pose: quaternion (w,x,y,z) + translation; Eigen Quaternion::toRotationMatrix order
template <typename T, typename S = float> HFCL_HD Pose<T> pose_from_quat(const S* p) {
  T w = T(p[0]), x = T(p[1]), y = T(p[2]), z = T(p[3]);
  T tx = T(2) * x, ty = T(2) * y, tz = T(2) * z;
  T twx = tx * w, twy = ty * w, twz = tz * w;
  T txx = tx * x, txy = ty * x, txz = tz * x;
  T tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Pose<T> r;
  r.R.r0 = mk<T>(T(1) - (tyy + tzz), txy - twz, txz + twy);
  r.R.r1 = mk<T>(txy + twz, T(1) - (txx + tzz), tyz - twx);
  r.R.r2 = mk<T>(txz - twy, tyz + twx, T(1) - (txx + tyy));
  r.t = mk<T>(T(p[4]), T(p[5]), T(p[6]));
  return r;
}

// Eigen isIdentity / isZero with dummy precision (used by MinkowskiDiff::set)
template <typename T> HFCL_HD bool is_identity(const M3<T>& A) {
  const T p = Lim<T>::tiny();
  bool ok = true;
  const T d[3] = {A.r0.x, A.r1.y, A.r2.z};
  for (int i = 0; i < 3; ++i) {
    T mn = habs(d[i]) < T(1) ? habs(d[i]) : T(1);
    ok = ok && (habs(d[i] - T(1)) <= mn * p);
  }
  ok = ok && habs(A.r0.y) <= p && habs(A.r0.z) <= p && habs(A.r1.x) <= p && habs(A.r1.z) <= p && habs(A.r2.x) <= p &&
       habs(A.r2.y) <= p;
  return ok;
}
template <typename T> HFCL_HD bool is_zero(const V3<T>& a) {
  const T p = Lim<T>::tiny();
  return habs(a.x) <= p && habs(a.y) <= p && habs(a.z) <= p;
}

}  // namespace hfcl
