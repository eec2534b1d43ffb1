// Closed-form SO3 / SE3 group math, host + device, float or double.
//
// Restates the formulas of the reference's lietorch (csrc/lietorch_ext/so3.h, se3.h) without Eigen:
// quaternion (x,y,z,w) + translation, data row [tx,ty,tz,qx,qy,qz,qw], tangent [tau, phi],
// quaternions re-normalised on load and after products (so3.h:36-38), EPS = 1e-6 (common.h:13).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define LIE_HD __host__ __device__ __forceinline__
#define LIE_EPS 1e-6

namespace lie {

template <typename S>
struct Vec3 {
  S x, y, z;
};
template <typename S>
LIE_HD Vec3<S> operator+(Vec3<S> a, Vec3<S> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename S>
LIE_HD Vec3<S> operator-(Vec3<S> a, Vec3<S> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename S>
LIE_HD Vec3<S> operator*(S s, Vec3<S> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename S>
LIE_HD Vec3<S> cross(Vec3<S> a, Vec3<S> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename S>
LIE_HD S dot(Vec3<S> a, Vec3<S> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

template <typename S>
struct Mat3 {
  S m[3][3];
  LIE_HD Vec3<S> operator*(Vec3<S> v) const {
    return {m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
            m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
  }
  LIE_HD Vec3<S> tmul(Vec3<S> v) const {  // M^T v
    return {m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z, m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
            m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z};
  }
};

template <typename S>
struct Quat {
  S x, y, z, w;
  LIE_HD Vec3<S> vec() const { return {x, y, z}; }
};

template <typename S>
LIE_HD Quat<S> normalized(Quat<S> q) {
  S n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  S inv = S(1) / n;
  return {q.x * inv, q.y * inv, q.z * inv, q.w * inv};
}

template <typename S>
struct SO3 {
  static constexpr int K = 3, N = 4;
  Quat<S> q;
  LIE_HD SO3() : q{0, 0, 0, 1} {}
  LIE_HD explicit SO3(Quat<S> q_) : q(normalized(q_)) {}
  LIE_HD explicit SO3(const S* d) : q(normalized(Quat<S>{d[0], d[1], d[2], d[3]})) {}
  LIE_HD void store(S* d) const { d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w; }
  LIE_HD SO3 inv() const { return SO3(Quat<S>{-q.x, -q.y, -q.z, q.w}); }
  LIE_HD SO3 operator*(const SO3& o) const {
    const Quat<S>&a = q, &b = o.q;
    return SO3(Quat<S>{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                       a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z});
  }
  LIE_HD Vec3<S> act(Vec3<S> p) const {  // so3.h:50-55
    Vec3<S> uv = cross(q.vec(), p);
    uv = uv + uv;
    return p + q.w * uv + cross(q.vec(), uv);
  }
  LIE_HD Mat3<S> matrix() const {  // Eigen toRotationMatrix
    S tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    S twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    S txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    S tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    Mat3<S> R;
    R.m[0][0] = 1 - (tyy + tzz); R.m[0][1] = txy - twz; R.m[0][2] = txz + twy;
    R.m[1][0] = txy + twz; R.m[1][1] = 1 - (txx + tzz); R.m[1][2] = tyz - twx;
    R.m[2][0] = txz - twy; R.m[2][1] = tyz + twx; R.m[2][2] = 1 - (txx + tyy);
    return R;
  }
  LIE_HD Vec3<S> log() const {  // so3.h:96-131
    S sn = q.x * q.x + q.y * q.y + q.z * q.z;
    S w = q.w, f;
    if (sn < S(LIE_EPS * LIE_EPS)) {
      f = S(2) / w - S(2.0 / 3.0) * sn / (w * w * w);
    } else {
      S n = sqrt(sn);
      if (fabs(w) < S(LIE_EPS)) f = (w > S(0) ? S(M_PI) : -S(M_PI)) / n;
      else f = S(2) * atan(n / w) / n;
    }
    return f * q.vec();
  }
  static LIE_HD SO3 exp(Vec3<S> phi) {  // so3.h:133-151
    S t2 = dot(phi, phi), t = sqrt(t2), im, re;
    if (t < S(LIE_EPS)) {
      S t4 = t2 * t2;
      im = S(0.5) - S(1.0 / 48.0) * t2 + S(1.0 / 3840.0) * t4;
      re = S(1) - S(1.0 / 8.0) * t2 + S(1.0 / 384.0) * t4;
    } else {
      im = sin(S(0.5) * t) / t;
      re = cos(S(0.5) * t);
    }
    return SO3(Quat<S>{im * phi.x, im * phi.y, im * phi.z, re});
  }
};

// I + c1 [phi]x + c2 [phi]x^2 applied to v
template <typename S>
LIE_HD Vec3<S> apply_poly(S c1, S c2, Vec3<S> phi, Vec3<S> v) {
  Vec3<S> pv = cross(phi, v);
  Vec3<S> ppv = cross(phi, pv);
  return v + c1 * pv + c2 * ppv;
}
template <typename S>
LIE_HD Vec3<S> left_jacobian_mul(Vec3<S> phi, Vec3<S> v) {  // so3.h:153-168
  S t2 = dot(phi, phi), t = sqrt(t2);
  S c1 = (t < S(LIE_EPS)) ? S(0.5) - S(1.0 / 24.0) * t2 : (S(1) - cos(t)) / t2;
  S c2 = (t < S(LIE_EPS)) ? S(1.0 / 6.0) - S(1.0 / 120.0) * t2 : (t - sin(t)) / (t2 * t);
  return apply_poly(c1, c2, phi, v);
}
template <typename S>
LIE_HD Vec3<S> left_jacobian_inv_mul(Vec3<S> phi, Vec3<S> v) {  // so3.h:170-184
  S t2 = dot(phi, phi), t = sqrt(t2), h = S(0.5) * t;
  S c2 = (t < S(LIE_EPS)) ? S(1.0 / 12.0) : (S(1) - t * cos(h) / (S(2) * sin(h))) / (t * t);
  return apply_poly(S(-0.5), c2, phi, v);
}

template <typename S>
struct SE3 {
  static constexpr int K = 6, N = 7;
  Vec3<S> t;
  SO3<S> r;
  LIE_HD SE3() : t{0, 0, 0}, r() {}
  LIE_HD SE3(SO3<S> r_, Vec3<S> t_) : t(t_), r(r_) {}
  LIE_HD explicit SE3(const S* d) : t{d[0], d[1], d[2]}, r(d + 3) {}
  LIE_HD void store(S* d) const { d[0] = t.x; d[1] = t.y; d[2] = t.z; r.store(d + 3); }
  LIE_HD SE3 inv() const { SO3<S> ri = r.inv(); return SE3(ri, S(-1) * ri.act(t)); }  // se3.h:40
  LIE_HD SE3 operator*(const SE3& o) const { return SE3(r * o.r, t + r.act(o.t)); }    // se3.h:48-50
  LIE_HD Vec3<S> act(Vec3<S> p) const { return r.act(p) + t; }
  LIE_HD void act4(const S* p, S* o) const {  // se3.h:54-58
    Vec3<S> v = r.act(Vec3<S>{p[0], p[1], p[2]}) + p[3] * t;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = p[3];
  }
  // Adj = [[R, t^R],[0,R]]  (se3.h:60-69)
  LIE_HD void adj(const S* a, S* b) const {
    Mat3<S> R = r.matrix();
    Vec3<S> Ra{a[0], a[1], a[2]}, Rb{a[3], a[4], a[5]};
    Vec3<S> u = R * Ra, w = R * Rb;
    Vec3<S> top = u + cross(t, w);
    b[0] = top.x; b[1] = top.y; b[2] = top.z; b[3] = w.x; b[4] = w.y; b[5] = w.z;
  }
  // Adj^T a = [R^T a1, (t^R)^T a1 + R^T a2] = [R^T a1, R^T (a1 x t) ... ] with (t^)^T = -t^
  LIE_HD void adjT(const S* a, S* b) const {
    Mat3<S> R = r.matrix();
    Vec3<S> a1{a[0], a[1], a[2]}, a2{a[3], a[4], a[5]};
    Vec3<S> o1 = R.tmul(a1);
    Vec3<S> o2 = R.tmul(a2 - cross(t, a1));
    b[0] = o1.x; b[1] = o1.y; b[2] = o1.z; b[3] = o2.x; b[4] = o2.y; b[5] = o2.z;
  }
  LIE_HD void log(S* a) const {  // se3.h:117-125
    Vec3<S> phi = r.log();
    Vec3<S> tau = left_jacobian_inv_mul(phi, t);
    a[0] = tau.x; a[1] = tau.y; a[2] = tau.z; a[3] = phi.x; a[4] = phi.y; a[5] = phi.z;
  }
  static LIE_HD SE3 exp(const S* a) {  // se3.h:127-136
    Vec3<S> tau{a[0], a[1], a[2]}, phi{a[3], a[4], a[5]};
    return SE3(SO3<S>::exp(phi), left_jacobian_mul(phi, tau));
  }
};

}  // namespace lie
