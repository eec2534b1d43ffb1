// Closed-form SO3 / RxSO3 / SE3 / Sim3 group math, host + device, float or double.
//
// Restates the formulas of the reference's lietorch (csrc/lietorch_ext/so3.h, rxso3.h, se3.h, sim3.h) without Eigen:
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


// ---- Jacobian blocks for the backward passes (so3.h:153-202, se3.h:60-214).  Small dense matrices as S[R][C].
template <typename S>
LIE_HD void hat3(Vec3<S> v, S (&M)[3][3]) {
  M[0][0] = 0; M[0][1] = -v.z; M[0][2] = v.y;
  M[1][0] = v.z; M[1][1] = 0; M[1][2] = -v.x;
  M[2][0] = -v.y; M[2][1] = v.x; M[2][2] = 0;
}
template <typename S>
LIE_HD void mat3_mul(const S (&A)[3][3], const S (&B)[3][3], S (&C)[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
}
// I + c1 Phi + c2 Phi^2
template <typename S>
LIE_HD void poly3(S c1, S c2, Vec3<S> phi, S (&J)[3][3]) {
  S P[3][3], P2[3][3];
  hat3(phi, P);
  mat3_mul(P, P, P2);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) J[i][j] = (i == j ? S(1) : S(0)) + c1 * P[i][j] + c2 * P2[i][j];
}
template <typename S>
LIE_HD void so3_left_jacobian(Vec3<S> phi, S (&J)[3][3]) {  // so3.h:153-168
  S t2 = dot(phi, phi), t = sqrt(t2);
  S c1 = (t < S(LIE_EPS)) ? S(0.5) - S(1.0 / 24.0) * t2 : (S(1) - cos(t)) / t2;
  S c2 = (t < S(LIE_EPS)) ? S(1.0 / 6.0) - S(1.0 / 120.0) * t2 : (t - sin(t)) / (t2 * t);
  poly3(c1, c2, phi, J);
}
template <typename S>
LIE_HD void so3_left_jacobian_inverse(Vec3<S> phi, S (&J)[3][3]) {  // so3.h:170-184
  S t2 = dot(phi, phi), t = sqrt(t2), h = S(0.5) * t;
  S c2 = (t < S(LIE_EPS)) ? S(1.0 / 12.0) : (S(1) - t * cos(h) / (S(2) * sin(h))) / (t * t);
  poly3(S(-0.5), c2, phi, J);
}
template <typename S>
LIE_HD void se3_calcQ(const S* a, S (&Q)[3][3]) {  // se3.h:138-163
  Vec3<S> tau{a[0], a[1], a[2]}, phi{a[3], a[4], a[5]};
  S T[3][3], P[3][3], PT[3][3], TP[3][3], PTP[3][3], PP[3][3], PPT[3][3], TPP[3][3], PTPP[3][3], PPTP[3][3];
  hat3(tau, T);
  hat3(phi, P);
  mat3_mul(P, T, PT); mat3_mul(T, P, TP); mat3_mul(PT, P, PTP); mat3_mul(P, P, PP);
  mat3_mul(PP, T, PPT); mat3_mul(T, PP, TPP); mat3_mul(PTP, P, PTPP); mat3_mul(PP, TP, PPTP);
  S t = sqrt(dot(phi, phi)), t2 = t * t, t4 = t2 * t2;
  S c1 = (t < S(LIE_EPS)) ? S(1.0 / 6.0) - S(1.0 / 120.0) * t2 : (t - sin(t)) / (t2 * t);
  S c2 = (t < S(LIE_EPS)) ? S(1.0 / 24.0) - S(1.0 / 720.0) * t2 : (t2 + 2 * cos(t) - 2) / (2 * t4);
  S c3 = (t < S(LIE_EPS)) ? S(1.0 / 120.0) - S(1.0 / 2520.0) * t2 : (2 * t - 3 * sin(t) + t * cos(t)) / (2 * t4 * t);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      Q[i][j] = S(0.5) * T[i][j] + c1 * (PT[i][j] + TP[i][j] + PTP[i][j]) +
                c2 * (PPT[i][j] + TPP[i][j] - 3 * PTP[i][j]) + c3 * (PTPP[i][j] + PPTP[i][j]);
}

// ---- RxSO3 (rotation x positive scale; rxso3.h) and Sim3 (sim3.h).  Rows [qx,qy,qz,qw,s] / [t, q, s];
// tangents [phi, sigma] / [tau, phi, sigma].
template <typename S>
LIE_HD void mat3_inverse(const S (&A)[3][3], S (&B)[3][3]) {  // cofactor inverse (sim3.h:147 `W.inverse()`)
  S c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2],
    c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
  S inv = S(1) / (A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02);
  B[0][0] = c00 * inv; B[1][0] = c01 * inv; B[2][0] = c02 * inv;
  B[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * inv;
  B[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * inv;
  B[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * inv;
  B[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * inv;
  B[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * inv;
  B[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * inv;
}

// W(phi, sigma) = A Phi + B Phi^2 + C I, the translation part of the Sim3 exponential (rxso3.h:183-224)
template <typename S>
LIE_HD void rxso3_calcW(const S* ps, S (&W)[3][3]) {
  Vec3<S> phi{ps[0], ps[1], ps[2]};
  const S sigma = ps[3], theta = sqrt(dot(phi, phi)), scale = exp(sigma), one(1), half(0.5);
  S A, B, C;
  if (fabs(sigma) < S(LIE_EPS)) {
    C = one;
    if (fabs(theta) < S(LIE_EPS)) {
      A = half;
      B = S(1.0 / 6.0);
    } else {
      S t2 = theta * theta;
      A = (one - cos(theta)) / t2;
      B = (theta - sin(theta)) / (t2 * theta);
    }
  } else {
    C = (scale - one) / sigma;
    if (fabs(theta) < S(LIE_EPS)) {
      S s2 = sigma * sigma;
      A = ((sigma - one) * scale + one) / s2;
      B = (scale * half * s2 + scale - one - sigma * scale) / (s2 * sigma);
    } else {
      S t2 = theta * theta, a = scale * sin(theta), b = scale * cos(theta), c = t2 + sigma * sigma;
      A = (a * sigma + (one - b) * theta) / (theta * c);
      B = (C - ((b - one) * sigma + a * theta) / c) * one / t2;
    }
  }
  S P[3][3], P2[3][3];
  hat3(phi, P);
  mat3_mul(P, P, P2);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) W[i][j] = A * P[i][j] + B * P2[i][j] + (i == j ? C : S(0));
}

template <typename S>
struct RxSO3 {
  static constexpr int K = 4, N = 5;
  SO3<S> r;
  S s;
  LIE_HD RxSO3() : r(), s(1) {}
  LIE_HD RxSO3(SO3<S> r_, S s_) : r(r_), s(s_) {}
  LIE_HD explicit RxSO3(const S* d) : r(d), s(d[4]) {}
  LIE_HD void store(S* d) const { r.store(d); d[4] = s; }
  LIE_HD RxSO3 inv() const { return RxSO3(r.inv(), S(1) / s); }                     // rxso3.h:50
  LIE_HD RxSO3 operator*(const RxSO3& o) const { return RxSO3(r * o.r, s * o.s); }  // rxso3.h:58-60
  LIE_HD Vec3<S> act(Vec3<S> p) const { return s * r.act(p); }                      // rxso3.h:62-67
  LIE_HD void act4(const S* p, S* o) const {
    Vec3<S> v = act(Vec3<S>{p[0], p[1], p[2]});
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = p[3];
  }
  LIE_HD void log(S* a) const {  // rxso3.h:127-163
    Vec3<S> phi = r.log();
    a[0] = phi.x; a[1] = phi.y; a[2] = phi.z; a[3] = ::log(s);
  }
  static LIE_HD RxSO3 exp(const S* a) {  // rxso3.h:165-181
    return RxSO3(SO3<S>::exp(Vec3<S>{a[0], a[1], a[2]}), ::exp(a[3]));
  }
};

template <typename S>
struct Sim3 {
  static constexpr int K = 7, N = 8;
  Vec3<S> t;
  RxSO3<S> r;
  LIE_HD Sim3() : t{0, 0, 0}, r() {}
  LIE_HD Sim3(RxSO3<S> r_, Vec3<S> t_) : t(t_), r(r_) {}
  LIE_HD explicit Sim3(const S* d) : t{d[0], d[1], d[2]}, r(d + 3) {}
  LIE_HD void store(S* d) const { d[0] = t.x; d[1] = t.y; d[2] = t.z; r.store(d + 3); }
  LIE_HD Sim3 inv() const { RxSO3<S> ri = r.inv(); return Sim3(ri, S(-1) * ri.act(t)); }  // sim3.h:43
  LIE_HD Sim3 operator*(const Sim3& o) const { return Sim3(r * o.r, t + r.act(o.t)); }     // sim3.h:51-53
  LIE_HD Vec3<S> act(Vec3<S> p) const { return r.act(p) + t; }
  LIE_HD void act4(const S* p, S* o) const {  // sim3.h:57-61
    Vec3<S> v = r.act(Vec3<S>{p[0], p[1], p[2]}) + p[3] * t;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = p[3];
  }
  LIE_HD void log(S* a) const {  // sim3.h:143-152
    S ps[4], W[3][3], Wi[3][3];
    r.log(ps);
    rxso3_calcW(ps, W);
    mat3_inverse(W, Wi);
    a[0] = Wi[0][0] * t.x + Wi[0][1] * t.y + Wi[0][2] * t.z;
    a[1] = Wi[1][0] * t.x + Wi[1][1] * t.y + Wi[1][2] * t.z;
    a[2] = Wi[2][0] * t.x + Wi[2][1] * t.y + Wi[2][2] * t.z;
    a[3] = ps[0]; a[4] = ps[1]; a[5] = ps[2]; a[6] = ps[3];
  }
  static LIE_HD Sim3 exp(const S* a) {  // sim3.h:154-163
    S W[3][3];
    rxso3_calcW(a + 3, W);
    Vec3<S> tt{W[0][0] * a[0] + W[0][1] * a[1] + W[0][2] * a[2], W[1][0] * a[0] + W[1][1] * a[1] + W[1][2] * a[2],
               W[2][0] * a[0] + W[2][1] * a[1] + W[2][2] * a[2]};
    return Sim3(RxSO3<S>::exp(a + 3), tt);
  }
};

// Per-group Jacobian providers.  K x K matrices are S[K][K]; everything the reference's backward kernels use
// (lietorch_gpu.cu:36-275): Adj(), adj(b), left_jacobian(a), left_jacobian_inverse(a), act / act4 Jacobians,
// the 4x4 matrix and the orthogonal projector.
template <typename G, typename S>
struct Jac;

template <typename S>
struct Jac<SO3<S>, S> {
  static constexpr int K = 3, N = 4;
  static LIE_HD void Adj(const SO3<S>& X, S (&A)[3][3]) {
    Mat3<S> R = X.matrix();
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) A[i][j] = R.m[i][j];
  }
  static LIE_HD void adj(const S* b, S (&A)[3][3]) { hat3(Vec3<S>{b[0], b[1], b[2]}, A); }
  static LIE_HD void left_jacobian(const S* a, S (&J)[3][3]) { so3_left_jacobian(Vec3<S>{a[0], a[1], a[2]}, J); }
  static LIE_HD void left_jacobian_inverse(const S* a, S (&J)[3][3]) {
    so3_left_jacobian_inverse(Vec3<S>{a[0], a[1], a[2]}, J);
  }
  static LIE_HD void log(const SO3<S>& X, S* a) { Vec3<S> v = X.log(); a[0] = v.x; a[1] = v.y; a[2] = v.z; }
  static LIE_HD void act_jacobian(Vec3<S> q, S (&J)[3][3]) { hat3(S(-1) * q, J); }
  static LIE_HD void act4_jacobian(const S* q, S (&J)[4][3]) {
    S H[3][3];
    hat3(Vec3<S>{-q[0], -q[1], -q[2]}, H);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) J[i][j] = H[i][j];
    J[3][0] = J[3][1] = J[3][2] = 0;
  }
  static LIE_HD void matrix4(const SO3<S>& X, S (&T)[4][4]) {
    Mat3<S> R = X.matrix();
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) T[i][j] = (i < 3 && j < 3) ? R.m[i][j] : (i == j ? S(1) : S(0));
  }
  static LIE_HD void projector(const SO3<S>& X, S (&P)[4][4]) {  // so3.h:72-80
    S H[3][3];
    hat3(Vec3<S>{-X.q.x, -X.q.y, -X.q.z}, H);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) P[i][j] = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) P[i][j] = S(0.5) * ((i == j ? X.q.w : S(0)) + H[i][j]);
    P[3][0] = S(-0.5) * X.q.x; P[3][1] = S(-0.5) * X.q.y; P[3][2] = S(-0.5) * X.q.z;
  }
};

template <typename S>
struct Jac<SE3<S>, S> {
  static constexpr int K = 6, N = 7;
  static LIE_HD void Adj(const SE3<S>& X, S (&A)[6][6]) {  // [[R, t^R],[0,R]]
    Mat3<S> R = X.r.matrix();
    S Rm[3][3], Tx[3][3], TR[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Rm[i][j] = R.m[i][j];
    hat3(X.t, Tx);
    mat3_mul(Tx, Rm, TR);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        A[i][j] = Rm[i][j]; A[i][j + 3] = TR[i][j]; A[i + 3][j] = 0; A[i + 3][j + 3] = Rm[i][j];
      }
  }
  static LIE_HD void adj(const S* b, S (&A)[6][6]) {  // [[Phi, Tau],[0,Phi]]
    S T[3][3], P[3][3];
    hat3(Vec3<S>{b[0], b[1], b[2]}, T);
    hat3(Vec3<S>{b[3], b[4], b[5]}, P);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) { A[i][j] = P[i][j]; A[i][j + 3] = T[i][j]; A[i + 3][j] = 0; A[i + 3][j + 3] = P[i][j]; }
  }
  static LIE_HD void left_jacobian(const S* a, S (&J6)[6][6]) {  // [[J, Q],[0,J]]
    S J[3][3], Q[3][3];
    so3_left_jacobian(Vec3<S>{a[3], a[4], a[5]}, J);
    se3_calcQ(a, Q);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) { J6[i][j] = J[i][j]; J6[i][j + 3] = Q[i][j]; J6[i + 3][j] = 0; J6[i + 3][j + 3] = J[i][j]; }
  }
  static LIE_HD void left_jacobian_inverse(const S* a, S (&J6)[6][6]) {  // [[Ji, -Ji Q Ji],[0,Ji]]
    S Ji[3][3], Q[3][3], A1[3][3], A2[3][3];
    so3_left_jacobian_inverse(Vec3<S>{a[3], a[4], a[5]}, Ji);
    se3_calcQ(a, Q);
    mat3_mul(Ji, Q, A1);
    mat3_mul(A1, Ji, A2);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) { J6[i][j] = Ji[i][j]; J6[i][j + 3] = -A2[i][j]; J6[i + 3][j] = 0; J6[i + 3][j + 3] = Ji[i][j]; }
  }
  static LIE_HD void log(const SE3<S>& X, S* a) { X.log(a); }
  static LIE_HD void act_jacobian(Vec3<S> q, S (&J)[3][6]) {  // [I | hat(-q)]
    S H[3][3];
    hat3(S(-1) * q, H);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) { J[i][j] = (i == j ? S(1) : S(0)); J[i][j + 3] = H[i][j]; }
  }
  static LIE_HD void act4_jacobian(const S* q, S (&J)[4][6]) {  // [[q3 I, hat(-q)],[0,0]]
    S H[3][3];
    hat3(Vec3<S>{-q[0], -q[1], -q[2]}, H);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) { J[i][j] = (i == j ? q[3] : S(0)); J[i][j + 3] = H[i][j]; }
    for (int j = 0; j < 6; ++j) J[3][j] = 0;
  }
  static LIE_HD void matrix4(const SE3<S>& X, S (&T)[4][4]) {
    Mat3<S> R = X.r.matrix();
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[i][j] = R.m[i][j]; }
    T[0][3] = X.t.x; T[1][3] = X.t.y; T[2][3] = X.t.z;
    T[3][0] = T[3][1] = T[3][2] = 0; T[3][3] = 1;
  }
  static LIE_HD void projector(const SE3<S>& X, S (&P)[7][7]) {  // se3.h:107-115
    S H[3][3], P4[4][4];
    hat3(S(-1) * X.t, H);
    Jac<SO3<S>, S>::projector(X.r, P4);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) P[i][j] = 0;
    for (int i = 0; i < 3; ++i) { P[i][i] = 1; for (int j = 0; j < 3; ++j) P[i][j + 3] = H[i][j]; }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) P[i + 3][j + 3] = P4[i][j];
  }
};

template <typename S>
struct Jac<RxSO3<S>, S> {
  static constexpr int K = 4, N = 5;
  static LIE_HD void Adj(const RxSO3<S>& X, S (&A)[4][4]) {  // blockdiag(R, 1)  (rxso3.h:75-79)
    Mat3<S> R = X.r.matrix();
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) A[i][j] = (i < 3 && j < 3) ? R.m[i][j] : (i == j ? S(1) : S(0));
  }
  static LIE_HD void adj(const S* b, S (&A)[4][4]) {  // rxso3.h:116-125
    S P[3][3];
    hat3(Vec3<S>{b[0], b[1], b[2]}, P);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) A[i][j] = (i < 3 && j < 3) ? P[i][j] : S(0);
  }
  static LIE_HD void left_jacobian(const S* a, S (&J)[4][4]) {  // rxso3.h:278-284
    S J3[3][3];
    so3_left_jacobian(Vec3<S>{a[0], a[1], a[2]}, J3);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) J[i][j] = (i < 3 && j < 3) ? J3[i][j] : (i == j ? S(1) : S(0));
  }
  static LIE_HD void left_jacobian_inverse(const S* a, S (&J)[4][4]) {  // rxso3.h:286-292
    S J3[3][3];
    so3_left_jacobian_inverse(Vec3<S>{a[0], a[1], a[2]}, J3);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) J[i][j] = (i < 3 && j < 3) ? J3[i][j] : (i == j ? S(1) : S(0));
  }
  static LIE_HD void log(const RxSO3<S>& X, S* a) { X.log(a); }
  static LIE_HD void act_jacobian(Vec3<S> q, S (&J)[3][4]) {  // [hat(-q) | q]  (rxso3.h:294-299)
    S H[3][3];
    hat3(S(-1) * q, H);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) J[i][j] = H[i][j]; }
    J[0][3] = q.x; J[1][3] = q.y; J[2][3] = q.z;
  }
  static LIE_HD void act4_jacobian(const S* q, S (&J)[4][4]) {  // rxso3.h:301-307
    S H[3][3];
    hat3(Vec3<S>{-q[0], -q[1], -q[2]}, H);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) J[i][j] = H[i][j]; J[i][3] = q[i]; }
    for (int j = 0; j < 4; ++j) J[3][j] = 0;
  }
  static LIE_HD void matrix4(const RxSO3<S>& X, S (&T)[4][4]) {  // blockdiag(s R, 1)
    Mat3<S> R = X.r.matrix();
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) T[i][j] = (i < 3 && j < 3) ? X.s * R.m[i][j] : (i == j ? S(1) : S(0));
  }
  static LIE_HD void projector(const RxSO3<S>& X, S (&P)[5][5]) {  // rxso3.h:92-106
    S P4[4][4];
    Jac<SO3<S>, S>::projector(X.r, P4);
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 5; ++j) P[i][j] = (i < 4 && j < 4) ? P4[i][j] : S(0);
    P[4][3] = X.s;
  }
};

template <typename S>
struct Jac<Sim3<S>, S> {
  static constexpr int K = 7, N = 8;
  static LIE_HD void Adj(const Sim3<S>& X, S (&A)[7][7]) {  // sim3.h:86-99
    Mat3<S> R = X.r.r.matrix();
    S Rm[3][3], Tx[3][3], TR[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Rm[i][j] = R.m[i][j];
    hat3(X.t, Tx);
    mat3_mul(Tx, Rm, TR);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) A[i][j] = (i == j ? S(1) : S(0));
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) { A[i][j] = X.r.s * Rm[i][j]; A[i][j + 3] = TR[i][j]; A[i + 3][j + 3] = Rm[i][j]; }
    A[0][6] = -X.t.x; A[1][6] = -X.t.y; A[2][6] = -X.t.z;
  }
  static LIE_HD void adj(const S* b, S (&A)[7][7]) {  // sim3.h:122-141
    S T[3][3], P[3][3];
    hat3(Vec3<S>{b[0], b[1], b[2]}, T);
    hat3(Vec3<S>{b[3], b[4], b[5]}, P);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) A[i][j] = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        A[i][j] = P[i][j] + (i == j ? b[6] : S(0)); A[i][j + 3] = T[i][j]; A[i + 3][j + 3] = P[i][j];
      }
    A[0][6] = -b[0]; A[1][6] = -b[1]; A[2][6] = -b[2];
  }
  static LIE_HD void mat7_mul(const S (&A)[7][7], const S (&B)[7][7], S (&C)[7][7]) {
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) {
        S acc = 0;
        for (int k = 0; k < 7; ++k) acc += A[i][k] * B[k][j];
        C[i][j] = acc;
      }
  }
  // The reference evaluates both Jacobians as TRUNCATED series in Xi = adj(a) (sim3.h:165-184; the 1/720 Xi^5
  // term of its left Jacobian sits after the `return ...;` and is dead code) - restated as written.
  static LIE_HD void left_jacobian(const S* a, S (&J)[7][7]) {
    S Xi[7][7], Xi2[7][7], Xi3[7][7], Xi4[7][7];
    adj(a, Xi);
    mat7_mul(Xi, Xi, Xi2);
    mat7_mul(Xi, Xi2, Xi3);
    mat7_mul(Xi2, Xi2, Xi4);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j)
        J[i][j] = (i == j ? S(1) : S(0)) + S(1.0 / 2.0) * Xi[i][j] + S(1.0 / 6.0) * Xi2[i][j] +
                  S(1.0 / 24.0) * Xi3[i][j] + S(1.0 / 120.0) * Xi4[i][j];
  }
  static LIE_HD void left_jacobian_inverse(const S* a, S (&J)[7][7]) {
    S Xi[7][7], Xi2[7][7], Xi4[7][7];
    adj(a, Xi);
    mat7_mul(Xi, Xi, Xi2);
    mat7_mul(Xi2, Xi2, Xi4);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j)
        J[i][j] = (i == j ? S(1) : S(0)) - S(1.0 / 2.0) * Xi[i][j] + S(1.0 / 12.0) * Xi2[i][j] -
                  S(1.0 / 720.0) * Xi4[i][j];
  }
  static LIE_HD void log(const Sim3<S>& X, S* a) { X.log(a); }
  static LIE_HD void act_jacobian(Vec3<S> q, S (&J)[3][7]) {  // [I | hat(-q) | q]  (sim3.h:186-193)
    S H[3][3];
    hat3(S(-1) * q, H);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) { J[i][j] = (i == j ? S(1) : S(0)); J[i][j + 3] = H[i][j]; }
    J[0][6] = q.x; J[1][6] = q.y; J[2][6] = q.z;
  }
  static LIE_HD void act4_jacobian(const S* q, S (&J)[4][7]) {  // sim3.h:195-202
    S H[3][3];
    hat3(Vec3<S>{-q[0], -q[1], -q[2]}, H);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) { J[i][j] = (i == j ? q[3] : S(0)); J[i][j + 3] = H[i][j]; }
      J[i][6] = q[i];
    }
    for (int j = 0; j < 7; ++j) J[3][j] = 0;
  }
  static LIE_HD void matrix4(const Sim3<S>& X, S (&T)[4][4]) {  // sim3.h:63-68
    Mat3<S> R = X.r.r.matrix();
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[i][j] = X.r.s * R.m[i][j]; }
    T[0][3] = X.t.x; T[1][3] = X.t.y; T[2][3] = X.t.z;
    T[3][0] = T[3][1] = T[3][2] = 0; T[3][3] = 1;
  }
  static LIE_HD void projector(const Sim3<S>& X, S (&P)[8][8]) {  // sim3.h:77-84
    S H[3][3], P5[5][5];
    hat3(S(-1) * X.t, H);
    Jac<RxSO3<S>, S>::projector(X.r, P5);
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j) P[i][j] = 0;
    for (int i = 0; i < 3; ++i) { P[i][i] = 1; for (int j = 0; j < 3; ++j) P[i][j + 3] = H[i][j]; }
    P[0][6] = X.t.x; P[1][6] = X.t.y; P[2][6] = X.t.z;
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 5; ++j) P[i + 3][j + 3] = P5[i][j];
  }
};

}  // namespace lie
