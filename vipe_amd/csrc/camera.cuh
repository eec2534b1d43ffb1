// Camera models of the dense BA (pinhole, MEI), per-pixel, with the Jacobians the solver needs.
// Restates vipe/utils/cameras.py:131-207 (pinhole) and :228-336 (MEI); MIN_DEPTH = 0.1 (cameras.py:48).
#pragma once
#include "common.cuh"

namespace cam {

constexpr float MIN_DEPTH = 0.1f;

struct Intr {
  float fx, fy, cx, cy, k1;
};

// intrinsics row [4+D] at full resolution -> scaled by `s` (cameras.py:212-213, :345-348: k1 is not scaled)
__host__ __device__ inline Intr load_scaled(const float* p, int D, float s) {
  Intr I;
  I.fx = p[0] * s;
  I.fy = p[1] * s;
  I.cx = p[2] * s;
  I.cy = p[3] * s;
  I.k1 = D > 0 ? p[4] : 0.0f;
  return I;
}

// inverse projection: X0 = (X, Y, 1, d).  dXf[f] = d(X,Y)/d(intrinsic f), f = 0 focal, f = 1 k1 (MEI)
template <int CAM, int F>
__device__ __forceinline__ void iproj(const Intr& I, float u, float v, float& X, float& Y, float (&dX)[F > 0 ? F : 1],
                                      float (&dY)[F > 0 ? F : 1]) {
  if constexpr (CAM == VIPE_CAM_PINHOLE) {
    X = (u - I.cx) / I.fx;
    Y = (v - I.cy) / I.fy;
    if constexpr (F > 0) {
      dX[0] = -X / I.fx;
      dY[0] = -Y / I.fy;
    }
  } else {
    const float k1 = I.k1;
    const float ub = (u - I.cx) / I.fx, vb = (v - I.cy) / I.fy;
    const float r2 = ub * ub + vb * vb;
    const float q = sqrtf(1.0f + (1.0f - k1 * k1) * r2);
    const float factor = (k1 + q) / (1.0f + r2);
    X = ub * factor / (factor - k1);
    Y = vb * factor / (factor - k1);
    if constexpr (F > 0) {
      const float k2 = k1 * k1, k3 = k2 * k1, q2 = q * q, r4 = r2 * r2;
      const float f_num = -k3 * r4 - k3 * r2 - k2 * q * r2 - k1 * q2 * r2 - k1 * q2 + k1 * r4 + k1 * r2 - q2 * q;
      const float f_den = I.fx * q * (k2 * r4 - 2.0f * k1 * q * r2 + q2);
      dX[0] = ub * f_num / f_den;
      dY[0] = vb * f_num / f_den;
      if constexpr (F > 1) {
        const float tt = -k1 * (r2 + 1.0f) + k1 + q;
        const float k_num = (k1 + q) * (k1 * r2 + q * (r2 + 1.0f) - q) - (k1 * r2 - q) * tt;
        const float k_den = q * tt * tt;
        dX[1] = ub * k_num / k_den;
        dY[1] = vb * k_num / k_den;
      }
    }
  }
}

// projection of X1 = (X,Y,Z) with the z < MIN_DEPTH -> 1 clamp (cameras.py:175-177, 297-298).
// Jp = d(x,y)/d(X,Y,Z) (2x3; the 4th column is zero), Jf = d(x,y)/d(intrinsic f) at the TARGET view.
template <int CAM, bool JAC, int F>
__device__ __forceinline__ void proj(const Intr& I, float X, float Y, float Zin, float& x, float& y,
                                     float (&Jp)[2][3], float (&Jf)[2][F > 0 ? F : 1]) {
  const float Z = Zin < MIN_DEPTH ? 1.0f : Zin;
  if constexpr (CAM == VIPE_CAM_PINHOLE) {
    const float d = 1.0f / Z;
    x = I.fx * (X * d) + I.cx;
    y = I.fy * (Y * d) + I.cy;
    if constexpr (JAC) {
      Jp[0][0] = I.fx * d; Jp[0][1] = 0.0f; Jp[0][2] = -I.fx * X * d * d;
      Jp[1][0] = 0.0f; Jp[1][1] = I.fy * d; Jp[1][2] = -I.fy * Y * d * d;
    }
    if constexpr (F > 0) {
      Jf[0][0] = X * d;
      Jf[1][0] = Y * d;
    }
  } else {
    const float k1 = I.k1;
    const float r = sqrtf(X * X + Y * Y + Z * Z);
    const float rbase = Z + k1 * r;
    const float d = 1.0f / rbase;
    x = I.fx * (X * d) + I.cx;
    y = I.fy * (Y * d) + I.cy;
    if constexpr (JAC) {
      const float rd = rbase * rbase * r;
      Jp[0][0] = I.fx * (-k1 * X * X + rbase * r) / rd;
      Jp[0][1] = -I.fx * k1 * X * Y / rd;
      Jp[0][2] = -I.fx * X * (k1 * Z + r) / rd;
      Jp[1][0] = -I.fy * k1 * X * Y / rd;
      Jp[1][1] = I.fy * (-k1 * Y * Y + rbase * r) / rd;
      Jp[1][2] = -I.fy * Y * (k1 * Z + r) / rd;
    }
    if constexpr (F > 0) {
      Jf[0][0] = X * d;
      Jf[1][0] = Y * d;
      if constexpr (F > 1) {
        Jf[0][1] = -I.fx * r * X * d * d;
        Jf[1][1] = -I.fy * r * Y * d * d;
      }
    }
  }
}

}  // namespace cam
