// Per-term rigid transforms shared by the reprojection and BA kernels.
// T = R_qj^-1 * G_pj * G_pi^-1 * R_qi  (vipe/slam/maths/geom.py:251-252), as rotation matrix + translation.
#pragma once
#include "camera.cuh"
#include "lie_math.h"

struct Rigid {
  float R[9];
  float t[3];
};

__device__ __forceinline__ Rigid to_rigid(const lie::SE3<float>& X) {
  Rigid o;
  lie::Mat3<float> M = X.r.matrix();
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) o.R[3 * i + j] = M.m[i][j];
  o.t[0] = X.t.x; o.t[1] = X.t.y; o.t[2] = X.t.z;
  return o;
}

struct TermGeom {
  Rigid T;    // full transform acting on X0
  Rigid G;    // G_ij = G_pj * G_pi^-1          (for Ji = -Adj(G_ij)^T Jj, geom.py:277)
  Rigid Rr;   // R_qj^-1                        (Ja rows <- Adj(R_qj^-1)^T, geom.py:273)
  cam::Intr Ij;  // target-view intrinsics at 1/8 scale
  int e;      // term index
  int sj;     // slot of pose pj in the reduced system, -1 if fixed
  int merge;  // pi == pj: Jj is added to Ji (same variable block)
  int rig_adj;  // R_qj is not the identity: Ja rows need Adj(R_qj^-1)^T
};

__device__ __forceinline__ void term_transforms(const float* poses, const float* rig, int pi, int qi, int pj, int qj,
                                                Rigid& T, Rigid& G, Rigid& Rr) {
  using SE3f = lie::SE3<float>;
  SE3f Gi(poses + 7 * pi), Gj(poses + 7 * pj);
  SE3f Gij = Gj * Gi.inv();
  SE3f Rji = SE3f(rig + 7 * qj).inv();
  SE3f Tt = (Rji * Gij) * SE3f(rig + 7 * qi);
  T = to_rigid(Tt);
  G = to_rigid(Gij);
  Rr = to_rigid(Rji);
}

// b = Adj(X)^T a for X = (R,t):  [R^T a1, R^T (a2 - t x a1)]   (se3.h:83)
__device__ __forceinline__ void adjT_apply(const Rigid& X, const float* a, float* b) {
  const float c0 = a[3] - (X.t[1] * a[2] - X.t[2] * a[1]);
  const float c1 = a[4] - (X.t[2] * a[0] - X.t[0] * a[2]);
  const float c2 = a[5] - (X.t[0] * a[1] - X.t[1] * a[0]);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    b[j] = X.R[0 + j] * a[0] + X.R[3 + j] * a[1] + X.R[6 + j] * a[2];
    b[3 + j] = X.R[0 + j] * c0 + X.R[3 + j] * c1 + X.R[6 + j] * c2;
  }
}
