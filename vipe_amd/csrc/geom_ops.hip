// slam_ext geometry kernels that the SLAM host code calls between update iterations:
//   frame_distance (edge proposal, LIVE: vipe/slam/maths/geom.py:343), depth_filter (map extraction, LIVE:
//   vipe/slam/components/buffer.py:625), projmap and iproj (bound but dormant).
// Replaces csrc/slam_ext/geom_kernels.cu:434-861.  Arithmetic follows the reference kernels: relative pose from
// raw quaternions without re-normalisation (relSE3, :145-156), quaternion point action (actSO3, :106-116),
// MIN_DEPTH = 0.25 (:33), float math except where the reference's double literals promote (depth_filter's
// 1.0 / d terms, :783-790).
#include "common.cuh"
#include "lie_math.h"

#pragma clang fp contract(off)

namespace {

constexpr float GEOM_MIN_DEPTH = 0.25f;

struct Rel {
  float t[3], q[4];
};

__device__ __forceinline__ void act_so3(const float* q, const float* X, float* Y) {
  float uv[3];
  uv[0] = 2.0f * (q[1] * X[2] - q[2] * X[1]);
  uv[1] = 2.0f * (q[2] * X[0] - q[0] * X[2]);
  uv[2] = 2.0f * (q[0] * X[1] - q[1] * X[0]);
  Y[0] = X[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  Y[1] = X[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  Y[2] = X[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

__device__ __forceinline__ void act_se3(const Rel& T, const float* X, float* Y) {
  act_so3(T.q, X, Y);
  Y[3] = X[3];
  Y[0] += X[3] * T.t[0];
  Y[1] += X[3] * T.t[1];
  Y[2] += X[3] * T.t[2];
}

__device__ __forceinline__ Rel rel_se3(const float* pi, const float* pj) {
  const float *ti = pi, *qi = pi + 3, *tj = pj, *qj = pj + 3;
  Rel r;
  r.q[0] = -qj[3] * qi[0] + qj[0] * qi[3] - qj[1] * qi[2] + qj[2] * qi[1];
  r.q[1] = -qj[3] * qi[1] + qj[1] * qi[3] - qj[2] * qi[0] + qj[0] * qi[2];
  r.q[2] = -qj[3] * qi[2] + qj[2] * qi[3] - qj[0] * qi[1] + qj[1] * qi[0];
  r.q[3] = qj[3] * qi[3] + qj[0] * qi[0] + qj[1] * qi[1] + qj[2] * qi[2];
  float tmp[3];
  act_so3(r.q, ti, tmp);
  r.t[0] = tj[0] - tmp[0];
  r.t[1] = tj[1] - tmp[1];
  r.t[2] = tj[2] - tmp[2];
  return r;
}

// the distance of one ordered pair: lanes stride over the pixels, wave-shuffle + LDS reduction; valid in thread 0
__device__ __forceinline__ float pair_distance(const Rel& T, const float* Ii, const float* Ij, const float* __restrict__ d, int ht,
                                               int wd, float beta, float (*red)[4]) {
  const float fxi = Ii[0], fyi = Ii[1], cxi = Ii[2], cyi = Ii[3];
  const float fxj = Ij[0], fyj = Ij[1], cxj = Ij[2], cyj = Ij[3];
  float accum = 0.f, valid = 0.f, total = 0.f;
  for (int k = threadIdx.x; k < ht * wd; k += blockDim.x) {
    const float u = (float)(k % wd), v = (float)(k / wd);
    float Xi[4] = {(u - cxi) / fxi, (v - cyi) / fyi, 1.0f, d[k]}, Xj[4];
    act_se3(T, Xi, Xj);
    float du = fxj * (Xj[0] / Xj[2]) + cxj - u;
    float dv = fyj * (Xj[1] / Xj[2]) + cyj - v;
    float dd = sqrtf(du * du + dv * dv);
    total += beta;
    if (Xj[2] > GEOM_MIN_DEPTH) {
      accum += beta * dd;
      valid += beta;
    }
    Xj[0] = Xi[0] + Xi[3] * T.t[0];
    Xj[1] = Xi[1] + Xi[3] * T.t[1];
    Xj[2] = Xi[2] + Xi[3] * T.t[2];
    du = fxj * (Xj[0] / Xj[2]) + cxj - u;
    dv = fyj * (Xj[1] / Xj[2]) + cyj - v;
    dd = sqrtf(du * du + dv * dv);
    total += (1 - beta);
    if (Xj[2] > GEOM_MIN_DEPTH) {
      accum += (1 - beta) * dd;
      valid += (1 - beta);
    }
  }
  accum = wave_sum(accum);
  valid = wave_sum(valid);
  total = wave_sum(total);
  __syncthreads();  // red may still be read by thread 0 of an earlier call
  if (lane_id() == 0) {
    red[0][wave_id()] = accum;
    red[1][wave_id()] = valid;
    red[2][wave_id()] = total;
  }
  __syncthreads();
  const float a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  const float vv = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const float tt = red[2][0] + red[2][1] + red[2][2] + red[2][3];
  return (vv / (tt + 1e-8) < 0.75) ? 1000.0f : a / vv;  // geom_kernels.cu:674
}

// one workgroup per candidate pair
__global__ __launch_bounds__(256) void frame_distance_kernel(const float* __restrict__ poses,
                                                             const float* __restrict__ disps,
                                                             const float* __restrict__ intr,
                                                             const int64_t* __restrict__ pi,
                                                             const int64_t* __restrict__ pj,
                                                             const int64_t* __restrict__ ri,
                                                             const int64_t* __restrict__ rj,
                                                             const int64_t* __restrict__ di, float* __restrict__ dist,
                                                             int ht, int wd, float beta) {
  const int b = blockIdx.x;
  __shared__ Rel T;
  __shared__ float red[3][4];
  if (threadIdx.x == 0) T = rel_se3(poses + 7 * pi[b], poses + 7 * pj[b]);
  __syncthreads();
  const float r = pair_distance(T, intr + 4 * ri[b], intr + 4 * rj[b], disps + (int64_t)di[b] * ht * wd, ht, wd, beta, red);
  if (threadIdx.x == 0) dist[b] = r;
}

// [fused] GraphBuffer.frame_distance_dense_disp (buffer.py:550-593) for one candidate (keyframe pi, view qi) -> (pj, qj):
// the per-view poses R_q^-1 G_p (geom.py:338: the reference expands them for ALL frames on the host side), the pinhole
// intrinsics at 1 / factor scale (geom.py:335; MEI through its pinhole equivalent, cameras.py:338-343), the distance and -
// `bidir` - the reverse one, averaged.  The group operations are lie_math.h's, applied exactly as the lietorch calls of the
// unfused path apply them (every operand re-normalised when it is loaded), so both paths give the same bits.
__global__ __launch_bounds__(256) void frame_distance_rig_kernel(const float* __restrict__ poses, const float* __restrict__ rig,
                                                                 const float* __restrict__ disps,
                                                                 const float* __restrict__ intr_full, int idim, float factor,
                                                                 const int64_t* __restrict__ pi, const int64_t* __restrict__ qi,
                                                                 const int64_t* __restrict__ pj, const int64_t* __restrict__ qj,
                                                                 float* __restrict__ dist, int V, int ht, int wd, float beta,
                                                                 int bidir) {
  const int b = blockIdx.x;
  __shared__ Rel Tij, Tji;
  __shared__ float I[2][4];
  __shared__ float red[3][4];
  if (threadIdx.x == 0) {
    float e[2][7];
    const int64_t pp[2] = {pi[b], pj[b]}, qq[2] = {qi[b], qj[b]};
    for (int s = 0; s < 2; ++s) {
      float tmp[7];
      lie::SE3<float>(rig + 7 * qq[s]).inv().store(tmp);  // lietorch inv, stored ...
      (lie::SE3<float>(tmp) * lie::SE3<float>(poses + 7 * pp[s])).store(e[s]);  // ... and loaded again by the product
      const float* k = intr_full + (int64_t)idim * qq[s];
      for (int c = 0; c < 4; ++c) I[s][c] = k[c] / factor;
      if (idim == 5) {  // MEI: f / (1 + k1)
        I[s][0] = I[s][0] / (1 + k[4]);
        I[s][1] = I[s][1] / (1 + k[4]);
      }
    }
    Tij = rel_se3(e[0], e[1]);
    Tji = rel_se3(e[1], e[0]);
  }
  __syncthreads();
  const int64_t P = (int64_t)ht * wd;
  float r = pair_distance(Tij, I[0], I[1], disps + (pi[b] * V + qi[b]) * P, ht, wd, beta, red);
  if (bidir) {
    const float r2 = pair_distance(Tji, I[1], I[0], disps + (pj[b] * V + qj[b]) * P, ht, wd, beta, red);
    r = 0.5f * (r + r2);
  }
  if (threadIdx.x == 0) dist[b] = r;
}

// one lane per (keyframe slot, pixel): walks the 6 temporal neighbours itself, so the count needs no atomics
__global__ __launch_bounds__(256) void depth_filter_kernel(const float* __restrict__ poses,
                                                           const float* __restrict__ disps,
                                                           const float* __restrict__ intr,
                                                           const int64_t* __restrict__ inds,
                                                           const float* __restrict__ thresh, float* __restrict__ counter,
                                                           int num, int ht, int wd) {
  const int b = blockIdx.y;
  const int index = blockIdx.x * blockDim.x + threadIdx.x;
  const int ix = (int)inds[b];
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float t = thresh[b];
  __shared__ Rel T[6];
  __shared__ int jxs[6];
  if (threadIdx.x < 6) {
    const int nb = threadIdx.x;
    const int jx = (nb < 3) ? ix - nb - 1 : ix + nb - 2;  // geom_kernels.cu:709
    jxs[nb] = jx;
    if (jx >= 0 && jx < num) T[nb] = rel_se3(poses + 7 * ix, poses + 7 * jx);
  }
  __syncthreads();
  if (index >= ht * wd) return;
  const int i = index / wd, j = index % wd;
  const float di = disps[((int64_t)ix * ht + i) * wd + j];
  float cnt = 0.0f;
  for (int nb = 0; nb < 6; ++nb) {
    const int jx = jxs[nb];
    if (jx < 0 || jx >= num) continue;
    float Xi[4] = {((float)j - cx) / fx, ((float)i - cy) / fy, 1.0f, di}, Xj[4];
    act_se3(T[nb], Xi, Xj);
    const float uj = fx * (Xj[0] / Xj[2]) + cx;
    const float vj = fy * (Xj[1] / Xj[2]) + cy;
    const float dj = Xj[3] / Xj[2];
    const int u0 = (int)floorf(uj), v0 = (int)floorf(vj);
    if (u0 >= 0 && v0 >= 0 && u0 < wd - 1 && v0 < ht - 1) {
      const float* dn = disps + (int64_t)jx * ht * wd;
      const float d00 = dn[v0 * wd + u0], d01 = dn[v0 * wd + u0 + 1];
      const float d10 = dn[(v0 + 1) * wd + u0], d11 = dn[(v0 + 1) * wd + u0 + 1];
      const double inv = 1.0 / (double)dj, tt = (double)t;  // the reference's `1.0 / dj` is a double expression
      if (fabs(inv - 1.0 / (double)d00) < tt) cnt += 1.0f;
      else if (fabs(inv - 1.0 / (double)d01) < tt) cnt += 1.0f;
      else if (fabs(inv - 1.0 / (double)d10) < tt) cnt += 1.0f;
      else if (fabs(inv - 1.0 / (double)d11) < tt) cnt += 1.0f;
    }
  }
  counter[((int64_t)b * ht + i) * wd + j] = cnt;
}

__global__ __launch_bounds__(256) void projmap_kernel(const float* __restrict__ poses, const float* __restrict__ disps,
                                                      const float* __restrict__ intr, const int64_t* __restrict__ ii,
                                                      const int64_t* __restrict__ jj, float* __restrict__ coords,
                                                      float* __restrict__ valid, int ht, int wd) {
  const int b = blockIdx.y;
  __shared__ Rel T;
  if (threadIdx.x == 0) T = rel_se3(poses + 7 * ii[b], poses + 7 * jj[b]);
  __syncthreads();
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ht * wd) return;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  const float u = (float)(k % wd), v = (float)(k / wd);
  float Xi[4] = {(u - cx) / fx, (v - cy) / fy, 1.0f, disps[(int64_t)ii[b] * ht * wd + k]}, Xj[4];
  act_se3(T, Xi, Xj);
  float* c = coords + ((int64_t)b * ht * wd + k) * 3;
  c[0] = u;
  c[1] = v;
  if (Xj[2] > 0.01f) {
    c[0] = fx * (Xj[0] / Xj[2]) + cx;
    c[1] = fy * (Xj[1] / Xj[2]) + cy;
  }
  valid[(int64_t)b * ht * wd + k] = (Xj[2] > GEOM_MIN_DEPTH) ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void iproj_kernel(const float* __restrict__ poses, const float* __restrict__ disps,
                                                    const float* __restrict__ intr, float* __restrict__ points, int ht,
                                                    int wd) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ht * wd) return;
  const float fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
  Rel T;
  for (int q = 0; q < 3; ++q) T.t[q] = poses[7 * b + q];
  for (int q = 0; q < 4; ++q) T.q[q] = poses[7 * b + 3 + q];
  float Xi[4] = {((float)(k % wd) - cx) / fx, ((float)(k / wd) - cy) / fy, 1.0f, disps[(int64_t)b * ht * wd + k]}, Xj[4];
  act_se3(T, Xi, Xj);
  float* p = points + ((int64_t)b * ht * wd + k) * 3;
  p[0] = Xj[0] / Xj[3];
  p[1] = Xj[1] / Xj[3];
  p[2] = Xj[2] / Xj[3];
}

}  // namespace

VIPE_EXPORT int vipe_frame_distance(const float* d_poses, const float* d_disps, const float* d_intrinsics,
                                    const int64_t* d_pi, const int64_t* d_pj, const int64_t* d_qi, const int64_t* d_qj,
                                    const int64_t* d_di, float* d_dist, int M, int ht, int wd, float beta, void* stream) {
  VIPE_CHECK_ARG(M >= 0 && ht > 0 && wd > 0);
  if (M == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_poses && d_disps && d_intrinsics && d_pi && d_pj && d_qi && d_qj && d_di && d_dist);
  frame_distance_kernel<<<M, 256, 0, as_stream(stream)>>>(d_poses, d_disps, d_intrinsics, d_pi, d_pj, d_qi, d_qj, d_di,
                                                          d_dist, ht, wd, beta);
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_frame_distance_rig(const float* d_poses, const float* d_rig, const float* d_disps, const float* d_intrinsics,
                                        int intr_dim, float intr_factor, const int64_t* d_pi, const int64_t* d_qi,
                                        const int64_t* d_pj, const int64_t* d_qj, float* d_dist, int M, int n_views, int ht,
                                        int wd, float beta, int bidirectional, void* stream) {
  VIPE_CHECK_ARG(M >= 0 && ht > 0 && wd > 0 && n_views > 0 && (intr_dim == 4 || intr_dim == 5) && intr_factor > 0);
  if (M == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_poses && d_rig && d_disps && d_intrinsics && d_pi && d_pj && d_qi && d_qj && d_dist);
  frame_distance_rig_kernel<<<M, 256, 0, as_stream(stream)>>>(d_poses, d_rig, d_disps, d_intrinsics, intr_dim, intr_factor, d_pi,
                                                              d_qi, d_pj, d_qj, d_dist, n_views, ht, wd, beta, bidirectional);
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_depth_filter(const float* d_poses, const float* d_disps, const float* d_intrinsics,
                                  const int64_t* d_inds, const float* d_thresh, float* d_counter, int n, int num,
                                  int ht, int wd, void* stream) {
  VIPE_CHECK_ARG(n >= 0 && num >= 0 && ht > 0 && wd > 0 && num <= 65535);
  if (num == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_poses && d_disps && d_intrinsics && d_inds && d_thresh && d_counter);
  depth_filter_kernel<<<dim3((ht * wd + 255) / 256, num), 256, 0, as_stream(stream)>>>(
      d_poses, d_disps, d_intrinsics, d_inds, d_thresh, d_counter, n, ht, wd);
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_projmap(const float* d_poses, const float* d_disps, const float* d_intrinsics,
                             const int64_t* d_ii, const int64_t* d_jj, float* d_coords, float* d_valid, int E, int ht,
                             int wd, void* stream) {
  VIPE_CHECK_ARG(E >= 0 && ht > 0 && wd > 0 && E <= 65535);
  if (E == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_poses && d_disps && d_intrinsics && d_ii && d_jj && d_coords && d_valid);
  projmap_kernel<<<dim3((ht * wd + 255) / 256, E), 256, 0, as_stream(stream)>>>(d_poses, d_disps, d_intrinsics, d_ii,
                                                                                d_jj, d_coords, d_valid, ht, wd);
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_iproj(const float* d_poses, const float* d_disps, const float* d_intrinsics, float* d_points,
                           int n, int ht, int wd, void* stream) {
  VIPE_CHECK_ARG(n >= 0 && ht > 0 && wd > 0 && n <= 65535);
  if (n == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_poses && d_disps && d_intrinsics && d_points);
  iproj_kernel<<<dim3((ht * wd + 255) / 256, n), 256, 0, as_stream(stream)>>>(d_poses, d_disps, d_intrinsics, d_points,
                                                                              ht, wd);
  return vipe_launch_status();
}
