// Dense reprojection coords1 = pi(T_ij * pi^-1(px, disp_i)) for every term, plus the fused motion-feature
// form used by FactorGraph.update.  Replaces the ~20 small torch kernels + lietorch mul/inv/act4 of
// GraphBuffer.reproject_dense_disp (vipe/slam/components/buffer.py:527-548 -> maths/geom.py:187-263) and
// the cat/permute/clamp of factor_graph.py:259-261.  One block row per term, one lane per pixel,
// coalesced float2 stores; the term transform is built once per block in LDS.
#include "term_geom.cuh"

namespace {

struct ReprojArgs {
  const float *poses, *disps, *intr, *rig;
  const int64_t *pi, *qi, *pj, *qj, *di;
  const float* target;
  float* coords;
  float* valid;
  void* motn;
  int M, ht, wd, V, D;
  float inv_factor;
};

template <int CAM, int MOTN /*0 none, 1 f16 [M,4,P], 2 f32 [M,4,P], 3 f16 channels-last [M,P,4]*/>
__global__ __launch_bounds__(256) void reproject_kernel(ReprojArgs a) {
  const int e = blockIdx.y;
  const int P = a.ht * a.wd;
  __shared__ Rigid T;
  __shared__ cam::Intr Ii, Ij;
  __shared__ int dframe;
  if (threadIdx.x == 0) {
    const int pi = (int)a.pi[e], qi = (int)a.qi[e], pj = (int)a.pj[e], qj = (int)a.qj[e];
    Rigid G, Rr;
    term_transforms(a.poses, a.rig, pi, qi, pj, qj, T, G, Rr);
    Ii = cam::load_scaled(a.intr + qi * (4 + a.D), a.D, a.inv_factor);
    Ij = cam::load_scaled(a.intr + qj * (4 + a.D), a.D, a.inv_factor);
    dframe = (int)a.di[e];
  }
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const float u = (float)(p % a.wd), v = (float)(p / a.wd);
  const float d = a.disps[(int64_t)dframe * P + p];
  float X0, Y0, dX[1], dY[1];
  cam::iproj<CAM, 0>(Ii, u, v, X0, Y0, dX, dY);
  const float X = T.R[0] * X0 + T.R[1] * Y0 + T.R[2] + T.t[0] * d;
  const float Y = T.R[3] * X0 + T.R[4] * Y0 + T.R[5] + T.t[1] * d;
  const float Z = T.R[6] * X0 + T.R[7] * Y0 + T.R[8] + T.t[2] * d;
  float x, y, Jp[2][3], Jf[2][1];
  cam::proj<CAM, false, 0>(Ij, X, Y, Z, x, y, Jp, Jf);
  const int64_t o = (int64_t)e * P + p;
  reinterpret_cast<float2*>(a.coords)[o] = make_float2(x, y);
  if (a.valid) a.valid[o] = (Z > cam::MIN_DEPTH) ? 1.0f : 0.0f;  // X0.z == 1 > MIN_DEPTH (geom.py:263)
  if constexpr (MOTN != 0) {
    const float2 tg = reinterpret_cast<const float2*>(a.target)[o];
    float m[4] = {x - u, y - v, tg.x - x, tg.y - y};  // factor_graph.py:259
    if constexpr (MOTN == 3) {
      typedef _Float16 half4v __attribute__((ext_vector_type(4)));
      half4v hv;
#pragma unroll
      for (int c = 0; c < 4; ++c) hv[c] = (half_t)fminf(fmaxf(m[c], -64.0f), 64.0f);
      reinterpret_cast<half4v*>(a.motn)[o] = hv;
      return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float mc = fminf(fmaxf(m[c], -64.0f), 64.0f);
      const int64_t oo = ((int64_t)e * 4 + c) * P + p;
      if constexpr (MOTN == 1) reinterpret_cast<half_t*>(a.motn)[oo] = (half_t)mc;
      else reinterpret_cast<float*>(a.motn)[oo] = mc;
    }
  }
}

template <int MOTN>
int launch(const ReprojArgs& a, int camera, hipStream_t s) {
  dim3 grid((a.ht * a.wd + 255) / 256, a.M), block(256);
  if (camera == VIPE_CAM_PINHOLE) reproject_kernel<VIPE_CAM_PINHOLE, MOTN><<<grid, block, 0, s>>>(a);
  else if (camera == VIPE_CAM_MEI) reproject_kernel<VIPE_CAM_MEI, MOTN><<<grid, block, 0, s>>>(a);
  else return VIPE_EINVAL;
  return vipe_launch_status();
}

}  // namespace

VIPE_EXPORT int vipe_reproject(const float* d_poses, const float* d_disps, const float* d_intrinsics,
                               const float* d_rig, const int64_t* d_pi, const int64_t* d_qi, const int64_t* d_pj,
                               const int64_t* d_qj, const int64_t* d_di, float* d_coords, float* d_valid, int M,
                               int ht, int wd, int n_views, int camera, float intr_factor, void* stream) {
  VIPE_CHECK_ARG(M >= 0 && M <= 65535 && ht > 0 && wd > 0 && n_views >= 1 && intr_factor > 0);
  if (M == 0) return VIPE_OK;  // before the pointer checks: an empty tensor has a null data pointer
  VIPE_CHECK_ARG(d_poses && d_disps && d_intrinsics && d_rig && d_pi && d_qi && d_pj && d_qj && d_di && d_coords);
  ReprojArgs a{d_poses, d_disps, d_intrinsics, d_rig, d_pi, d_qi, d_pj, d_qj, d_di, nullptr, d_coords, d_valid,
               nullptr, M, ht, wd, n_views, camera == VIPE_CAM_MEI ? 1 : 0, 1.0f / intr_factor};
  return launch<0>(a, camera, as_stream(stream));
}

VIPE_EXPORT int vipe_reproject_motion(const float* d_poses, const float* d_disps, const float* d_intrinsics,
                                      const float* d_rig, const int64_t* d_pi, const int64_t* d_qi,
                                      const int64_t* d_pj, const int64_t* d_qj, const int64_t* d_di,
                                      const float* d_target, float* d_coords, void* d_motn, int M, int ht, int wd,
                                      int n_views, int camera, float intr_factor, int motn_dtype, void* stream) {
  VIPE_CHECK_ARG(M >= 0 && M <= 65535 && ht > 0 && wd > 0 && n_views >= 1 && intr_factor > 0);
  if (M == 0) return VIPE_OK;  // before the pointer checks: an empty tensor has a null data pointer
  VIPE_CHECK_ARG(d_poses && d_disps && d_intrinsics && d_rig && d_pi && d_qi && d_pj && d_qj && d_di && d_coords);
  VIPE_CHECK_ARG(d_target && d_motn);
  ReprojArgs a{d_poses, d_disps, d_intrinsics, d_rig, d_pi, d_qi, d_pj, d_qj, d_di, d_target, d_coords, nullptr,
               d_motn, M, ht, wd, n_views, camera == VIPE_CAM_MEI ? 1 : 0, 1.0f / intr_factor};
  if (motn_dtype == VIPE_F16) return launch<1>(a, camera, as_stream(stream));
  if (motn_dtype == VIPE_F32) return launch<2>(a, camera, as_stream(stream));
  return VIPE_EINVAL;
}

VIPE_EXPORT int vipe_reproject_motion_nhwc(const float* d_poses, const float* d_disps, const float* d_intrinsics,
                                           const float* d_rig, const int64_t* d_pi, const int64_t* d_qi,
                                           const int64_t* d_pj, const int64_t* d_qj, const int64_t* d_di,
                                           const float* d_target, float* d_coords, void* d_motn, int M, int ht,
                                           int wd, int n_views, int camera, float intr_factor, void* stream) {
  VIPE_CHECK_ARG(M >= 0 && M <= 65535 && ht > 0 && wd > 0 && n_views >= 1 && intr_factor > 0);
  if (M == 0) return VIPE_OK;  // before the pointer checks: an empty tensor has a null data pointer
  VIPE_CHECK_ARG(d_poses && d_disps && d_intrinsics && d_rig && d_pi && d_qi && d_pj && d_qj && d_di && d_coords);
  VIPE_CHECK_ARG(d_target && d_motn);
  ReprojArgs a{d_poses, d_disps, d_intrinsics, d_rig, d_pi, d_qi, d_pj, d_qj, d_di, d_target, d_coords, nullptr,
               d_motn, M, ht, wd, n_views, camera == VIPE_CAM_MEI ? 1 : 0, 1.0f / intr_factor};
  return launch<3>(a, camera, as_stream(stream));
}
