// API-completeness kernels off the per-iteration hot loop:
//   altcorr_forward  (droid_net_ext; backend's volume-free correlation, altcorr_kernel.cu:26-138)
//   scatter          (scatter_ext; atomic scatter sum/mul/min/max + arg pass, scatter_cuda.cu:20-55)
//   corr sampler     (corr_ext; FlowNet-style local correlation, correlation_cuda_kernel.cu:27-214)
//   segment mean     ([fused] GraphAgg's scatter_mean over the edges of each source node, droid_net.py:420-421)
#include "common.cuh"

#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------------------------------------ altcorr
// altcorr_forward (altcorr_kernel.cu:26-138): the 7 x 7 bilinear window of a source pixel correlated against the target
// map WITHOUT a volume - per tap a C-channel dot product.  Workgroup = a 4 x 8 tile of source pixels (the reference's
// tile) x 8 lanes per pixel, lane j = tap row j of the 8 x 8 integer taps.  Per coordinate set and 32-channel slab:
//   * the bounding box of the tile's 32 windows (smooth flow: ~11 x 15 positions for 2048 tap reads) is staged in LDS ONCE,
//     as [4-channel chunk][position] float4 rows (odd pitch: the 8 chunk writes of a position and the 64 lanes' tap reads
//     spread over the banks) - 128 contiguous bytes per position from L2 instead of a re-gather per tap and pixel;
//   * lane (pixel, j) forms the 8 dot products of its tap row against the pixel's slab held in registers, takes row j + 1
//     from its neighbour lane by shuffle and owns output row j: out[j][ox] accumulates the four bilinear contributions in
//     the reference's order - taps (j, ox), (j, ox + 1), (j + 1, ox), (j + 1, ox + 1), slab after slab (:50, 95-133) -
//     so every output is written once, from registers (the reference: 4 global read-modify-writes per tap and slab).
// Windows scattered too widely for the LDS box (> ALT_BOX positions) read their taps straight from L2 with the same
// arithmetic.  fp32 accumulation for both dtypes; -ffp-contract=off keeps mul and add apart.
constexpr int ALT_BOX = 383;  // positions of a staged box (odd: it is also the largest pitch)

template <typename T>
__device__ __forceinline__ float4 alt_load4(const T* p, int c, int C) {
  // channels c .. c + 3 of a channels-last pixel (zero beyond C)
  if (c + 4 <= C && (C & 3) == 0) {
    if constexpr (sizeof(T) == 4) {
      return *reinterpret_cast<const float4*>(p + c);
    } else {
      typedef _Float16 half4 __attribute__((ext_vector_type(4)));
      const half4 h = *reinterpret_cast<const half4*>(p + c);
      return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
  }
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) v.x = (float)p[c];
  if (c + 1 < C) v.y = (float)p[c + 1];
  if (c + 2 < C) v.z = (float)p[c + 2];
  if (c + 3 < C) v.w = (float)p[c + 3];
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void altcorr_forward_kernel(const T* __restrict__ f1, const T* __restrict__ f2,
                                                              const float* __restrict__ coords, T* __restrict__ corr,
                                                              int H1, int W1, int H2, int W2, int N, int C) {
  constexpr int R = 3, RD = 7;
  __shared__ float4 box[8 * ALT_BOX];
  __shared__ int bb[4];  // box: min x, min y, max x, max y of the windows' first taps
  const int tid = threadIdx.x, j = tid & 7, pl = tid >> 3;
  const int tiles_x = (W1 + 7) >> 3;
  const int ty0 = (blockIdx.x / tiles_x) * 4, tx0 = (blockIdx.x % tiles_x) * 8;
  const int b = blockIdx.y;
  const int py = ty0 + (pl >> 3), px = tx0 + (pl & 7);
  const bool pok = py < H1 && px < W1;
  const int P = H1 * W1, p = py * W1 + px;
  const T* a = f1 + ((int64_t)b * P + (pok ? p : 0)) * C;
  const T* f2b = f2 + (int64_t)b * H2 * W2 * C;
  for (int n = 0; n < N; ++n) {
    float2 c = make_float2(0.f, 0.f);
    if (pok) c = reinterpret_cast<const float2*>(coords)[((int64_t)b * N + n) * P + p];
    const float fx = floorf(c.x), fy = floorf(c.y);
    const float dx = c.x - fx, dy = c.y - fy;
    // keep absurd coordinates (the operator clamps flow to +-64 px, not positions) inside int range
    const int bx = (int)fminf(fmaxf(fx, -1.0e6f), 1.0e6f) - R, by = (int)fminf(fmaxf(fy, -1.0e6f), 1.0e6f) - R;
    // ---- bounding box of the tile's windows, clipped to the target map
    __syncthreads();  // previous coordinate set done with bb / box
    if (tid == 0) { bb[0] = 1 << 30; bb[1] = 1 << 30; bb[2] = -(1 << 30); bb[3] = -(1 << 30); }
    __syncthreads();
    if (pok && j == 0) {
      atomicMin(&bb[0], bx); atomicMin(&bb[1], by); atomicMax(&bb[2], bx); atomicMax(&bb[3], by);
    }
    __syncthreads();
    const int x0 = max(bb[0], 0), y0 = max(bb[1], 0);
    const int x1 = min(bb[2] + RD, W2 - 1), y1 = min(bb[3] + RD, H2 - 1);
    const int BW = x1 - x0 + 1, BH = y1 - y0 + 1;
    const bool empty = BW <= 0 || BH <= 0;                               // every tap of the tile lies outside the map
    const bool staged = !empty && (int64_t)BW * BH <= ALT_BOX;
    const int pitch = (BW * BH) | 1;
    float acc[RD];
#pragma unroll
    for (int q = 0; q < RD; ++q) acc[q] = 0.0f;
    const float w_se = (1 - dy) * (1 - dx), w_sw = (1 - dy) * dx, w_ne = dy * (1 - dx), w_nw = dy * dx;
    const int ty = by + j;  // this lane's tap row
    for (int c0 = 0; c0 < C && !empty; c0 += 32) {
      if (staged) {
        __syncthreads();  // previous slab's readers are done
        for (int i = tid; i < BW * BH * 8; i += 256) {
          const int pos = i >> 3, k4 = i & 7;
          const int yy = y0 + pos / BW, xx = x0 + pos % BW;
          box[k4 * pitch + pos] = alt_load4(f2b + ((int64_t)yy * W2 + xx) * C, c0 + 4 * k4, C);
        }
        __syncthreads();
      }
      float av[32];
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        const float4 v = pok ? alt_load4(a, c0 + 4 * k4, C) : make_float4(0.f, 0.f, 0.f, 0.f);
        av[4 * k4] = v.x; av[4 * k4 + 1] = v.y; av[4 * k4 + 2] = v.z; av[4 * k4 + 3] = v.w;
      }
      float s[RD + 1];
      const bool rowin = ty >= 0 && ty < H2;
#pragma unroll
      for (int ix = 0; ix <= RD; ++ix) {
        const int tx = bx + ix;
        float t = 0.0f;
        if (pok && rowin && tx >= 0 && tx < W2) {
          if (staged) {
            const float4* g = box + (ty - y0) * BW + (tx - x0);
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
              const float4 v = g[k4 * pitch];
              t += av[4 * k4] * v.x; t += av[4 * k4 + 1] * v.y; t += av[4 * k4 + 2] * v.z; t += av[4 * k4 + 3] * v.w;
            }
          } else {
            const T* g = f2b + ((int64_t)ty * W2 + tx) * C;
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
              const float4 v = alt_load4(g, c0 + 4 * k4, C);
              t += av[4 * k4] * v.x; t += av[4 * k4 + 1] * v.y; t += av[4 * k4 + 2] * v.z; t += av[4 * k4 + 3] * v.w;
            }
          }
        }
        s[ix] = t;
      }
      // output row j of this pixel: rows j (own) and j + 1 (lane j + 1 of the 8-lane group)
#pragma unroll
      for (int ox = 0; ox < RD; ++ox) {
        const float n0 = __shfl_down(s[ox], 1, 8), n1 = __shfl_down(s[ox + 1], 1, 8);
        acc[ox] += s[ox] * w_se;
        acc[ox] += s[ox + 1] * w_sw;
        acc[ox] += n0 * w_ne;
        acc[ox] += n1 * w_nw;
      }
    }
    if (pok && j < RD) {
      T* o = corr + (((int64_t)b * N + n) * RD * RD) * P + p;
#pragma unroll
      for (int ox = 0; ox < RD; ++ox) o[(int64_t)(j + RD * ox) * P] = (T)acc[ox];
    }
  }
}

// adjoint of altcorr_forward with respect to both feature maps (altcorr_kernel.cu:140-264): one wave per (b, n, pixel),
// lanes over channels.  For each of the (2r+2)^2 tap positions the gradients of the up-to-four outputs that the tap
// contributes to are folded with their bilinear weights into one scalar g; then grad_f1[p] += g f2[tap] (kept in
// registers over the taps, one atomic per channel at the end because several n share fmap1[p]) and
// grad_f2[tap] += g f1[p] (atomic).  coords receive no gradient, as in the reference.
template <int R>
__global__ __launch_bounds__(256) void altcorr_backward_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                               const float* __restrict__ coords,
                                                               const float* __restrict__ cgrad, float* __restrict__ g1,
                                                               float* __restrict__ g2, int H1, int W1, int H2, int W2,
                                                               int N, int C, int64_t total) {
  constexpr int RD = 2 * R + 1;
  const int P = H1 * W1;
  const int lane = threadIdx.x & 63;
  for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < total; item += (int64_t)gridDim.x * 4) {
    const int p = (int)(item % P);
    const int n = (int)((item / P) % N);
    const int b = (int)(item / ((int64_t)P * N));
    const float2 c = reinterpret_cast<const float2*>(coords)[((int64_t)b * N + n) * P + p];
    const float fx = floorf(c.x), fy = floorf(c.y);
    const float dx = c.x - fx, dy = c.y - fy;
    const int bx = (int)fx - R, by = (int)fy - R;
    const float* cg = cgrad + (((int64_t)b * N + n) * RD * RD) * P + p;
    const float* a = f1 + ((int64_t)b * P + p) * C;
    float* ga = g1 + ((int64_t)b * P + p) * C;
    for (int c0 = lane; c0 < C; c0 += 64) {
      const float av = a[c0];
      float acc1 = 0.0f;
      for (int iy = 0; iy <= RD; ++iy)
        for (int ix = 0; ix <= RD; ++ix) {
          const int h2 = by + iy, w2 = bx + ix;
          if (h2 < 0 || h2 >= H2 || w2 < 0 || w2 >= W2) continue;
          float g = 0.0f;
          if (iy > 0 && ix > 0) g += cg[(int64_t)((iy - 1) + RD * (ix - 1)) * P] * (dy * dx);
          if (iy > 0 && ix < RD) g += cg[(int64_t)((iy - 1) + RD * ix) * P] * (dy * (1 - dx));
          if (iy < RD && ix > 0) g += cg[(int64_t)(iy + RD * (ix - 1)) * P] * ((1 - dy) * dx);
          if (iy < RD && ix < RD) g += cg[(int64_t)(iy + RD * ix) * P] * ((1 - dy) * (1 - dx));
          const int64_t o2 = (((int64_t)b * H2 + h2) * W2 + w2) * C + c0;
          acc1 += g * f2[o2];
          atomicAdd(g2 + o2, g * av);
        }
      atomicAdd(ga + c0, acc1);
    }
  }
}

// ------------------------------------------------------------------------------------------------ scatter
template <typename T>
__device__ __forceinline__ void atomic_combine(T* addr, T v, int reduce);

template <>
__device__ __forceinline__ void atomic_combine<float>(float* addr, float v, int reduce) {
  if (reduce == 0 || reduce == 2) { atomicAdd(addr, v); return; }
  unsigned* ua = reinterpret_cast<unsigned*>(addr);
  unsigned old = *ua, assumed;
  do {
    assumed = old;
    const float cur = __uint_as_float(assumed);
    const float nv = reduce == 1 ? cur * v : (reduce == 3 ? fminf(cur, v) : fmaxf(cur, v));
    old = atomicCAS(ua, assumed, __float_as_uint(nv));
  } while (old != assumed);
}
template <>
__device__ __forceinline__ void atomic_combine<double>(double* addr, double v, int reduce) {
  if (reduce == 0 || reduce == 2) { atomicAdd(addr, v); return; }
  unsigned long long* ua = reinterpret_cast<unsigned long long*>(addr);
  unsigned long long old = *ua, assumed;
  do {
    assumed = old;
    const double cur = __longlong_as_double(assumed);
    const double nv = reduce == 1 ? cur * v : (reduce == 3 ? fmin(cur, v) : fmax(cur, v));
    old = atomicCAS(ua, assumed, (unsigned long long)__double_as_longlong(nv));
  } while (old != assumed);
}
template <>
__device__ __forceinline__ void atomic_combine<half_t>(half_t* addr, half_t v, int reduce) {
  // CAS on the enclosing aligned 32-bit word (atomics.cuh:148-300 of the reference does the same)
  const uintptr_t ad = reinterpret_cast<uintptr_t>(addr);
  unsigned* ua = reinterpret_cast<unsigned*>(ad & ~(uintptr_t)3);
  const bool hi = ad & 2;
  unsigned old = *ua, assumed;
  do {
    assumed = old;
    unsigned short bits = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
    half_t cur;
    __builtin_memcpy(&cur, &bits, 2);
    const float cf = (float)cur, vf = (float)v;
    const half_t nv = (half_t)(reduce == 0 || reduce == 2 ? cf + vf : reduce == 1 ? cf * vf : reduce == 3 ? fminf(cf, vf) : fmaxf(cf, vf));
    unsigned short nb;
    __builtin_memcpy(&nb, &nv, 2);
    const unsigned nw = hi ? ((assumed & 0xffffu) | ((unsigned)nb << 16)) : ((assumed & 0xffff0000u) | nb);
    old = atomicCAS(ua, assumed, nw);
  } while (old != assumed);
}

// ROWS: `index` is [E] (one slot per row of the scattered dimension, the shape GraphAgg / scatter_mean callers have,
// droid_net.py:420-421) instead of an int64 copy of src's whole shape.  Rows whose slot lies outside [0, N) are skipped
// (torch raises for them; nothing is written out of bounds here).
template <typename T, bool ROWS>
__global__ void scatter_kernel(const T* __restrict__ src, const int64_t* __restrict__ index, T* __restrict__ out,
                               int64_t E, int64_t K, int64_t N, int64_t numel, int reduce) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / (E * K), k = i % K;
    const int64_t idx = ROWS ? index[(i / K) % E] : index[i];
    if (idx < 0 || idx >= N) continue;
    atomic_combine<T>(out + b * N * K + idx * K + k, src[i], reduce);
  }
}
template <typename T, bool ROWS>
__global__ void scatter_arg_kernel(const T* __restrict__ src, const int64_t* __restrict__ index, const T* __restrict__ out,
                                   int64_t* __restrict__ arg, int64_t E, int64_t K, int64_t N, int64_t numel) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / (E * K), e = (i / K) % E, k = i % K;
    const int64_t idx = ROWS ? index[e] : index[i];
    if (idx < 0 || idx >= N) continue;
    const int64_t o = b * N * K + idx * K + k;
    if (src[i] == out[o]) arg[o] = e;  // scatter_cuda.cu:50-52 (ties: last writer wins)
  }
}

template <typename T, bool ROWS>
int run_scatter(const void* src, const int64_t* index, void* out, int64_t* arg, int64_t outer, int64_t E, int64_t K,
                int64_t N, int reduce, hipStream_t s) {
  const int64_t numel = outer * E * K;
  if (numel == 0) return VIPE_OK;
  const int blocks = (int)std::min<int64_t>((numel + 255) / 256, 8192);
  scatter_kernel<T, ROWS><<<blocks, 256, 0, s>>>((const T*)src, index, (T*)out, E, K, N, numel, reduce);
  if (reduce >= 3) {
    if (!arg) return VIPE_EINVAL;
    scatter_arg_kernel<T, ROWS><<<blocks, 256, 0, s>>>((const T*)src, index, (const T*)out, arg, E, K, N, numel);
  }
  return vipe_launch_status();
}

// ------------------------------------------------------------------------------------------------ segment mean
// out[k, :] = mean over edges e with ix[e] == k of src[e, coff:coff+C]; deterministic (fixed edge order per k).
__global__ __launch_bounds__(256) void segment_mean_kernel(const half_t* __restrict__ src, int src_ctot, int src_coff,
                                                           const int* __restrict__ order, const int* __restrict__ rowptr,
                                                           half_t* __restrict__ out, int64_t rows_per_item /*h*w*/, int C) {
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  const int k = blockIdx.y;
  const int beg = rowptr[k], end = rowptr[k + 1];
  const int c8 = C / 8;
  const int64_t n8 = rows_per_item * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / c8;
    const int ch = (int)(i % c8) * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = beg; q < end; ++q) {
      const int e = order[q];
      const half8 v = *reinterpret_cast<const half8*>(src + ((int64_t)e * rows_per_item + row) * src_ctot + src_coff + ch);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
    const float inv = 1.0f / (float)max(end - beg, 1);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)(acc[j] * inv);
    *reinterpret_cast<half8*>(out + ((int64_t)k * rows_per_item + row) * C + ch) = o;
  }
}

// ------------------------------------------------------------------------------------------------ corr sampler
// NCHW inputs, one lane per output element (n, ph, pw, h, w)
template <typename T>
__global__ void corr_sampler_forward_kernel(const T* __restrict__ in1, const T* __restrict__ in2, T* __restrict__ out,
                                            int B, int C, int H, int W, int oH, int oW, int kH, int kW, int patchH,
                                            int patchW, int padH, int padW, int dilH, int dilW, int dpH, int dpW, int dH,
                                            int dW) {
  const int64_t total = (int64_t)B * patchH * patchW * oH * oW;
  // correlation_cuda_kernel.cu:43-51 (forward): radius = dilation * (patch - 1) / 2 in integer arithmetic
  const int radH = dpH * (patchH - 1) / 2, radW = dpW * (patchW - 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % oW), h = (int)((i / oW) % oH);
    const int pw = (int)((i / ((int64_t)oW * oH)) % patchW), ph = (int)((i / ((int64_t)oW * oH * patchW)) % patchH);
    const int n = (int)(i / ((int64_t)oW * oH * patchW * patchH));
    const int si = -padH + h * dH, sj = -padW + w * dW;
    const int phd = ph * dpH - radH, pwd = pw * dpW - radW;
    float s = 0.0f;
    for (int a = 0; a < kH; ++a) {
      const int i1 = si + a * dilH, i2 = i1 + phd;
      if (i1 < 0 || i1 >= H || i2 < 0 || i2 >= H) continue;
      for (int bq = 0; bq < kW; ++bq) {
        const int j1 = sj + bq * dilW, j2 = j1 + pwd;
        if (j1 < 0 || j1 >= W || j2 < 0 || j2 >= W) continue;
        for (int c = 0; c < C; ++c)
          s += (float)in1[(((int64_t)n * C + c) * H + i1) * W + j1] * (float)in2[(((int64_t)n * C + c) * H + i2) * W + j2];
      }
    }
    out[i] = (T)s;
  }
}

__global__ void corr_sampler_backward_kernel(const float* __restrict__ in1, const float* __restrict__ in2,
                                             const float* __restrict__ gout, float* __restrict__ g1,
                                             float* __restrict__ g2, int B, int C, int H, int W, int oH, int oW, int kH,
                                             int kW, int patchH, int patchW, int padH, int padW, int dilH, int dilW,
                                             int dpH, int dpW, int dH, int dW) {
  const int64_t total = (int64_t)B * patchH * patchW * oH * oW;
  // correlation_cuda_kernel.cu:96-117,157-179: the reference's backward kernels centre the patch at (patch - 1) / 2 BEFORE
  // scaling by the patch dilation - its forward kernel (:43-51) after; the two differ for even patch sizes with odd dilation
  const int radH = dpH * ((patchH - 1) / 2), radW = dpW * ((patchW - 1) / 2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % oW), h = (int)((i / oW) % oH);
    const int pw = (int)((i / ((int64_t)oW * oH)) % patchW), ph = (int)((i / ((int64_t)oW * oH * patchW)) % patchH);
    const int n = (int)(i / ((int64_t)oW * oH * patchW * patchH));
    const int si = -padH + h * dH, sj = -padW + w * dW;
    const int phd = ph * dpH - radH, pwd = pw * dpW - radW;
    const float g = gout[i];
    for (int a = 0; a < kH; ++a) {
      const int i1 = si + a * dilH, i2 = i1 + phd;
      if (i1 < 0 || i1 >= H || i2 < 0 || i2 >= H) continue;
      for (int bq = 0; bq < kW; ++bq) {
        const int j1 = sj + bq * dilW, j2 = j1 + pwd;
        if (j1 < 0 || j1 >= W || j2 < 0 || j2 >= W) continue;
        for (int c = 0; c < C; ++c) {
          const int64_t o1 = (((int64_t)n * C + c) * H + i1) * W + j1, o2 = (((int64_t)n * C + c) * H + i2) * W + j2;
          atomicAdd(g1 + o1, g * in2[o2]);
          atomicAdd(g2 + o2, g * in1[o1]);
        }
      }
    }
  }
}

}  // namespace

VIPE_EXPORT int vipe_altcorr_forward(const void* d_fmap1, const void* d_fmap2, const float* d_coords, void* d_corr,
                                     int B, int H1, int W1, int H2, int W2, int N, int C, int radius, int dtype,
                                     void* stream) {
  VIPE_CHECK_ARG(B >= 0 && N >= 0 && H1 > 0 && W1 > 0 && H2 > 0 && W2 > 0 && C > 0 && B <= 65535 && N <= 65535);
  if (B == 0 || N == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_fmap1 && d_fmap2 && d_coords && d_corr);
  if (radius != 3) return VIPE_EUNSUPPORTED;
  dim3 grid(((H1 + 3) / 4) * ((W1 + 7) / 8), B);
  hipStream_t s = as_stream(stream);
  if (dtype == VIPE_F32)
    altcorr_forward_kernel<float><<<grid, 256, 0, s>>>((const float*)d_fmap1, (const float*)d_fmap2, d_coords,
                                                       (float*)d_corr, H1, W1, H2, W2, N, C);
  else if (dtype == VIPE_F16)
    altcorr_forward_kernel<half_t><<<grid, 256, 0, s>>>((const half_t*)d_fmap1, (const half_t*)d_fmap2, d_coords,
                                                        (half_t*)d_corr, H1, W1, H2, W2, N, C);
  else return VIPE_EINVAL;
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_altcorr_backward(const float* d_fmap1, const float* d_fmap2, const float* d_coords,
                                      const float* d_corr_grad, float* d_fmap1_grad, float* d_fmap2_grad, int B, int H1,
                                      int W1, int H2, int W2, int N, int C, int radius, void* stream) {
  VIPE_CHECK_ARG(B >= 0 && N >= 0 && H1 > 0 && W1 > 0 && H2 > 0 && W2 > 0 && C > 0);
  if (B == 0 || N == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_fmap1 && d_fmap2 && d_coords && d_corr_grad && d_fmap1_grad && d_fmap2_grad);
  if (radius != 3) return VIPE_EUNSUPPORTED;
  const int64_t total = (int64_t)B * N * H1 * W1;
  const int blocks = (int)std::min<int64_t>((total + 3) / 4, 256 * 16);
  altcorr_backward_kernel<3><<<blocks, 256, 0, as_stream(stream)>>>(d_fmap1, d_fmap2, d_coords, d_corr_grad, d_fmap1_grad,
                                                                    d_fmap2_grad, H1, W1, H2, W2, N, C, total);
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_scatter(const void* d_src, const int64_t* d_index, void* d_out, int64_t* d_arg_out, int64_t outer,
                             int64_t src_dim, int64_t inner, int64_t out_dim, int reduce, int dtype, void* stream) {
  VIPE_CHECK_ARG(outer >= 0 && src_dim >= 0 && inner >= 0 && out_dim >= 0 && reduce >= 0 && reduce <= 4);
  if (outer * src_dim * inner == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_src && d_index && d_out);
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case VIPE_F16: return run_scatter<half_t, false>(d_src, d_index, d_out, d_arg_out, outer, src_dim, inner, out_dim, reduce, s);
    case VIPE_F32: return run_scatter<float, false>(d_src, d_index, d_out, d_arg_out, outer, src_dim, inner, out_dim, reduce, s);
    case VIPE_F64: return run_scatter<double, false>(d_src, d_index, d_out, d_arg_out, outer, src_dim, inner, out_dim, reduce, s);
  }
  return VIPE_EINVAL;
}

VIPE_EXPORT int vipe_scatter_rows(const void* d_src, const int64_t* d_index, void* d_out, int64_t* d_arg_out, int64_t outer,
                                  int64_t src_dim, int64_t inner, int64_t out_dim, int reduce, int dtype, void* stream) {
  VIPE_CHECK_ARG(outer >= 0 && src_dim >= 0 && inner >= 0 && out_dim >= 0 && reduce >= 0 && reduce <= 4);
  if (outer * src_dim * inner == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_src && d_index && d_out);
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case VIPE_F16: return run_scatter<half_t, true>(d_src, d_index, d_out, d_arg_out, outer, src_dim, inner, out_dim, reduce, s);
    case VIPE_F32: return run_scatter<float, true>(d_src, d_index, d_out, d_arg_out, outer, src_dim, inner, out_dim, reduce, s);
    case VIPE_F64: return run_scatter<double, true>(d_src, d_index, d_out, d_arg_out, outer, src_dim, inner, out_dim, reduce, s);
  }
  return VIPE_EINVAL;
}

// Host-memory twin of vipe_scatter for CPU tensors (the reference's Python-level scatter_add / scatter_mean accept
// them, vipe/ext/scatter.py:24-63): the same combine rules in a sequential loop (deterministic; ties of min / max
// resolve to the LAST source row, as the device kernel's second pass does).
namespace {
template <typename T>
void scatter_host(const T* src, const int64_t* index, T* out, int64_t* arg, int64_t outer, int64_t E, int64_t K, int64_t N,
                  int reduce) {
  for (int64_t b = 0; b < outer; ++b)
    for (int64_t e = 0; e < E; ++e)
      for (int64_t k = 0; k < K; ++k) {
        const int64_t i = (b * E + e) * K + k;
        const int64_t o = b * N * K + index[i] * K + k;
        const T v = src[i], cur = out[o];
        if (reduce == 0 || reduce == 2) out[o] = cur + v;
        else if (reduce == 1) out[o] = cur * v;
        else if (reduce == 3) { if (v <= cur) { out[o] = v; arg[o] = e; } }
        else { if (v >= cur) { out[o] = v; arg[o] = e; } }
      }
}
}  // namespace

VIPE_EXPORT int vipe_scatter_host(const void* h_src, const int64_t* h_index, void* h_out, int64_t* h_arg_out, int64_t outer,
                                  int64_t src_dim, int64_t inner, int64_t out_dim, int reduce, int dtype) {
  VIPE_CHECK_ARG(outer >= 0 && src_dim >= 0 && inner >= 0 && out_dim >= 0 && reduce >= 0 && reduce <= 4);
  if (outer * src_dim * inner == 0) return VIPE_OK;
  VIPE_CHECK_ARG(h_src && h_index && h_out && (reduce < 3 || h_arg_out));
  for (int64_t i = 0; i < outer * src_dim * inner; ++i) VIPE_CHECK_ARG(h_index[i] >= 0 && h_index[i] < out_dim);
  switch (dtype) {
    case VIPE_F32: scatter_host<float>((const float*)h_src, h_index, (float*)h_out, h_arg_out, outer, src_dim, inner, out_dim, reduce); return VIPE_OK;
    case VIPE_F64: scatter_host<double>((const double*)h_src, h_index, (double*)h_out, h_arg_out, outer, src_dim, inner, out_dim, reduce); return VIPE_OK;
  }
  return VIPE_EINVAL;
}

VIPE_EXPORT int vipe_segment_mean_nhwc_f16(const void* d_src, int src_ctot, int src_coff, const int* d_order,
                                           const int* d_rowptr, void* d_out, int n_out, int64_t rows_per_item, int C,
                                           void* stream) {
  VIPE_CHECK_ARG(n_out >= 0 && rows_per_item > 0 && C > 0 && C % 8 == 0 && src_ctot % 8 == 0 && src_coff % 8 == 0);
  if (n_out == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_src && d_order && d_rowptr && d_out && n_out <= 65535);
  const int64_t n8 = rows_per_item * (C / 8);
  dim3 grid((int)std::min<int64_t>((n8 + 255) / 256, 1024), n_out);
  segment_mean_kernel<<<grid, 256, 0, as_stream(stream)>>>((const half_t*)d_src, src_ctot, src_coff, d_order, d_rowptr,
                                                           (half_t*)d_out, rows_per_item, C);
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_scatter_mean_rows_f16(const void*, const int64_t*, void*, int, int, int64_t, void*) {
  return VIPE_EUNSUPPORTED;  // superseded by vipe_segment_mean_nhwc_f16 (CSR built once per edge set)
}

VIPE_EXPORT int vipe_corr_sampler_forward(const void* d_in1, const void* d_in2, void* d_out, int B, int C, int H, int W,
                                          int kH, int kW, int patchH, int patchW, int padH, int padW, int dilH, int dilW,
                                          int dil_patchH, int dil_patchW, int dH, int dW, int dtype, void* stream) {
  VIPE_CHECK_ARG(B >= 0 && C > 0 && H > 0 && W > 0 && kH > 0 && kW > 0 && patchH > 0 && patchW > 0 && dH > 0 && dW > 0);
  const int oH = (H + 2 * padH - ((kH - 1) * dilH + 1)) / dH + 1, oW = (W + 2 * padW - ((kW - 1) * dilW + 1)) / dW + 1;
  const int64_t total = (int64_t)B * patchH * patchW * oH * oW;
  if (total <= 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_in1 && d_in2 && d_out);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
  hipStream_t s = as_stream(stream);
  if (dtype == VIPE_F32)
    corr_sampler_forward_kernel<float><<<blocks, 256, 0, s>>>((const float*)d_in1, (const float*)d_in2, (float*)d_out, B, C,
                                                              H, W, oH, oW, kH, kW, patchH, patchW, padH, padW, dilH, dilW,
                                                              dil_patchH, dil_patchW, dH, dW);
  else if (dtype == VIPE_F16)
    corr_sampler_forward_kernel<half_t><<<blocks, 256, 0, s>>>((const half_t*)d_in1, (const half_t*)d_in2, (half_t*)d_out,
                                                               B, C, H, W, oH, oW, kH, kW, patchH, patchW, padH, padW, dilH,
                                                               dilW, dil_patchH, dil_patchW, dH, dW);
  else return VIPE_EINVAL;
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_corr_sampler_backward(const void* d_in1, const void* d_in2, const void* d_grad_out, void* d_grad1,
                                           void* d_grad2, int B, int C, int H, int W, int kH, int kW, int patchH,
                                           int patchW, int padH, int padW, int dilH, int dilW, int dil_patchH,
                                           int dil_patchW, int dH, int dW, int dtype, void* stream) {
  if (dtype != VIPE_F32) return VIPE_EUNSUPPORTED;
  VIPE_CHECK_ARG(B >= 0 && C > 0 && H > 0 && W > 0);
  const int oH = (H + 2 * padH - ((kH - 1) * dilH + 1)) / dH + 1, oW = (W + 2 * padW - ((kW - 1) * dilW + 1)) / dW + 1;
  const int64_t total = (int64_t)B * patchH * patchW * oH * oW;
  if (total <= 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_in1 && d_in2 && d_grad_out && d_grad1 && d_grad2);
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
  corr_sampler_backward_kernel<<<blocks, 256, 0, as_stream(stream)>>>(
      (const float*)d_in1, (const float*)d_in2, (const float*)d_grad_out, (float*)d_grad1, (float*)d_grad2, B, C, H, W, oH,
      oW, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dil_patchH, dil_patchW, dH, dW);
  return vipe_launch_status();
}

// Host-memory twins of the correlation sampler for CPU tensors (the reference dispatches them to correlation.cpp:
// correlation_sampler.cpp:44-58; pinned by outputs of that very file, tests/golden/corr_sampler_reference.npz).  The
// accumulation order of the kernels above, one output (forward) / one gradient sample (backward) after the other -
// float32 only.
VIPE_EXPORT int vipe_corr_sampler_forward_host(const float* h_in1, const float* h_in2, float* h_out, int B, int C, int H, int W,
                                               int kH, int kW, int patchH, int patchW, int padH, int padW, int dilH, int dilW,
                                               int dil_patchH, int dil_patchW, int dH, int dW) {
  VIPE_CHECK_ARG(B >= 0 && C > 0 && H > 0 && W > 0 && kH > 0 && kW > 0 && patchH > 0 && patchW > 0 && dH > 0 && dW > 0);
  const int oH = (H + 2 * padH - ((kH - 1) * dilH + 1)) / dH + 1, oW = (W + 2 * padW - ((kW - 1) * dilW + 1)) / dW + 1;
  const int64_t total = (int64_t)B * patchH * patchW * oH * oW;
  if (total <= 0) return VIPE_OK;
  VIPE_CHECK_ARG(h_in1 && h_in2 && h_out);
  // correlation.cpp:73-74,100-101: the CPU implementation centres the patch at (patch - 1) / 2 before the dilation
  const int radH = dil_patchH * ((patchH - 1) / 2), radW = dil_patchW * ((patchW - 1) / 2);
  for (int64_t i = 0; i < total; ++i) {
    const int w = (int)(i % oW), h = (int)((i / oW) % oH);
    const int pw = (int)((i / ((int64_t)oW * oH)) % patchW), ph = (int)((i / ((int64_t)oW * oH * patchW)) % patchH);
    const int n = (int)(i / ((int64_t)oW * oH * patchW * patchH));
    const int si = -padH + h * dH, sj = -padW + w * dW;
    const int phd = ph * dil_patchH - radH, pwd = pw * dil_patchW - radW;
    float s = 0.0f;
    for (int a = 0; a < kH; ++a) {
      const int i1 = si + a * dilH, i2 = i1 + phd;
      if (i1 < 0 || i1 >= H || i2 < 0 || i2 >= H) continue;
      for (int bq = 0; bq < kW; ++bq) {
        const int j1 = sj + bq * dilW, j2 = j1 + pwd;
        if (j1 < 0 || j1 >= W || j2 < 0 || j2 >= W) continue;
        for (int c = 0; c < C; ++c)
          s += h_in1[(((int64_t)n * C + c) * H + i1) * W + j1] * h_in2[(((int64_t)n * C + c) * H + i2) * W + j2];
      }
    }
    h_out[i] = s;
  }
  return VIPE_OK;
}

VIPE_EXPORT int vipe_corr_sampler_backward_host(const float* h_in1, const float* h_in2, const float* h_grad_out, float* h_grad1,
                                                float* h_grad2, int B, int C, int H, int W, int kH, int kW, int patchH,
                                                int patchW, int padH, int padW, int dilH, int dilW, int dil_patchH,
                                                int dil_patchW, int dH, int dW) {
  VIPE_CHECK_ARG(B >= 0 && C > 0 && H > 0 && W > 0 && kH > 0 && kW > 0 && patchH > 0 && patchW > 0 && dH > 0 && dW > 0);
  const int oH = (H + 2 * padH - ((kH - 1) * dilH + 1)) / dH + 1, oW = (W + 2 * padW - ((kW - 1) * dilW + 1)) / dW + 1;
  const int64_t total = (int64_t)B * patchH * patchW * oH * oW;
  if (total <= 0) return VIPE_OK;
  VIPE_CHECK_ARG(h_in1 && h_in2 && h_grad_out && h_grad1 && h_grad2);  // gradients are accumulated into: caller zeroes
  // correlation.cpp:73-74,100-101: the CPU implementation centres the patch at (patch - 1) / 2 before the dilation
  const int radH = dil_patchH * ((patchH - 1) / 2), radW = dil_patchW * ((patchW - 1) / 2);
  for (int64_t i = 0; i < total; ++i) {
    const int w = (int)(i % oW), h = (int)((i / oW) % oH);
    const int pw = (int)((i / ((int64_t)oW * oH)) % patchW), ph = (int)((i / ((int64_t)oW * oH * patchW)) % patchH);
    const int n = (int)(i / ((int64_t)oW * oH * patchW * patchH));
    const int si = -padH + h * dH, sj = -padW + w * dW;
    const int phd = ph * dil_patchH - radH, pwd = pw * dil_patchW - radW;
    const float g = h_grad_out[i];
    for (int a = 0; a < kH; ++a) {
      const int i1 = si + a * dilH, i2 = i1 + phd;
      if (i1 < 0 || i1 >= H || i2 < 0 || i2 >= H) continue;
      for (int bq = 0; bq < kW; ++bq) {
        const int j1 = sj + bq * dilW, j2 = j1 + pwd;
        if (j1 < 0 || j1 >= W || j2 < 0 || j2 >= W) continue;
        for (int c = 0; c < C; ++c) {
          const int64_t o1 = (((int64_t)n * C + c) * H + i1) * W + j1, o2 = (((int64_t)n * C + c) * H + i2) * W + j2;
          h_grad1[o1] += g * h_in2[o2];
          h_grad2[o2] += g * h_in1[o1];
        }
      }
    }
  }
  return VIPE_OK;
}

// ---- utils_ext.nearest_neighbours (knn.cu:27-67): exact kNN by brute force.  One lane = one query with its k best in
// registers (kept ascending by insertion); the tree travels through LDS in tiles of 1024 points that all 256 queries
// of the workgroup scan (broadcast reads).  4e9 distance evaluations (a 512 x 384 query grid against 21 000 projected
// points) take a few milliseconds; the reference builds and walks a kd-tree per call.
namespace {
constexpr int NN_TILE = 1024, NN_KMAX = 8;

__global__ __launch_bounds__(256) void nearest_kernel(const float* __restrict__ query, int qdim, const float* __restrict__ tree,
                                                      int tdim, int64_t M, int64_t N, int knn, float* __restrict__ dist,
                                                      int* __restrict__ idx) {
  __shared__ float4 pts[NN_TILE];
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (q < M) {
    qx = query[q * qdim];
    if (qdim > 1) qy = query[q * qdim + 1];
    if (qdim > 2) qz = query[q * qdim + 2];
  }
  float bd[NN_KMAX];
  int bi[NN_KMAX];
#pragma unroll
  for (int k = 0; k < NN_KMAX; ++k) { bd[k] = __builtin_inff(); bi[k] = -1; }
  for (int64_t t0 = 0; t0 < N; t0 += NN_TILE) {
    const int cnt = (int)(N - t0 < NN_TILE ? N - t0 : NN_TILE);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += 256) {
      const float* p = tree + (t0 + i) * tdim;
      pts[i] = make_float4(p[0], tdim > 1 ? p[1] : 0.f, tdim > 2 ? p[2] : 0.f, 0.f);
    }
    __syncthreads();
    for (int i = 0; i < cnt; ++i) {
      const float4 p = pts[i];
      const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
      const float d = dx * dx + dy * dy + dz * dz;
      if (d < bd[NN_KMAX - 1] && (knn == NN_KMAX || d < bd[knn - 1])) {
        // insertion into the ascending list (strict <: an equal distance keeps the earlier, i.e. lower, index first)
        float cd = d;
        int ci = (int)(t0 + i);
#pragma unroll
        for (int k = 0; k < NN_KMAX; ++k) {
          if (k < knn && cd < bd[k]) {
            const float td = bd[k]; const int ti = bi[k];
            bd[k] = cd; bi[k] = ci;
            cd = td; ci = ti;
          }
        }
      }
    }
  }
  if (q < M) {
#pragma unroll
    for (int k = 0; k < NN_KMAX; ++k)
      if (k < knn) { dist[q * knn + k] = bd[k]; idx[q * knn + k] = bi[k]; }
  }
}
}  // namespace

VIPE_EXPORT int vipe_nearest_neighbours(const float* d_query, int qdim, const float* d_tree, int tdim, int64_t M,
                                                   int64_t N, int knn, float* d_dist, int* d_idx, void* stream) {
  VIPE_CHECK_ARG(M >= 0 && N >= 0 && qdim >= 1 && qdim <= 3 && tdim >= 1 && tdim <= 3 && knn >= 1 && knn <= NN_KMAX);
  VIPE_CHECK_ARG(N >= knn && N <= 0x7fffffff);
  if (M == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_query && d_tree && d_dist && d_idx);
  nearest_kernel<<<(unsigned)((M + 255) / 256), 256, 0, as_stream(stream)>>>(d_query, qdim, d_tree, tdim, M, N, knn, d_dist, d_idx);
  return vipe_launch_status();
}

// ---- [fused] the motion filter's keyframe score (motion_filter.py:103-110): per view the mean magnitude of the one-iteration
// flow (the operator's fp16 head output) over the usable pixels.  The reference: slice, cast, norm, two means, a division
// - six launches per frame; here one block per view.
namespace {
__global__ __launch_bounds__(256) void flow_score_kernel(const float* __restrict__ dw, const unsigned char* __restrict__ invalid,
                                                         float* __restrict__ score, int P) {
  const int v = blockIdx.x;
  float sf = 0.0f, sw = 0.0f;
  for (int k = threadIdx.x; k < P; k += 256) {
    const float4 d = reinterpret_cast<const float4*>(dw)[(int64_t)v * P + k];
    const float fx = (float)(half_t)d.x, fy = (float)(half_t)d.y;  // the heads are fp16 under autocast (droid_net.py:486-487)
    const float w = invalid ? (invalid[(int64_t)v * P + k] ? 0.0f : 1.0f) : 1.0f;
    sf += sqrtf(fx * fx + fy * fy) * w;
    sw += w;
  }
  __shared__ float red[2][4];
  sf = wave_sum(sf);
  sw = wave_sum(sw);
  if (lane_id() == 0) { red[0][wave_id()] = sf; red[1][wave_id()] = sw; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float f = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)P;
    const float w = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)P;
    score[v] = invalid ? f / (w + 1e-6f) : f;
  }
}
}  // namespace

VIPE_EXPORT int vipe_flow_score(const float* d_dw, const unsigned char* d_invalid, float* d_score, int n_views, int P, void* stream) {
  VIPE_CHECK_ARG(d_dw && d_score && n_views > 0 && P > 0);
  flow_score_kernel<<<n_views, 256, 0, as_stream(stream)>>>(d_dw, d_invalid, d_score, P);
  return vipe_launch_status();
}
