// Windowed bilinear lookup into the all-pairs correlation pyramid (droid_net_ext.corr_index_*).
//
// Replaces csrc/droid_net_ext/correlation_kernels.cu:22-159 of the reference.  One lane owns one
// source pixel (b,y,x): it reads the (2r+2)^2 integer taps around floor(coords) from that pixel's
// [h2,w2] slab and keeps the (2r+1)^2 outputs in registers, so every output is written exactly once,
// coalesced along x (the reference does 4 global read-modify-writes per tap).
//
// Numerics (bit-parity contract): per output the four contributions are added in the reference's tap
// order (i = x offset outer, j = y offset inner).  For half, every product and every sum is rounded
// to half (c10::Half operators) and the bilinear weight is rounded to half first.  For float/double the
// update `corr += s * w` is one fused multiply-add (the contraction nvcc applies to that statement).
// Out-of-bounds taps are skipped in the reference; adding their +-0 products instead is bit-identical
// because an accumulator that starts at +0 can never hold -0.
#include <stdlib.h>

#include "common.cuh"

namespace {

template <typename T>
struct Acc;  // arithmetic in the storage type's rounding

template <>
struct Acc<half_t> {
  using reg = float;  // holds a half-representable value
  static __device__ __forceinline__ reg load(const half_t* p) { return (float)*p; }
  static __device__ __forceinline__ reg weight(float w) {
    // the f32 product must be ROUNDED TO f32 before the half conversion (scalar_t(dx * dy) in the reference);
    // without the barrier hipcc fuses mul + cvt into v_fma_mixlo_f16, which rounds the exact product once
    asm volatile("" : "+v"(w));
    return (float)(half_t)w;
  }
  static __device__ __forceinline__ reg madd(reg acc, reg s, reg w) {
    float prod = (float)(half_t)(s * w);  // exact product of two halves, rounded once to half
    return (float)(half_t)(acc + prod);
  }
  static __device__ __forceinline__ half_t store(reg v) { return (half_t)v; }
};
template <>
struct Acc<float> {
  using reg = float;
  static __device__ __forceinline__ reg load(const float* p) { return *p; }
  static __device__ __forceinline__ reg weight(float w) { return w; }
  static __device__ __forceinline__ reg madd(reg acc, reg s, reg w) { return __builtin_fmaf(s, w, acc); }
  static __device__ __forceinline__ float store(reg v) { return v; }
};
template <>
struct Acc<double> {
  using reg = double;
  static __device__ __forceinline__ reg load(const double* p) { return *p; }
  static __device__ __forceinline__ reg weight(float w) { return (double)w; }
  static __device__ __forceinline__ reg madd(reg acc, reg s, reg w) { return __builtin_fma(s, w, acc); }
  static __device__ __forceinline__ double store(reg v) { return v; }
};

// One pixel, one level.  R = radius (compile-time for full unrolling), RD = 2R+1.
template <typename T, int R>
__device__ __forceinline__ void lookup_pixel(const T* __restrict__ slab, int h2, int w2, float x0, float y0,
                                             T* __restrict__ out, int64_t out_stride) {
  constexpr int RD = 2 * R + 1;
  using A = Acc<T>;
  using reg = typename A::reg;
  const float fx = floorf(x0), fy = floorf(y0);
  const float dx = x0 - fx, dy = y0 - fy;
  const int bx = (int)fx - R, by = (int)fy - R;
  const reg w11 = A::weight(dx * dy);                  // tap (i,j) -> out[i-1][j-1]
  const reg w10 = A::weight(dx * (1.0f - dy));         // tap (i,j) -> out[i-1][j]
  const reg w01 = A::weight((1.0f - dx) * dy);         // tap (i,j) -> out[i][j-1]
  const reg w00 = A::weight((1.0f - dx) * (1.0f - dy));// tap (i,j) -> out[i][j]

  // taps: s[i][j] = slab[by + j][bx + i]   (i <-> x, j <-> y), zero outside the map
  reg s[RD + 1][RD + 1];
#pragma unroll
  for (int j = 0; j <= RD; ++j) {
    const int y1 = by + j;
    const bool yin = (y1 >= 0) & (y1 < h2);
    const T* row = slab + (int64_t)(yin ? y1 : 0) * w2;
#pragma unroll
    for (int i = 0; i <= RD; ++i) {
      const int x1 = bx + i;
      const bool in = yin & (x1 >= 0) & (x1 < w2);
      s[i][j] = in ? A::load(row + x1) : (reg)0;
    }
  }
#pragma unroll
  for (int a = 0; a < RD; ++a) {
#pragma unroll
    for (int b = 0; b < RD; ++b) {
      reg acc = (reg)0;
      acc = A::madd(acc, s[a][b], w00);          // visited at (i=a,   j=b)
      acc = A::madd(acc, s[a][b + 1], w01);      //            (i=a,   j=b+1)
      acc = A::madd(acc, s[a + 1][b], w10);      //            (i=a+1, j=b)
      acc = A::madd(acc, s[a + 1][b + 1], w11);  //            (i=a+1, j=b+1)
      out[(int64_t)(a * RD + b) * out_stride] = A::store(acc);
    }
  }
}

// grid: (ceil(P/256), B); coords [B,2,h1,w1]
template <typename T, int R>
__global__ __launch_bounds__(256) void corr_index_forward_kernel(const T* __restrict__ volume,
                                                                  const float* __restrict__ coords,
                                                                  T* __restrict__ corr, int h1, int w1, int h2,
                                                                  int w2) {
  const int P = h1 * w1;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= P) return;
  const float x0 = coords[((int64_t)n * 2 + 0) * P + p];
  const float y0 = coords[((int64_t)n * 2 + 1) * P + p];
  constexpr int RD = 2 * R + 1;
  const T* slab = volume + ((int64_t)n * P + p) * ((int64_t)h2 * w2);
  T* out = corr + (int64_t)n * RD * RD * P + p;
  lookup_pixel<T, R>(slab, h2, w2, x0, y0, out, P);
}

struct LevelPtrs {
  const void* p[8];
};

// grid: (ceil(P/256), B, levels); coords [B,h1,w1,2]; out [B, L*RD*RD, h1, w1], or with nhwc_stride > 0
// out [B,h1,w1,nhwc_stride] (channel = level*RD*RD + a*RD + b; channels >= L*RD*RD are zero filled)
template <typename T, int R>
__global__ __launch_bounds__(256) void corr_pyramid_lookup_kernel(LevelPtrs lv, const float* __restrict__ coords,
                                                                   T* __restrict__ out, int h1, int w1, int h2,
                                                                   int w2, int L, int nhwc_stride) {
  const int P = h1 * w1;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  const int l = blockIdx.z;
  if (p >= P) return;
  const float2 c = reinterpret_cast<const float2*>(coords)[(int64_t)n * P + p];
  const float sc = 1.0f / (float)(1 << l);  // coords / 2**l is exact (droid_net.py:78)
  const int h2l = h2 >> l, w2l = w2 >> l;
  constexpr int RD = 2 * R + 1;
  const T* slab = reinterpret_cast<const T*>(lv.p[l]) + ((int64_t)n * P + p) * ((int64_t)h2l * w2l);
  if (nhwc_stride > 0) {
    T* o = out + ((int64_t)n * P + p) * nhwc_stride + l * (RD * RD);
    lookup_pixel<T, R>(slab, h2l, w2l, c.x * sc, c.y * sc, o, 1);
    if (l == L - 1)
      for (int q = L * RD * RD; q < nhwc_stride; ++q) out[((int64_t)n * P + p) * nhwc_stride + q] = (T)0;
  } else {
    T* o = out + ((int64_t)n * L + l) * (RD * RD) * P + p;
    lookup_pixel<T, R>(slab, h2l, w2l, c.x * sc, c.y * sc, o, P);
  }
}

// ---- fp16 fast path (radius 3, level widths multiples of 8): 8 lanes per (pixel, level), lane j owns tap row j.
// Each lane fetches its 8 taps with TWO aligned 16-byte loads (the row's 16-byte chunks around floor(x) - 3) instead
// of eight 2-byte loads, extracts them with a funnel shift, gets row j+1 from its neighbour lane by shuffle and
// produces the 7 outputs out[a][j], a = 0..6, with exactly the reference's per-output addition chain.  A wave-level
// load instruction now covers 8 pixels x 8 rows x 16 B, i.e. 4x fewer cache-line visits per pixel than the
// lane-per-pixel kernel.
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// The 7 outputs out[a][j], a = 0..6, of ONE tap row j of one level, in packed fp16 arithmetic.  d[0..7]: the two aligned
// 16-byte chunks of row y = by + j (zero outside the map), sh = (bx & 7): taps t[i] = halves d[sh + i], i = 0..7; the taps
// nb[i] of row y + 1 come from lane + 1 of the 8-lane group.  out[a] = ((t[a] w00 + nb[a] w01) + t[a+1] w10) + nb[a+1] w11
// with every product and every sum rounded to half - c10::Half's operators compute in float and round to half, which
// for two half operands is the correctly rounded half product / sum (the float product of two halves is exact; the
// float sum is exact unless the exponents differ by > 12, when both roundings return the larger operand), i.e. exactly
// v_pk_mul_f16 / v_pk_add_f16.  -ffp-contract=off keeps the compiler from fusing them into v_pk_fma_f16.
// Results: o[k] = (out[2k], out[2k+1]) as packed pairs, k = 0..3 (out[7] is a by-product that no one reads).
__device__ __forceinline__ void row_outputs_pk(const unsigned (&d)[8], int sh, float w00, float w01, float w10, float w11,
                                               half2v (&o)[4]) {
  // funnel: e[k] = halves (sh + 2k, sh + 2k + 1) of the 16-half window.  Dword stages as bit-selects (v_bfi_b32; written
  // as ternaries the compiler turns the stages into a dynamically indexed scratch array), then one v_alignbit by 0 / 16
  const unsigned m1 = 0u - ((unsigned)(sh >> 1) & 1u), m2 = 0u - ((unsigned)(sh >> 2) & 1u);
  unsigned a1[7], f[5];
#pragma unroll
  for (int k = 0; k < 7; ++k) a1[k] = (d[k + 1] & m1) | (d[k] & ~m1);
#pragma unroll
  for (int k = 0; k < 5; ++k) f[k] = (a1[k + 2] & m2) | (a1[k] & ~m2);
  const unsigned hs = ((unsigned)sh & 1u) << 4;
  unsigned e[5];  // e[k] = (t[2k], t[2k+1]); e[4] = (t[8], .) only feeds the unused out[7]
#pragma unroll
  for (int k = 0; k < 4; ++k) e[k] = __builtin_amdgcn_alignbit(f[k + 1], f[k], hs);
  e[4] = 0;
  const half_t h00 = (half_t)w00, h01 = (half_t)w01, h10 = (half_t)w10, h11 = (half_t)w11;
  const half2v v00 = {h00, h00}, v01 = {h01, h01}, v10 = {h10, h10}, v11 = {h11, h11};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned se = __builtin_amdgcn_alignbyte(e[k + 1], e[k], 2);  // (t[2k+1], t[2k+2])
    const unsigned ne = __shfl_down(e[k], 1, 8), nse = __shfl_down(se, 1, 8);
    half2v acc = __builtin_bit_cast(half2v, e[k]) * v00;   // (half)(0 + p) == p
    acc = acc + __builtin_bit_cast(half2v, ne) * v01;
    acc = acc + __builtin_bit_cast(half2v, se) * v10;
    acc = acc + __builtin_bit_cast(half2v, nse) * v11;
    o[k] = acc;
  }
}

__global__ __launch_bounds__(256) void corr_pyramid_lookup_rows_kernel(LevelPtrs lv, const float* __restrict__ coords,
                                                                       half_t* __restrict__ out, int h1, int w1, int h2,
                                                                       int w2, int L, int nhwc_stride) {
  constexpr int R = 3, RD = 7;
  using A = Acc<half_t>;
  const int P = h1 * w1;
  const int j = threadIdx.x & 7;                         // tap row
  const int p = blockIdx.x * 32 + (threadIdx.x >> 3);    // pixel
  const int n = blockIdx.y, l = blockIdx.z;
  const bool pok = p < P;
  const int pc = pok ? p : P - 1;
  const float2 c = reinterpret_cast<const float2*>(coords)[(int64_t)n * P + pc];
  const float sc = 1.0f / (float)(1 << l);
  const float x0 = c.x * sc, y0 = c.y * sc;
  const int h2l = h2 >> l, w2l = w2 >> l;
  const float fx = floorf(x0), fy = floorf(y0);
  const float dx = x0 - fx, dy = y0 - fy;
  const int bx = (int)fx - R, by = (int)fy - R;
  const float w11 = A::weight(dx * dy), w10 = A::weight(dx * (1.0f - dy));
  const float w01 = A::weight((1.0f - dx) * dy), w00 = A::weight((1.0f - dx) * (1.0f - dy));

  // two aligned chunks of row y1 covering x in [8*c0, 8*c0 + 16)
  const int y1 = by + j;
  const int c0 = bx >> 3, sh = bx & 7;
  const int nchunks = w2l >> 3;
  const bool rowok = (y1 >= 0) & (y1 < h2l);
  const half_t* slab = reinterpret_cast<const half_t*>(lv.p[l]) + ((int64_t)n * P + pc) * ((int64_t)h2l * w2l);
  const uint4v* rowp = reinterpret_cast<const uint4v*>(slab + (int64_t)(rowok ? y1 : 0) * w2l);
  uint4v lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
  if (rowok & (c0 >= 0) & (c0 < nchunks)) lo = rowp[c0];
  if (rowok & (c0 + 1 >= 0) & (c0 + 1 < nchunks)) hi = rowp[c0 + 1];
  const unsigned d[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  half2v o[4];
  row_outputs_pk(d, sh, w00, w01, w10, w11, o);
  if (!pok || j >= RD) return;
#pragma unroll
  for (int a = 0; a < RD; ++a) {
    const int ch = l * (RD * RD) + a * RD + j;
    const half_t v = o[a >> 1][a & 1];
    if (nhwc_stride > 0) out[((int64_t)n * P + p) * nhwc_stride + ch] = v;
    else out[((int64_t)n * L * (RD * RD) + ch) * P + p] = v;
  }
  if (nhwc_stride > 0 && l == L - 1 && j == 0)
    for (int qq = L * RD * RD; qq < nhwc_stride; ++qq) out[((int64_t)n * P + p) * nhwc_stride + qq] = (half_t)0;
}

// ---- channels-last, 4 levels in one workgroup: the 8 loads of a lane (2 per level) are issued back to back (4x the
// memory-level parallelism of the per-level launch) and the pixel's 196 + 4 channels are staged in LDS so that the
// output leaves as full 16-byte stores, 400 contiguous bytes per pixel (the per-level kernel writes 2-byte pieces:
// 437 MB of HBM writes for 339 MB of output at E = 276).  Same taps, weights and addition chains as above.
constexpr int LK4_PITCH = 208;  // halves per staged pixel (16-byte aligned rows)

__global__ __launch_bounds__(256) void corr_pyramid_lookup_rows4_kernel(LevelPtrs lv, const float* __restrict__ coords,
                                                                        half_t* __restrict__ out, int h1, int w1, int h2,
                                                                        int w2, int nhwc_stride) {
  constexpr int R = 3, RD = 7, L = 4;
  using A = Acc<half_t>;
  __shared__ __align__(16) half_t stage[32 * LK4_PITCH];
  const int P = h1 * w1;
  const int j = threadIdx.x & 7;       // tap row
  const int pl = threadIdx.x >> 3;     // pixel of the workgroup
  const int p = blockIdx.x * 32 + pl;
  const int n = blockIdx.y;
  const bool pok = p < P;
  const int pc = pok ? p : P - 1;
  const float2 c = reinterpret_cast<const float2*>(coords)[(int64_t)n * P + pc];
  uint4v lo[L], hi[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float sc = 1.0f / (float)(1 << l);
    const int h2l = h2 >> l, w2l = w2 >> l;
    const int bx = (int)floorf(c.x * sc) - R, by = (int)floorf(c.y * sc) - R;
    const int y1 = by + j, c0 = bx >> 3, nchunks = w2l >> 3;
    const bool rowok = (y1 >= 0) & (y1 < h2l);
    const half_t* slab = reinterpret_cast<const half_t*>(lv.p[l]) + ((int64_t)n * P + pc) * ((int64_t)h2l * w2l);
    const uint4v* rowp = reinterpret_cast<const uint4v*>(slab + (int64_t)(rowok ? y1 : 0) * w2l);
    lo[l] = uint4v{0, 0, 0, 0};
    hi[l] = uint4v{0, 0, 0, 0};
    if (rowok & (c0 >= 0) & (c0 < nchunks)) lo[l] = rowp[c0];
    if (rowok & (c0 + 1 >= 0) & (c0 + 1 < nchunks)) hi[l] = rowp[c0 + 1];
  }
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float sc = 1.0f / (float)(1 << l);
    const float x0 = c.x * sc, y0 = c.y * sc;
    const float fx = floorf(x0), fy = floorf(y0);
    const float dx = x0 - fx, dy = y0 - fy;
    const int sh = ((int)fx - R) & 7;
    const float w11 = A::weight(dx * dy), w10 = A::weight(dx * (1.0f - dy));
    const float w01 = A::weight((1.0f - dx) * dy), w00 = A::weight((1.0f - dx) * (1.0f - dy));
    const unsigned d[8] = {lo[l][0], lo[l][1], lo[l][2], lo[l][3], hi[l][0], hi[l][1], hi[l][2], hi[l][3]};
    half2v o[4];
    row_outputs_pk(d, sh, w00, w01, w10, w11, o);
    if (j < RD) {
#pragma unroll
      for (int a = 0; a < RD; ++a) stage[pl * LK4_PITCH + l * (RD * RD) + a * RD + j] = o[a >> 1][a & 1];
    }
  }
  if (j < 4) stage[pl * LK4_PITCH + L * RD * RD + j] = (half_t)0;
  __syncthreads();
  // 16-byte pieces of the staged pixels: nhwc_stride / 8 per pixel
  const int cpp = nhwc_stride >> 3;
  for (int i = threadIdx.x; i < 32 * cpp; i += 256) {
    const int pq = i / cpp, ck = i % cpp;
    const int pg = blockIdx.x * 32 + pq;
    if (pg < P)
      *reinterpret_cast<uint4v*>(out + ((int64_t)n * P + pg) * nhwc_stride + ck * 8) =
          *reinterpret_cast<const uint4v*>(stage + pq * LK4_PITCH + ck * 8);
  }
}

// ---- lookup fused with the correlation encoder's 1x1 convolution (droid_net.py:436-437 first layer, 196 -> Cout,
// + bias + activation): the 196 looked-up channels of 32 pixels are staged in LDS as above and consumed right there
// as the B operand of v_mfma_f32_16x16x32_f16; the [E,h,w,200] intermediate (339 MB written and read back per update
// at E = 276) never exists.  Persistent workgroups: each wave keeps its 32 output channels x 224 k of the packed
// weights in registers (loaded once), so only the gather and the 256-byte output rows touch memory.
constexpr int LKC_PITCH = 232;   // halves per staged pixel: 224 k + 8 (row stride 464 B spreads the 16-lane b128 reads)
constexpr int LKC_OPITCH = 136;  // halves per staged output pixel (128 couts + 8)

typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <bool BLK>  // BLK: levels in VIPE_PYRAMID_BLOCKED (padded grid), else the reference layout
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void corr_lookup_conv_kernel(
    LevelPtrs lv, const float* __restrict__ coords, const half_t* __restrict__ wpk, const float* __restrict__ bias,
    half_t* __restrict__ out, int out_ctot, int out_coff, int h1, int w1, int h2, int w2, int B, int cout_pad, int act,
    const int* __restrict__ slots) {
  constexpr int R = 3, RD = 7, L = 4;
  using A = Acc<half_t>;
  __shared__ __align__(16) half_t stage[32 * LKC_PITCH];
  __shared__ __align__(16) half_t ostage[32 * LKC_OPITCH];
  const int P = h1 * w1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = tid & 7, pl = tid >> 3;
  const int l16 = lane & 15, kg = lane >> 4;
  const int gpp = (P + 31) / 32;
  const int ngroups = B * gpp;
  // padded blocked layout (include/vipe_amd.h): G groups of 64 source pixels, S strips of 32 columns, R row groups of 4
  const int bG = (P + 63) >> 6, bS = (w2 + 31) >> 5, bR = ((h2 + 7) >> 3) << 1;
  // this wave's weights: couts 32 wave + 16 i + l16, k = 32 s + 8 kg .. + 7  (packed [k / 64][cout_pad][64])
  half8v af[2][7];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      const int k = 32 * s + 8 * kg, r = 32 * wave + 16 * i + l16;
      af[i][s] = *reinterpret_cast<const half8v*>(wpk + ((int64_t)(k >> 6) * cout_pad + r) * 64 + (k & 63));
    }
  float4 bv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) bv[i] = *reinterpret_cast<const float4*>(bias + 32 * wave + 16 * i + 4 * kg);
  // zero the k padding of the staged rows once (columns 196..231 are never written by the lookup)
  for (int i = tid; i < 32 * (LKC_PITCH - L * RD * RD); i += 256)
    stage[(i / (LKC_PITCH - L * RD * RD)) * LKC_PITCH + L * RD * RD + i % (LKC_PITCH - L * RD * RD)] = (half_t)0;

  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int n = grp / gpp, p = (grp % gpp) * 32 + pl;
    const bool pok = p < P;
    const int pc = pok ? p : P - 1;
    const float2 c = reinterpret_cast<const float2*>(coords)[(int64_t)n * P + pc];
    const int ns = slots ? slots[n] : n;  // pooled pyramids: edge n lives in slot slots[n] of the level buffers
    uint4v lo[L], hi[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float sc = 1.0f / (float)(1 << l);
      const int h2l = h2 >> l, w2l = w2 >> l;
      const int bx = (int)floorf(c.x * sc) - R, by = (int)floorf(c.y * sc) - R;
      const int y1 = by + j, c0 = bx >> 3;
      // 8-column pieces of a row: whole ones in the reference layout; in the blocked layout the last one may be partial
      // (columns >= w2l stored as zero by the build kernel)
      const int nchunks = BLK ? (w2l + 7) >> 3 : w2l >> 3;
      const bool rowok = (y1 >= 0) & (y1 < h2l);
      lo[l] = uint4v{0, 0, 0, 0};
      hi[l] = uint4v{0, 0, 0, 0};
#ifdef VIPE_LOOKUP_SKIP_LEVEL  // measurement builds only (scratch/build_variant_src.sh): what the loads of one level cost
      if (l == VIPE_LOOKUP_SKIP_LEVEL) continue;
#endif
      if (BLK && l < 2) {
        // VIPE_PYRAMID_BLOCKED (include/vipe_amd.h): 16-byte piece (row y1, columns 8 c ..) of source pixel pc lives in
        // run ((x-strip) * (R >> l) + y1 / 4) of the pixel's group of 64, tile c % T, tile row y1 % 4
        const int T = 4 >> l, rgs = bR >> l;
        const int64_t eg = (int64_t)ns * bG + (pc >> 6);
        const half_t* base = reinterpret_cast<const half_t*>(lv.p[l]);
        const int yr = rowok ? y1 : 0;
        auto piece = [&](int cc) {
          const int64_t run = eg * ((int64_t)rgs * bS) + (cc / T) * rgs + (yr >> 2);
          return *reinterpret_cast<const uint4v*>(base + (run * 64 + (pc & 63)) * (T * 32) + (cc % T) * 32 + (yr & 3) * 8);
        };
        if (rowok & (c0 >= 0) & (c0 < nchunks)) lo[l] = piece(c0);
        if (rowok & (c0 + 1 >= 0) & (c0 + 1 < nchunks)) hi[l] = piece(c0 + 1);
      } else {
        // one slab per source pixel: [h >> l][w >> l] (reference), or levels 2 / 3 of the blocked layout - a slab for every
        // pixel of every group of 64, R >> (l - 2) rows of 8 S / round_up(4 S, 8) entries
        const int hlp = BLK ? bR >> (l - 2) : h2l, wlp = BLK ? (l == 2 ? 8 * bS : ((4 * bS + 7) & ~7)) : w2l;
        const half_t* slab = reinterpret_cast<const half_t*>(lv.p[l]) + ((int64_t)ns * (BLK ? bG * 64 : P) + pc) * ((int64_t)hlp * wlp);
        const uint4v* rowp = reinterpret_cast<const uint4v*>(slab + (int64_t)(rowok ? y1 : 0) * wlp);
        if (rowok & (c0 >= 0) & (c0 < nchunks)) lo[l] = rowp[c0];
        if (rowok & (c0 + 1 >= 0) & (c0 + 1 < nchunks)) hi[l] = rowp[c0 + 1];
      }
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float sc = 1.0f / (float)(1 << l);
      const float x0 = c.x * sc, y0 = c.y * sc;
      const float fx = floorf(x0), fy = floorf(y0);
      const float dx = x0 - fx, dy = y0 - fy;
      const int sh = ((int)fx - R) & 7;
      const float w11 = A::weight(dx * dy), w10 = A::weight(dx * (1.0f - dy));
      const float w01 = A::weight((1.0f - dx) * dy), w00 = A::weight((1.0f - dx) * (1.0f - dy));
      const unsigned d[8] = {lo[l][0], lo[l][1], lo[l][2], lo[l][3], hi[l][0], hi[l][1], hi[l][2], hi[l][3]};
      half2v o[4];
      row_outputs_pk(d, sh, w00, w01, w10, w11, o);
      if (j < RD) {
#pragma unroll
        for (int a = 0; a < RD; ++a) stage[pl * LKC_PITCH + l * (RD * RD) + a * RD + j] = o[a >> 1][a & 1];
      }
    }
    __syncthreads();
    // ---- 1x1 convolution of the 32 staged pixels: D[cout, pixel]
    float4v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) acc[i][jj] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const half8v xf = *reinterpret_cast<const half8v*>(stage + (16 * jj + l16) * LKC_PITCH + 32 * s + 8 * kg);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][s], xf, acc[i][jj], 0, 0, 0);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        half4v hv;
        const float v0 = acc[i][jj][0] + bv[i].x, v1 = acc[i][jj][1] + bv[i].y;
        const float v2 = acc[i][jj][2] + bv[i].z, v3 = acc[i][jj][3] + bv[i].w;
        hv[0] = (half_t)(act == VIPE_ACT_RELU ? fmaxf(v0, 0.0f) : v0);
        hv[1] = (half_t)(act == VIPE_ACT_RELU ? fmaxf(v1, 0.0f) : v1);
        hv[2] = (half_t)(act == VIPE_ACT_RELU ? fmaxf(v2, 0.0f) : v2);
        hv[3] = (half_t)(act == VIPE_ACT_RELU ? fmaxf(v3, 0.0f) : v3);
        *reinterpret_cast<half4v*>(ostage + (16 * jj + l16) * LKC_OPITCH + 32 * wave + 16 * i + 4 * kg) = hv;
      }
    __syncthreads();
    // 256 contiguous bytes per pixel: 2 x 16 B per thread
    {
      const int pq = tid >> 3, ck = tid & 7;
      const int pg = (grp % gpp) * 32 + pq;
      if (pg < P) {
        half_t* dst = out + ((int64_t)n * P + pg) * out_ctot + out_coff;
        *reinterpret_cast<uint4v*>(dst + ck * 8) = *reinterpret_cast<const uint4v*>(ostage + pq * LKC_OPITCH + ck * 8);
        *reinterpret_cast<uint4v*>(dst + 64 + ck * 8) = *reinterpret_cast<const uint4v*>(ostage + pq * LKC_OPITCH + 64 + ck * 8);
      }
    }
  }
}

// adjoint: each lane owns its pixel's slab, so plain stores into a zero-filled gradient are race free.
template <typename T, int R>
__global__ __launch_bounds__(256) void corr_index_backward_kernel(const float* __restrict__ coords,
                                                                   const T* __restrict__ corr_grad,
                                                                   T* __restrict__ volume_grad, int h1, int w1,
                                                                   int h2, int w2) {
  constexpr int RD = 2 * R + 1;
  using A = Acc<T>;
  using reg = typename A::reg;
  const int P = h1 * w1;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (p >= P) return;
  const float x0 = coords[((int64_t)n * 2 + 0) * P + p];
  const float y0 = coords[((int64_t)n * 2 + 1) * P + p];
  const float fx = floorf(x0), fy = floorf(y0);
  const float dx = x0 - fx, dy = y0 - fy;
  const int bx = (int)fx - R, by = (int)fy - R;
  const reg w11 = A::weight(dx * dy), w10 = A::weight(dx * (1.0f - dy));
  const reg w01 = A::weight((1.0f - dx) * dy), w00 = A::weight((1.0f - dx) * (1.0f - dy));
  const T* g = corr_grad + (int64_t)n * RD * RD * P + p;
  T* slab = volume_grad + ((int64_t)n * P + p) * ((int64_t)h2 * w2);
  reg cg[RD][RD];
#pragma unroll
  for (int a = 0; a < RD; ++a)
#pragma unroll
    for (int b = 0; b < RD; ++b) cg[a][b] = A::load(g + (int64_t)(a * RD + b) * P);
#pragma unroll
  for (int i = 0; i <= RD; ++i) {
#pragma unroll
    for (int j = 0; j <= RD; ++j) {
      const int x1 = bx + i, y1 = by + j;
      if ((y1 >= 0) & (y1 < h2) & (x1 >= 0) & (x1 < w2)) {
        reg acc = (reg)0;  // correlation_kernels.cu:100-107 order
        if (i > 0 && j > 0) acc = A::madd(acc, cg[i - 1][j - 1], w11);
        if (i > 0 && j < RD) acc = A::madd(acc, cg[i - 1][j], w10);
        if (i < RD && j > 0) acc = A::madd(acc, cg[i][j - 1], w01);
        if (i < RD && j < RD) acc = A::madd(acc, cg[i][j], w00);
        slab[(int64_t)y1 * w2 + x1] = A::store(acc);
      }
    }
  }
}

template <typename T>
int launch_fwd(const void* vol, const float* coords, void* corr, int B, int h1, int w1, int h2, int w2, int r,
               hipStream_t s) {
  dim3 grid((h1 * w1 + 255) / 256, B), block(256);
  switch (r) {
    case 1: corr_index_forward_kernel<T, 1><<<grid, block, 0, s>>>((const T*)vol, coords, (T*)corr, h1, w1, h2, w2); break;
    case 2: corr_index_forward_kernel<T, 2><<<grid, block, 0, s>>>((const T*)vol, coords, (T*)corr, h1, w1, h2, w2); break;
    case 3: corr_index_forward_kernel<T, 3><<<grid, block, 0, s>>>((const T*)vol, coords, (T*)corr, h1, w1, h2, w2); break;
    case 4: corr_index_forward_kernel<T, 4><<<grid, block, 0, s>>>((const T*)vol, coords, (T*)corr, h1, w1, h2, w2); break;
    default: return VIPE_EUNSUPPORTED;
  }
  return vipe_launch_status();
}

template <typename T>
int launch_bwd(const float* coords, const void* cg, void* vg, int B, int h1, int w1, int h2, int w2, int r,
               hipStream_t s) {
  hipError_t e = hipMemsetAsync(vg, 0, sizeof(T) * (size_t)B * h1 * w1 * h2 * w2, s);
  if (e != hipSuccess) return (int)e;
  dim3 grid((h1 * w1 + 255) / 256, B), block(256);
  switch (r) {
    case 1: corr_index_backward_kernel<T, 1><<<grid, block, 0, s>>>(coords, (const T*)cg, (T*)vg, h1, w1, h2, w2); break;
    case 2: corr_index_backward_kernel<T, 2><<<grid, block, 0, s>>>(coords, (const T*)cg, (T*)vg, h1, w1, h2, w2); break;
    case 3: corr_index_backward_kernel<T, 3><<<grid, block, 0, s>>>(coords, (const T*)cg, (T*)vg, h1, w1, h2, w2); break;
    case 4: corr_index_backward_kernel<T, 4><<<grid, block, 0, s>>>(coords, (const T*)cg, (T*)vg, h1, w1, h2, w2); break;
    default: return VIPE_EUNSUPPORTED;
  }
  return vipe_launch_status();
}

template <typename T>
int launch_pyr(const LevelPtrs& lv, const float* coords, void* out, int B, int h1, int w1, int h2, int w2, int L,
               int r, int nhwc_stride, hipStream_t s) {
  dim3 grid((h1 * w1 + 255) / 256, B, L), block(256);
  switch (r) {
    case 3: corr_pyramid_lookup_kernel<T, 3><<<grid, block, 0, s>>>(lv, coords, (T*)out, h1, w1, h2, w2, L, nhwc_stride); break;
    case 4: corr_pyramid_lookup_kernel<T, 4><<<grid, block, 0, s>>>(lv, coords, (T*)out, h1, w1, h2, w2, L, nhwc_stride); break;
    default: return VIPE_EUNSUPPORTED;
  }
  return vipe_launch_status();
}

}  // namespace

VIPE_EXPORT int vipe_corr_index_forward(const void* d_volume, const float* d_coords, void* d_corr, int B, int h1,
                                        int w1, int h2, int w2, int radius, int dtype, void* stream) {
  VIPE_CHECK_ARG(B >= 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0 && B <= 65535);
  if (B == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_volume && d_coords && d_corr);
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case VIPE_F16: return launch_fwd<half_t>(d_volume, d_coords, d_corr, B, h1, w1, h2, w2, radius, s);
    case VIPE_F32: return launch_fwd<float>(d_volume, d_coords, d_corr, B, h1, w1, h2, w2, radius, s);
    case VIPE_F64: return launch_fwd<double>(d_volume, d_coords, d_corr, B, h1, w1, h2, w2, radius, s);
  }
  return VIPE_EINVAL;
}

VIPE_EXPORT int vipe_corr_index_backward(const float* d_coords, const void* d_corr_grad, void* d_volume_grad, int B,
                                         int h1, int w1, int h2, int w2, int radius, int dtype, void* stream) {
  VIPE_CHECK_ARG(B >= 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0 && B <= 65535);
  if (B == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_coords && d_corr_grad && d_volume_grad);
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case VIPE_F16: return launch_bwd<half_t>(d_coords, d_corr_grad, d_volume_grad, B, h1, w1, h2, w2, radius, s);
    case VIPE_F32: return launch_bwd<float>(d_coords, d_corr_grad, d_volume_grad, B, h1, w1, h2, w2, radius, s);
    case VIPE_F64: return launch_bwd<double>(d_coords, d_corr_grad, d_volume_grad, B, h1, w1, h2, w2, radius, s);
  }
  return VIPE_EINVAL;
}

static int pyramid_lookup_impl(const void* const* h_levels, const float* d_coords, void* d_out, int B, int h1, int w1,
                               int h2, int w2, int num_levels, int radius, int dtype, int nhwc_stride, void* stream) {
  VIPE_CHECK_ARG(num_levels >= 1 && num_levels <= 8 && B >= 0 && B <= 65535);
  VIPE_CHECK_ARG((h2 >> (num_levels - 1)) >= 1 && (w2 >> (num_levels - 1)) >= 1);
  if (B == 0) return VIPE_OK;
  VIPE_CHECK_ARG(h_levels && d_coords && d_out);
  LevelPtrs lv;
  for (int i = 0; i < num_levels; ++i) {
    VIPE_CHECK_ARG(h_levels[i]);
    lv.p[i] = h_levels[i];
  }
  hipStream_t s = as_stream(stream);
  if (dtype == VIPE_F16 && radius == 3 && ((w2 >> (num_levels - 1)) & 7) == 0) {
    if (num_levels == 4 && nhwc_stride == 200) {
      corr_pyramid_lookup_rows4_kernel<<<dim3((h1 * w1 + 31) / 32, B), 256, 0, s>>>(lv, d_coords, (half_t*)d_out, h1, w1,
                                                                                   h2, w2, nhwc_stride);
      return vipe_launch_status();
    }
    dim3 grid((h1 * w1 + 31) / 32, B, num_levels);
    corr_pyramid_lookup_rows_kernel<<<grid, 256, 0, s>>>(lv, d_coords, (half_t*)d_out, h1, w1, h2, w2, num_levels,
                                                          nhwc_stride);
    return vipe_launch_status();
  }
  switch (dtype) {
    case VIPE_F16: return launch_pyr<half_t>(lv, d_coords, d_out, B, h1, w1, h2, w2, num_levels, radius, nhwc_stride, s);
    case VIPE_F32: return launch_pyr<float>(lv, d_coords, d_out, B, h1, w1, h2, w2, num_levels, radius, nhwc_stride, s);
  }
  return VIPE_EINVAL;
}

VIPE_EXPORT int vipe_corr_pyramid_lookup(const void* const* h_levels, const float* d_coords, void* d_out, int B,
                                         int h1, int w1, int h2, int w2, int num_levels, int radius, int dtype,
                                         void* stream) {
  return pyramid_lookup_impl(h_levels, d_coords, d_out, B, h1, w1, h2, w2, num_levels, radius, dtype, 0, stream);
}

VIPE_EXPORT int vipe_corr_pyramid_lookup_nhwc(const void* const* h_levels, const float* d_coords, void* d_out, int B,
                                              int h1, int w1, int h2, int w2, int num_levels, int radius, int dtype,
                                              int channel_stride, void* stream) {
  const int rd = 2 * radius + 1;
  VIPE_CHECK_ARG(channel_stride >= num_levels * rd * rd);
  return pyramid_lookup_impl(h_levels, d_coords, d_out, B, h1, w1, h2, w2, num_levels, radius, dtype, channel_stride,
                             stream);
}

VIPE_EXPORT int vipe_corr_lookup_conv1x1(const void* const* h_levels, const float* d_coords, const void* d_w_packed,
                                         const float* d_bias, void* d_out, int out_ctot, int out_coff, int B, int h1,
                                         int w1, int h2, int w2, int Cout, int act, const int* d_slots, int layout,
                                         void* stream) {
  VIPE_CHECK_ARG(B >= 0 && h1 > 0 && w1 > 0 && h2 > 0 && w2 > 0);
  if (B == 0) return VIPE_OK;
  VIPE_CHECK_ARG(h_levels && d_coords && d_w_packed && d_bias && d_out);
  VIPE_CHECK_ARG(act == VIPE_ACT_NONE || act == VIPE_ACT_RELU);
  VIPE_CHECK_ARG(out_ctot % 8 == 0 && out_coff % 8 == 0 && out_coff + Cout <= out_ctot);
  VIPE_CHECK_ARG(layout == VIPE_PYRAMID_REFERENCE || layout == VIPE_PYRAMID_BLOCKED);
  // 4 levels, radius 3, fp16 volume, 128 output channels.  Reference layout: the coarsest level must still have
  // 8-element row chunks (16-byte loads); the blocked layout pads its rows itself: any grid with a non-empty level 3
  if (Cout != 128 || (h2 >> 3) < 1 || (w2 >> 3) < 1) return VIPE_EUNSUPPORTED;
  if (layout == VIPE_PYRAMID_REFERENCE && ((w2 >> 3) & 7) != 0) return VIPE_EUNSUPPORTED;
  if (layout == VIPE_PYRAMID_BLOCKED) VIPE_CHECK_ARG(h1 == h2 && w1 == w2);
  LevelPtrs lv;
  for (int i = 0; i < 4; ++i) {
    VIPE_CHECK_ARG(h_levels[i]);
    lv.p[i] = h_levels[i];
  }
  const int64_t ngroups = (int64_t)B * ((h1 * w1 + 31) / 32);
  const int blocks = (int)std::min<int64_t>(ngroups, 256 * 4);
  if (layout == VIPE_PYRAMID_BLOCKED)
    corr_lookup_conv_kernel<true><<<blocks, 256, 0, as_stream(stream)>>>(lv, d_coords, (const half_t*)d_w_packed, d_bias,
                                                                         (half_t*)d_out, out_ctot, out_coff, h1, w1, h2, w2,
                                                                         B, 128, act, d_slots);
  else
    corr_lookup_conv_kernel<false><<<blocks, 256, 0, as_stream(stream)>>>(lv, d_coords, (const half_t*)d_w_packed, d_bias,
                                                                          (half_t*)d_out, out_ctot, out_coff, h1, w1, h2, w2,
                                                                          B, 128, act, d_slots);
  return vipe_launch_status();
}
