// Feature / context encoders of the SLAM front end: the reference's `BasicEncoder`
// (vipe/slam/networks/droid_net.py:290-370: 7x7/2 stem, three stages of two residual blocks at 1/2, 1/4, 1/8
// resolution, 1x1 output conv; instance norm for the feature net, none for the context net) run on every frame by
// `MotionFilter.check` (motion_filter.py:58-150) under fp16 autocast.
//
// One frame is ~10.6 GFLOP per encoder over activations of a few MB: the work is launch / latency bound, not MFMA
// bound, so the design minimises passes over the activations rather than chasing matrix-core utilisation:
//   * NHWC fp16 activations; every convolution is an implicit GEMM on `v_mfma_f32_16x16x32_f16` with the weights as
//     A operand (a lane ends up with 4 consecutive output channels of one pixel -> 8-byte NHWC stores);
//   * a workgroup owns 4 x 32 output pixels x 32 output channels; per 32-channel input chunk the strided input halo
//     goes to LDS once and all KS*KS taps read shifted windows of it; the A fragments of all taps of the chunk are
//     fetched (L2-resident, 1 KiB coalesced per wave and tap) before the halo barrier;
//   * instance norm never gets its own pass: the producing convolution accumulates per-(image, channel) sum and
//     sum of squares of its fp16 outputs (wave shuffle -> LDS -> one atomic per channel and workgroup) and the
//     CONSUMING convolution applies (x - mean) * rstd and the ReLU while it stages its halo ("normalise on load");
//   * the residual tail `relu(x + relu(norm(conv2)))` is one elementwise kernel for the normalised net and is fused
//     into conv2's epilogue for the context net (no statistics needed there);
//   * the output 1x1 conv writes NCHW (what the correlation-volume build and the reference API expect) and applies
//     the context net's tanh / relu split (droid_net.py:525-527) in its epilogue.
#include "common.cuh"

namespace {

typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int TR = 4, TW = 32;  // output tile: 4 rows x 32 columns, one row per wave
constexpr int CT = 32;          // output channels per workgroup
constexpr float IN_EPS = 1e-5f; // nn.InstanceNorm2d default

struct EncConvArgs {
  const half_t* x;        // [B,H,W,Cin]
  const float* in_stats;  // [B,Cin,2] (sum, sum of squares) of x, or null: x is used as it is
  float in_inv_count;     // 1 / (H*W)
  const half_t* w;        // packed [KS*KS][Cin/32][Cout][32]
  const float* bias;      // [Cout]
  const half_t* res;      // [B,Ho,Wo,Cout] or null: out = relu(res + act(conv))
  half_t* y;              // NHWC [B,Ho,Wo,Cout] or NCHW [B,Cout,Ho,Wo]
  float* out_stats;       // [B,Cout,2] or null
  int B, H, W, Cin, Cout, Ho, Wo;
  int relu, nchw, tanh_split;  // tanh_split >= 0: channels < split get tanh, the others relu
};

template <int S>
__device__ __forceinline__ int halo_off(int hpix, int hc, int chunk) {
  // 64-byte rows (32 channels); the 16-byte chunk index is XORed with column bits so that the 16 pixels a B-fragment
  // read touches spread over the LDS banks
  const int f = (S == 1) ? ((hc >> 2) & 3) : ((hc >> 3) & 3);
  return hpix * 64 + ((chunk ^ f) << 4);
}

template <int KS, int S>
__global__ __launch_bounds__(256) void enc_conv_kernel(EncConvArgs a) {
  constexpr int PAD = KS / 2;
  constexpr int HR = (TR - 1) * S + KS, HC = (TW - 1) * S + KS;
  constexpr int NPIECE = HR * HC * 4;
  constexpr int NP = (NPIECE + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char halo[HR * HC * 64];
  __shared__ float s_mean[128], s_rstd[128], s_stat[CT * 2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TR - 1) / TR;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int co0 = blockIdx.y * CT;
  const int oy0 = ty * TR, ox0 = tx * TW;
  const int nch = a.Cin >> 5;

  if (a.in_stats) {
    for (int c = tid; c < a.Cin; c += 256) {
      const float s = a.in_stats[((size_t)b * a.Cin + c) * 2], ss = a.in_stats[((size_t)b * a.Cin + c) * 2 + 1];
      const float m = s * a.in_inv_count;
      const float var = fmaxf(ss * a.in_inv_count - m * m, 0.0f);
      s_mean[c] = m;
      s_rstd[c] = rsqrtf(var + IN_EPS);
    }
  }
  if (tid < CT * 2) s_stat[tid] = 0.0f;

  float4v acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = float4v{0, 0, 0, 0};

  const int gy0 = oy0 * S - PAD, gx0 = ox0 * S - PAD;
  for (int ck = 0; ck < nch; ++ck) {
    // A fragments of every tap of this chunk: lane = (cout row l&15, k group l>>4), 16 bytes each
    half8 wv[KS * KS][2];
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap)
#pragma unroll
      for (int m = 0; m < 2; ++m)
        wv[tap][m] = *(const half8*)(a.w + ((size_t)(tap * nch + ck) * a.Cout + co0 + m * 16 + (lane & 15)) * 32 +
                                     (lane >> 4) * 8);
    // halo pieces (16 bytes = 8 channels of one halo pixel)
    half8 pv[NP];
    bool inb[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * 256;
      const int hp = p >> 2, c16 = p & 3;
      const int hr = hp / HC, hc = hp - hr * HC;
      const int gy = gy0 + hr, gx = gx0 + hc;
      inb[i] = p < NPIECE && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      pv[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
      if (inb[i]) pv[i] = *(const half8*)(a.x + ((size_t)(b * a.H + gy) * a.W + gx) * a.Cin + ck * 32 + c16 * 8);
    }
    __syncthreads();  // previous chunk's readers are done (and s_mean / s_rstd are visible)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * 256;
      if (p < NPIECE) {
        const int hp = p >> 2, c16 = p & 3;
        const int hc = hp % HC;
        half8 v = pv[i];
        if (a.in_stats && inb[i]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = ck * 32 + c16 * 8 + j;
            const half_t nrm = f2h((h2f(v[j]) - s_mean[c]) * s_rstd[c]);
            v[j] = nrm > (half_t)0 ? nrm : (half_t)0;
          }
        }
        *(half8*)(halo + halo_off<S>(hp, hc, c16)) = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap) {
      const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int hr = wave * S + ky, hc = (n * 16 + (lane & 15)) * S + kx;
        const half8 bv = *(const half8*)(halo + halo_off<S>(hr * HC + hc, hc, lane >> 4));
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[tap][m], bv, acc[m][n], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane holds couts co0 + m*16 + 4*(lane>>4) + r of pixel (oy0 + wave, ox0 + n*16 + (lane&15))
  const int oy = oy0 + wave;
  float ssum[2][4], ssq[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) ssum[m][r] = ssq[m][r] = 0.0f;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int ox = ox0 + n * 16 + (lane & 15);
    const bool ok = oy < a.Ho && ox < a.Wo;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int co = co0 + m * 16 + 4 * (lane >> 4);
      half4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[m][n][r] + a.bias[co + r];
        half_t h = f2h(v);
        if (ok) { ssum[m][r] += h2f(h); ssq[m][r] += h2f(h) * h2f(h); }
        if (a.tanh_split >= 0) h = f2h(co + r < a.tanh_split ? tanhf(h2f(h)) : fmaxf(h2f(h), 0.0f));
        else if (a.relu) h = h > (half_t)0 ? h : (half_t)0;
        o[r] = h;
      }
      if (!ok) continue;
      if (a.res) {
        const half4 rv = *(const half4*)(a.res + ((size_t)(b * a.Ho + oy) * a.Wo + ox) * a.Cout + co);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const half_t sm = f2h(h2f(rv[r]) + h2f(o[r]));
          o[r] = sm > (half_t)0 ? sm : (half_t)0;
        }
      }
      if (a.nchw) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.y[((size_t)(b * a.Cout + co + r) * a.Ho + oy) * a.Wo + ox] = o[r];
      } else {
        *(half4*)(a.y + ((size_t)(b * a.Ho + oy) * a.Wo + ox) * a.Cout + co) = o;
      }
    }
  }
  if (a.out_stats) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = ssum[m][r], q = ssq[m][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, WAVE); q += __shfl_xor(q, o, WAVE); }
        if ((lane & 15) == 0) {
          const int c = m * 16 + 4 * (lane >> 4) + r;
          atomicAdd(&s_stat[c * 2], s);
          atomicAdd(&s_stat[c * 2 + 1], q);
        }
      }
    __syncthreads();
    if (tid < CT * 2) atomicAdd(&a.out_stats[((size_t)b * a.Cout + co0) * 2 + tid], s_stat[tid]);
  }
}

// ---- stem: 7x7 stride 2 pad 3, 3 (+1 zero) -> 32 channels (droid_net.py:309).  k = tap*4 + c, 7 steps of 32.
struct EncStemArgs {
  const half_t* x4;  // [B,H,W,4] normalised image, channel 3 = 0
  const half_t* w;   // packed [7][32][32]
  const float* bias;
  half_t* y;         // [B,Ho,Wo,32]
  float* out_stats;  // [B,32,2] or null
  int B, H, W, Ho, Wo, relu;
};

__global__ __launch_bounds__(256) void enc_stem_kernel(EncStemArgs a) {
  constexpr int KS = 7, S = 2, PAD = 3;
  constexpr int HR = (TR - 1) * S + KS, HC = (TW - 1) * S + KS;  // 13 x 69
  typedef half_t half4l __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) half4l halo[HR * HC];
  __shared__ float s_stat[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TR - 1) / TR;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int oy0 = ty * TR, ox0 = tx * TW;
  if (tid < 64) s_stat[tid] = 0.0f;
  for (int p = tid; p < HR * HC; p += 256) {
    const int hr = p / HC, hc = p - hr * HC;
    const int gy = oy0 * S - PAD + hr, gx = ox0 * S - PAD + hc;
    half4l v = half4l{0, 0, 0, 0};
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = *(const half4l*)(a.x4 + ((size_t)(b * a.H + gy) * a.W + gx) * 4);
    halo[p] = v;
  }
  half8 wv[7][2];
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int m = 0; m < 2; ++m) wv[s][m] = *(const half8*)(a.w + ((size_t)(s * 32) + m * 16 + (lane & 15)) * 32 + (lane >> 4) * 8);
  __syncthreads();
  float4v acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = float4v{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    int t0 = s * 8 + (lane >> 4) * 2, t1 = t0 + 1;
    t0 = t0 > 48 ? 48 : t0;  // taps >= 49 have zero weights; keep the read inside the tile
    t1 = t1 > 48 ? 48 : t1;
    const int ky0 = t0 / 7, kx0 = t0 - ky0 * 7, ky1 = t1 / 7, kx1 = t1 - ky1 * 7;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int col = (n * 16 + (lane & 15)) * S;
      const half4l lo = halo[(wave * S + ky0) * HC + col + kx0];
      const half4l hi = halo[(wave * S + ky1) * HC + col + kx1];
      const half8 bv = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[s][m], bv, acc[m][n], 0, 0, 0);
    }
  }
  const int oy = oy0 + wave;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int co = m * 16 + 4 * (lane >> 4);
    float ssum[4] = {0, 0, 0, 0}, ssq[4] = {0, 0, 0, 0};
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int ox = ox0 + n * 16 + (lane & 15);
      const bool ok = oy < a.Ho && ox < a.Wo;
      half4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        half_t h = f2h(acc[m][n][r] + a.bias[co + r]);
        if (ok) { ssum[r] += h2f(h); ssq[r] += h2f(h) * h2f(h); }
        if (a.relu) h = h > (half_t)0 ? h : (half_t)0;
        o[r] = h;
      }
      if (ok) *(half4*)(a.y + ((size_t)(b * a.Ho + oy) * a.Wo + ox) * 32 + co) = o;
    }
    if (a.out_stats) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = ssum[r], q = ssq[r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, WAVE); q += __shfl_xor(q, o, WAVE); }
        if ((lane & 15) == 0) {
          atomicAdd(&s_stat[(co + r) * 2], s);
          atomicAdd(&s_stat[(co + r) * 2 + 1], q);
        }
      }
    }
  }
  if (a.out_stats) {
    __syncthreads();
    if (tid < 64) atomicAdd(&a.out_stats[(size_t)b * 64 + tid], s_stat[tid]);
  }
}

// ---- image normalisation: [V,3,H,W] fp32 RGB in [0,1] -> [V,H,W,4] fp16 ((x - mean) / std, droid_net.py:512-516)
__global__ __launch_bounds__(256) void enc_prep_kernel(const float* __restrict__ img, half_t* __restrict__ out, int V,
                                                        int HW) {
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  const int64_t n = (int64_t)V * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t v = i / HW, p = i - v * HW;
    half4 o;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = f2h((img[(v * 3 + c) * HW + p] - mean[c]) / stdv[c]);
    o[3] = (half_t)0;
    *(half4*)(out + i * 4) = o;
  }
}

// ---- residual tail of the normalised net: out = relu(xres + relu(IN(raw))), xres = res or IN(res) when res_stats
// is given (the strided shortcut `downsample` = conv1x1 + norm, droid_net.py:217-232); res == null: out = relu(IN(raw))
__global__ __launch_bounds__(256) void enc_finish_kernel(const half_t* __restrict__ raw, const float* __restrict__ raw_stats,
                                                          const half_t* __restrict__ res, const float* __restrict__ res_stats,
                                                          half_t* __restrict__ out, int HW, int C, float inv_count) {
  const int b = blockIdx.y;
  const int c8n = C >> 3;
  const int64_t n = (int64_t)HW * c8n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c0 = (int)(i % c8n) * 8;
    const int64_t off = ((int64_t)b * HW) * C + (i / c8n) * C + c0;
    const half8 rv = *(const half8*)(raw + off);
    half8 xv = half8{0, 0, 0, 0, 0, 0, 0, 0};
    if (res) xv = *(const half8*)(res + off);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      const float s = raw_stats[((size_t)b * C + c) * 2], q = raw_stats[((size_t)b * C + c) * 2 + 1];
      const float m = s * inv_count;
      const float rstd = rsqrtf(fmaxf(q * inv_count - m * m, 0.0f) + IN_EPS);
      half_t y = f2h((h2f(rv[j]) - m) * rstd);
      y = y > (half_t)0 ? y : (half_t)0;
      if (res) {
        half_t x = xv[j];
        if (res_stats) {
          const float s2 = res_stats[((size_t)b * C + c) * 2], q2 = res_stats[((size_t)b * C + c) * 2 + 1];
          const float m2 = s2 * inv_count;
          x = f2h((h2f(x) - m2) * rsqrtf(fmaxf(q2 * inv_count - m2 * m2, 0.0f) + IN_EPS));
        }
        y = f2h(h2f(x) + h2f(y));
        y = y > (half_t)0 ? y : (half_t)0;
      }
      o[j] = y;
    }
    *(half8*)(out + off) = o;
  }
}

template <int KS, int S>
int launch_conv(const EncConvArgs& a, hipStream_t s) {
  const int tiles = a.B * ((a.Ho + TR - 1) / TR) * ((a.Wo + TW - 1) / TW);
  enc_conv_kernel<KS, S><<<dim3(tiles, a.Cout / CT), 256, 0, s>>>(a);
  return vipe_launch_status();
}

}  // namespace

VIPE_EXPORT int vipe_enc_prep(const float* d_img, void* d_x4, int V, int H, int W, void* stream) {
  VIPE_CHECK_ARG(d_img && d_x4 && V > 0 && H > 0 && W > 0);
  const int64_t n = (int64_t)V * H * W;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  enc_prep_kernel<<<blocks, 256, 0, as_stream(stream)>>>(d_img, (half_t*)d_x4, V, H * W);
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_enc_stem(const void* d_x4, const void* d_w, const float* d_bias, void* d_y, float* d_out_stats,
                              int B, int H, int W, int relu, void* stream) {
  VIPE_CHECK_ARG(d_x4 && d_w && d_bias && d_y && B > 0 && H > 0 && W > 0);
  EncStemArgs a;
  a.x4 = (const half_t*)d_x4; a.w = (const half_t*)d_w; a.bias = d_bias; a.y = (half_t*)d_y; a.out_stats = d_out_stats;
  a.B = B; a.H = H; a.W = W; a.Ho = (H + 6 - 7) / 2 + 1; a.Wo = (W + 6 - 7) / 2 + 1; a.relu = relu;
  const int tiles = B * ((a.Ho + TR - 1) / TR) * ((a.Wo + TW - 1) / TW);
  enc_stem_kernel<<<tiles, 256, 0, as_stream(stream)>>>(a);
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_enc_conv(const void* d_x, const float* d_in_stats, const void* d_w, const float* d_bias,
                              const void* d_res, void* d_y, float* d_out_stats, int B, int H, int W, int Cin, int Cout,
                              int ksize, int stride, int relu, int nchw, int tanh_split, void* stream) {
  VIPE_CHECK_ARG(d_x && d_w && d_bias && d_y && B > 0 && H > 0 && W > 0);
  VIPE_CHECK_ARG(Cin % 32 == 0 && Cin <= 128 && Cout % CT == 0);
  VIPE_CHECK_ARG(!(d_res && nchw));
  EncConvArgs a;
  a.x = (const half_t*)d_x; a.in_stats = d_in_stats; a.in_inv_count = 1.0f / ((float)H * (float)W);
  a.w = (const half_t*)d_w; a.bias = d_bias; a.res = (const half_t*)d_res; a.y = (half_t*)d_y; a.out_stats = d_out_stats;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  const int pad = ksize / 2;
  a.Ho = (H + 2 * pad - ksize) / stride + 1;
  a.Wo = (W + 2 * pad - ksize) / stride + 1;
  a.relu = relu; a.nchw = nchw; a.tanh_split = tanh_split;
  hipStream_t s = as_stream(stream);
  if (ksize == 3 && stride == 1) return launch_conv<3, 1>(a, s);
  if (ksize == 3 && stride == 2) return launch_conv<3, 2>(a, s);
  if (ksize == 1 && stride == 2) return launch_conv<1, 2>(a, s);
  if (ksize == 1 && stride == 1) return launch_conv<1, 1>(a, s);
  return VIPE_EUNSUPPORTED;
}

VIPE_EXPORT int vipe_enc_finish(const void* d_raw, const float* d_raw_stats, const void* d_res, const float* d_res_stats,
                                void* d_out, int B, int HW, int C, void* stream) {
  VIPE_CHECK_ARG(d_raw && d_raw_stats && d_out && B > 0 && HW > 0 && C % 8 == 0);
  VIPE_CHECK_ARG(!(d_res_stats && !d_res));
  const int64_t n = (int64_t)HW * (C / 8);
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  enc_finish_kernel<<<dim3(blocks, B), 256, 0, as_stream(stream)>>>((const half_t*)d_raw, d_raw_stats, (const half_t*)d_res,
                                                                    d_res_stats, (half_t*)d_out, HW, C, 1.0f / (float)HW);
  return vipe_launch_status();
}
