// NHWC fp16 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x16_f16), fp32 accumulate,
// with the flow-update operator's epilogues fused in (bias, per-image additive term, activation, GRU gates,
// global-context reduction, flow / confidence heads, eta).
//
// This is the MI355X answer to the cuDNN autocast convolutions of UpdateModule (vipe/slam/networks/
// droid_net.py:432-499) - the ~14 GFLOP/edge that dominate the update iteration.
//
// Mapping.  GEMM  D[cout, pixel] = sum_k Wp[cout, k] * X[k, pixel],  k = (tap, cin).
//   * weights are the MFMA "A" operand (rows = cout), activations the "B" operand (cols = pixel), so a lane's
//     accumulator registers hold 4 CONSECUTIVE output channels of one pixel -> 8-byte NHWC stores.
//   * workgroup = 4 waves (256 lanes), tile BMC couts x BNP(=128) pixels, K step 64; both operand tiles live in
//     LDS as [row][64 k] halves (128-byte rows) with the 16-byte chunk index XOR-swizzled by (row>>1)&7, which
//     makes the 8-lane ds_write_b128 groups and the 16-lane ds_read_b128 groups bank-conflict free.
//   * global -> register -> LDS staging, double buffered: the loads of K-step t+1 are issued before the MFMAs
//     of step t and written to the other LDS buffer after them (one barrier per K-step).
//   * activations: 8 consecutive lanes fetch the 128 contiguous bytes (64 channels) of one input pixel, so the
//     gather of a 3x3 tap is coalesced in 128-byte segments; out-of-image taps and channels >= Cin read as 0.
//   * the channel range may be split over two source tensors (the GRU's [r*net | inp, corr, flow] input), so
//     the concatenations of the reference are never materialised.
#include <stdlib.h>

#include <type_traits>

#include "common.cuh"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

enum Epi { EPI_PLAIN = 0, EPI_GLO = 1, EPI_ZR = 2, EPI_Q = 3, EPI_HEADS = 4, EPI_ETA = 5, EPI_PARTIAL = 6 };

struct ConvArgs {
  const half_t* x0; int x0_ctot, x0_coff;  // input channels [0, split)
  const half_t* x1; int x1_ctot, x1_coff;  // input channels [split, Cin)
  int split;
  const half_t* w;      // packed [K_pad/64][Cout_pad][64]
  const float* bias;    // [Cout_pad] (zero padded)
  const float* extra;   // [B][extra_stride] per-image additive term or null
  int extra_stride, extra_off;
  half_t* y; int y_ctot, y_coff;
  int B, H, W, Cin, Cin_pad, Cout, Cout_pad, KH, KW, nsteps;
  int act, epi;
  // epilogue operands
  const half_t* net; int net_ctot, net_coff;  // hidden state [M, .] (GLO, ZR, Q)
  const half_t* zbuf;                          // [M,128] (Q)
  half_t* y2; int y2_ctot, y2_coff;            // second output (ZR: r*net)
  float* fout;                                 // GLO: glo_sum [B,Cout]; HEADS: [M,4]; ETA: [M]
  const half_t* accinit; int ai_ctot, ai_coff;  // optional initial accumulator [M, ai_ctot] (a partial conv sum)
  int ai_f32;                                   // ... held as fp32 (the output of an EPI_PARTIAL launch) instead of fp16
  // flat tiling (FLAT kernels, any H x W): a tile is 256 consecutive positions q = y * Wp + x of the image padded to
  // pitch Wp = W + pad columns (zeros); filled by launch_conv
  int f_wp, f_hwp, f_tiles, f_pieces, f_xbytes, f_p1lo, f_p1hi;
  unsigned f_magic;  // ceil(2^32 / f_wp)
};

constexpr int BNP = 128;  // pixels per tile
constexpr int BK = 64;    // k per step

__device__ __forceinline__ int swz(int row, int k8) { return row * 128 + ((k8 ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case VIPE_ACT_RELU: return fmaxf(v, 0.0f);
    case VIPE_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    case VIPE_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// BMC: couts per tile (128 / 64 / 32); WM x WN waves; SMALLCIN: Cin == 4 (k = tap*4 + c)
template <int BMC, int WAVES_M, int WAVES_N, bool SMALLCIN>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
  constexpr int TM = BMC / (WAVES_M * 32);
  constexpr int TN = BNP / (WAVES_N * 32);
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int W_ITEMS = BMC * 8 / 256;  // 16-byte weight chunks per lane per K-step
  constexpr int X_ITEMS = BNP * 8 / 256;  // = 4

  extern __shared__ __align__(16) unsigned char lds[];
  // [2][BMC*128] weights, then [2][BNP*128] activations
  unsigned char* ldsW = lds;
  unsigned char* ldsX = lds + 2 * BMC * 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int64_t M = (int64_t)a.B * a.H * a.W;
  const int64_t pix0 = (int64_t)blockIdx.x * BNP;
  const int cout0 = blockIdx.y * BMC;
  const int HW = a.H * a.W;

  // ---- per-lane load descriptors.  item i = tid + 256*j  ->  row = i / 8, chunk = i % 8
  const int k8 = tid & 7;
  int xe[X_ITEMS], xy[X_ITEMS], xx[X_ITEMS];
  bool xvalid[X_ITEMS];
#pragma unroll
  for (int j = 0; j < X_ITEMS; ++j) {
    const int64_t m = pix0 + (tid >> 3) + 32 * j;
    xvalid[j] = m < M;
    const int64_t mm = xvalid[j] ? m : 0;
    xe[j] = (int)(mm / HW);
    const int rem = (int)(mm % HW);
    xy[j] = rem / a.W;
    xx[j] = rem % a.W;
  }
  const int ph = a.KH / 2, pw = a.KW / 2;
  const int csteps = a.Cin_pad / BK;  // K-steps per tap (generic mode)

  half8 wreg[W_ITEMS], xreg[X_ITEMS];

  auto issue_loads = [&](int s) {
    // weights: block s of the packed tensor is [Cout_pad][64] halves; rows cout0.. are contiguous
    const half_t* wb = a.w + ((int64_t)s * a.Cout_pad + cout0) * BK;
#pragma unroll
    for (int j = 0; j < W_ITEMS; ++j) {
      const int row = (tid >> 3) + 32 * j;
      wreg[j] = *reinterpret_cast<const half8*>(wb + row * BK + k8 * 8);
    }
    if constexpr (!SMALLCIN) {
      const int tap = s / csteps, c0 = (s % csteps) * BK;
      const int dy = tap / a.KW - ph, dx = tap % a.KW - pw;
      const int c = c0 + k8 * 8;
      const bool cok = c < a.Cin;
      const bool s0 = c < a.split;
      const half_t* src = s0 ? a.x0 : a.x1;
      const int ctot = s0 ? a.x0_ctot : a.x1_ctot;
      const int coff = s0 ? a.x0_coff + c : a.x1_coff + (c - a.split);
#pragma unroll
      for (int j = 0; j < X_ITEMS; ++j) {
        const int yy = xy[j] + dy, xc = xx[j] + dx;
        const bool ok = cok & xvalid[j] & (yy >= 0) & (yy < a.H) & (xc >= 0) & (xc < a.W);
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) v = *reinterpret_cast<const half8*>(src + ((int64_t)(xe[j] * a.H + yy) * a.W + xc) * ctot + coff);
        xreg[j] = v;
      }
    } else {
      // chunk = global k-chunk index: taps 2q, 2q+1 with 4 channels each
      const int q = s * 8 + k8;
#pragma unroll
      for (int j = 0; j < X_ITEMS; ++j) {
        half4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int tap = 2 * q + h;
          if (tap < a.KH * a.KW) {
            const int yy = xy[j] + tap / a.KW - ph, xc = xx[j] + tap % a.KW - pw;
            if (xvalid[j] & (yy >= 0) & (yy < a.H) & (xc >= 0) & (xc < a.W)) {
              const half4 v = *reinterpret_cast<const half4*>(
                  a.x0 + ((int64_t)(xe[j] * a.H + yy) * a.W + xc) * a.x0_ctot + a.x0_coff);
              if (h == 0) lo = v; else hi = v;
            }
          }
        }
        xreg[j] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
  };
  auto write_lds = [&](int buf) {
#pragma unroll
    for (int j = 0; j < W_ITEMS; ++j) {
      const int row = (tid >> 3) + 32 * j;
      *reinterpret_cast<half8*>(ldsW + buf * BMC * 128 + swz(row, k8)) = wreg[j];
    }
#pragma unroll
    for (int j = 0; j < X_ITEMS; ++j) {
      const int row = (tid >> 3) + 32 * j;
      *reinterpret_cast<half8*>(ldsX + buf * BNP * 128 + swz(row, k8)) = xreg[j];
    }
  };

  float16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  issue_loads(0);
  write_lds(0);
  __syncthreads();
  const int lrow = lane & 31, lhalf = lane >> 5;
  for (int s = 0; s < a.nsteps; ++s) {
    const int cur = s & 1;
    if (s + 1 < a.nsteps) issue_loads(s + 1);
    const unsigned char* bw = ldsW + cur * BMC * 128;
    const unsigned char* bx = ldsX + cur * BNP * 128;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      half8 wf[TM], xf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        wf[i] = *reinterpret_cast<const half8*>(bw + swz((wm * TM + i) * 32 + lrow, kk * 2 + lhalf));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        xf[j] = *reinterpret_cast<const half8*>(bx + swz((wn * TN + j) * 32 + lrow, kk * 2 + lhalf));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < a.nsteps) write_lds(cur ^ 1);
    __syncthreads();
  }

#include "conv_epilogue.inc"
}

// ---- LDS-DMA variant (generic Cin): both operand tiles go global -> LDS with global_load_lds_dwordx4, no staging
// registers and no ds_write.  A wave-instruction fills 1 KiB = 8 tile rows x 128 B; the XOR swizzle of the
// 16-byte chunks is applied on the SOURCE address (the LDS destination of an LDS-DMA is lane-linear).
// Out-of-image taps / channels >= Cin read a 16-byte zero page instead.  Blocks are remapped so that
// consecutive logical tiles (and the cout tiles of one pixel tile) run on the same XCD and share its L2.
__device__ __attribute__((aligned(16))) unsigned char g_zero_page[64];

// 16 bytes per lane, global -> LDS (wave-uniform LDS byte address + lane * 16), as inline asm: hipcc orders every
// later LDS read behind a visible LDS-DMA with s_waitcnt vmcnt(0), which would serialise load and compute; hidden
// in asm the DMA is counted by hand (one s_waitcnt vmcnt(0) before the barrier that publishes the buffer).
// M0 carries the LDS base and is compiler-reserved, so it is saved/restored inside the same statement.
__device__ __forceinline__ void glds16(const void* g, unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned keep;
  const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(base)
      : "memory");
#endif
}

__device__ __forceinline__ unsigned lds_address(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p;
}

template <int BMC, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void conv_mfma_glds_kernel(ConvArgs a, int gy) {
  constexpr int TM = BMC / (WAVES_M * 32);
  constexpr int TN = BNP / (WAVES_N * 32);
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");
  constexpr int WP = BMC / 32;  // 1-KiB weight pieces per wave per K-step
  constexpr int XP = BNP / 32;  // activation pieces per wave per K-step (4)

  extern __shared__ __align__(16) unsigned char lds[];
  unsigned char* ldsW = lds;
  unsigned char* ldsX = lds + 2 * BMC * 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int nb = gridDim.x;
  const int L = xcd_remap(blockIdx.x, nb);
  const int64_t M = (int64_t)a.B * a.H * a.W;
  const int64_t pix0 = (int64_t)(L / gy) * BNP;
  const int cout0 = (L % gy) * BMC;
  const int HW = a.H * a.W;
  const int ph = a.KH / 2, pw = a.KW / 2;
  const int csteps = a.Cin_pad / BK;
  const int r8 = lane >> 3, sl = lane & 7;

  // ---- per-lane descriptors of this wave's pieces
  int woff[WP];
#pragma unroll
  for (int q = 0; q < WP; ++q) {
    const int row = (wave * WP + q) * 8 + r8;
    woff[q] = row * BK + ((sl ^ ((row >> 1) & 7)) << 3);
  }
  const half_t* xb0[XP];
  const half_t* xb1[XP];
  int xk8[XP], py[XP], px[XP];
  unsigned xmask[XP];
#pragma unroll
  for (int q = 0; q < XP; ++q) {
    const int row = (wave * XP + q) * 8 + r8;
    const int64_t m = pix0 + row;
    const bool ok = m < M;
    const int64_t mm = ok ? m : 0;
    const int rem = (int)(mm % HW);
    py[q] = ok ? rem / a.W : -100000;  // invalid pixel: every tap fails the bounds test
    px[q] = rem % a.W;
    const int k8 = sl ^ ((row >> 1) & 7);
    xk8[q] = k8 * 8;
    xb0[q] = a.x0 + mm * a.x0_ctot + a.x0_coff + k8 * 8;
    xb1[q] = a.x1 + mm * a.x1_ctot + a.x1_coff + k8 * 8 - a.split;
    xmask[q] = 0;
  }
  for (int t = 0; t < a.KH * a.KW; ++t) {
    const int dy = t / a.KW - ph, dx = t % a.KW - pw;
#pragma unroll
    for (int q = 0; q < XP; ++q) {
      const int yy = py[q] + dy, xc = px[q] + dx;
      if (yy >= 0 && yy < a.H && xc >= 0 && xc < a.W) xmask[q] |= 1u << t;
    }
  }
  const half_t* zp = reinterpret_cast<const half_t*>(g_zero_page);

  auto issue = [&](int s, int buf) {
    const half_t* wb = a.w + ((int64_t)s * a.Cout_pad + cout0) * BK;
    unsigned char* lw = ldsW + buf * BMC * 128 + wave * WP * 1024;
#pragma unroll
    for (int q = 0; q < WP; ++q)
      glds16(wb + woff[q], lds_address(lw) + q * 1024);
    const int tap = s / csteps, c0 = (s % csteps) * BK;
    const int dy = tap / a.KW - ph, dx = tap % a.KW - pw;
    const bool s0 = c0 < a.split;  // uniform: split is a multiple of 64
    const int64_t delta = (int64_t)(dy * a.W + dx) * (s0 ? a.x0_ctot : a.x1_ctot) + c0;
    const int crem = a.Cin - c0;
    unsigned char* lx = ldsX + buf * BNP * 128 + wave * XP * 1024;
#pragma unroll
    for (int q = 0; q < XP; ++q) {
      const bool ok = ((xmask[q] >> tap) & 1u) && (xk8[q] < crem);
      const half_t* p = (s0 ? xb0[q] : xb1[q]) + delta;
      glds16(ok ? p : zp, lds_address(lx) + q * 1024);
    }
  };

  float16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int lrow = lane & 31, lhalf = lane >> 5;
  for (int s = 0; s < a.nsteps; ++s) {
    const int cur = s & 1;
    const unsigned char* bw = ldsW + cur * BMC * 128;
    const unsigned char* bx = ldsX + cur * BNP * 128;
    // 1. all fragments of this K-step into registers.  hipcc orders any LDS read after an in-flight LDS-DMA with
    //    s_waitcnt vmcnt(0), so the reads must be issued BEFORE the next step's DMA for the two to overlap.
    half8 wf[BK / 16][TM], xf[BK / 16][TN];
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        wf[kk][i] = *reinterpret_cast<const half8*>(bw + swz((wm * TM + i) * 32 + lrow, kk * 2 + lhalf));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        xf[kk][j] = *reinterpret_cast<const half8*>(bx + swz((wn * TN + j) * 32 + lrow, kk * 2 + lhalf));
    }
    __builtin_amdgcn_sched_barrier(0);
    // 2. next K-step's tiles, global -> LDS (other buffer), in flight during the MFMAs
    if (s + 1 < a.nsteps) issue(s + 1, cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    // 3. matrix cores
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk][i], xf[kk][j], acc[i][j], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#include "conv_epilogue.inc"
}

// ---- Halo-tile variant (image width 64, 1x1 / 3x3): the workgroup (8 waves) owns 4 full image rows (256 pixels)
// x 128 output channels.  For every 64-channel chunk the (4+2) x (64+2) input halo is brought to LDS ONCE and the
// 9 taps are read from it as shifted windows, so the L2 -> LDS traffic per FLOP is ~3x lower than with per-tap
// gathers (the measured limiter: ~70 GB/s per CU of LDS-DMA bandwidth).  Weights stream per (tap, chunk) through a
// 2 x 16 KB ring; the next chunk's halo is prefetched one 1-KiB piece per wave per tap step.
constexpr int HALO_TH = 4, HALO_TW = 64;
constexpr int HALO_PITCH = HALO_TW + 2;                 // 66
constexpr int HALO_ROWS = (HALO_TH + 2) * HALO_PITCH;  // 396
constexpr int HALO_PIECES = (HALO_ROWS + 7) / 8;       // 50
constexpr int HALO_LDS_ROWS = HALO_PIECES * 8;         // 400
constexpr int HALO_XP = (HALO_PIECES + 7) / 8;         // pieces per wave (7)

__device__ __forceinline__ void glds16_off(const void* base, unsigned voff_bytes, unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned keep;
  const unsigned lb = __builtin_amdgcn_readfirstlane(lds_addr);
  const uint64_t bu = (uint64_t)base;  // wave-uniform by construction; make that explicit for the "s" constraint
  const uint64_t bs = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(bu >> 32)) << 32) |
                      (unsigned)__builtin_amdgcn_readfirstlane((int)bu);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff_bytes), "s"(bs), "s"(lb)
      : "memory");
#endif
}

// Branch-free forms for the K loop of the halo kernels: the source base (or, per lane, the whole pointer), the lane offset
// and the LDS piece are SELECTED (s_cselect / v_cndmask) between the real transfer and a dummy read of the zero page into
// the sink, so that a tap's DMA issue is a dozen straight-line scalar instructions the scheduler can place between its
// matrix instructions - as `if (...) issue else dummy` it was four basic blocks and ~45 scalar instructions per tap that
// every wave executed between the barrier and its first fragment read (round-3 ISA reading).
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  return ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ void dma16s(uint64_t sbase, unsigned voff_bytes, unsigned lds_addr) {  // sbase, lds_addr wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff_bytes), "s"(sbase), "s"(lds_addr)
               : "memory");
#endif
}
__device__ __forceinline__ void dma16v(uint64_t lane_ptr, unsigned lds_addr) {  // per-lane source, lds_addr wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_ptr), "s"(lds_addr)
               : "memory");
#endif
}

// ---- Halo-tile kernel, K step 32: TWO workgroups per CU.
// Same tile (8 waves: 4 image rows x 64 px x BMC couts) and the same halo idea as above, but the channel chunk is 32
// (64-byte LDS rows): halo double buffer 2 x 25 KiB + weight ring 3 x 8 KiB = 75 KiB, so two workgroups share a CU
// (4 waves per SIMD) and one workgroup's prologue / epilogue / barrier stalls are covered by the other's MFMAs
// (measured on the 64-channel kernel: 12 us of 27..63 us per workgroup were spent outside the K loop with nothing
// to overlap them).  The epilogue applies bias + activation in the accumulator layout and stages the tile as fp16
// (69 KiB), then every lane blends / stores 8 consecutive channels of one pixel.
// LDS rows are 4 chunks of 16 B; logical chunk c of row r sits at slot c ^ ((r >> 2) & 3): the 16-lane groups of a
// ds_read_b128 (rows r0 + {0..3, 12..15, 20..27}, one logical chunk) then cover 16 distinct 16-byte slots of the
// 256-byte bank window for every window shift r0.
constexpr int H32_PIECES = (HALO_ROWS + 15) / 16;  // 25 one-KiB pieces (16 rows x 64 B) per halo chunk
constexpr int H32_XBYTES = H32_PIECES * 1024;      // 25600
constexpr int H32_XP = (H32_PIECES + 7) / 8;       // pieces per wave (4)
constexpr int H32_BK = 32;

// chunk swizzle of a 64-byte row.  32x32x16 fragments (16-lane read groups = rows r0 + {0..3, 12..15, 20..27}, one
// logical chunk): c ^ ((r >> 2) & 3).  16x16x32 fragments (read groups = rows r0 + {0..3, 12..15} with chunk kc and
// rows r0 + {4..11} with chunk kc ^ 1): c ^ 2 * ((r >> 2) & 1) - both cover 16 distinct 16-byte slots for every r0.
template <bool M16>
__device__ __forceinline__ int swzf(int row) { return M16 ? ((row >> 2) & 1) << 1 : (row >> 2) & 3; }
template <bool M16>
__device__ __forceinline__ int swz32(int row, int c) { return row * 64 + ((c ^ swzf<M16>(row)) << 4); }


// position q of a flat tile -> pixel index of image e ((e * H + y) * W + x), or -1 for the pad column / beyond the image
// (q / wp by the multiply-high of a 32-bit reciprocal, exact for q < 2^32 / wp: three integer instructions and no temporaries
// the K loop would have to keep live - this runs at every halo-piece issue)
__device__ __forceinline__ int flat_pixel(int q, int e, int H, int W, int wp, int hwp, unsigned magic) {
  if ((unsigned)q >= (unsigned)hwp) return -1;  // also q < 0
  const int y = (int)__umulhi((unsigned)q, magic);
  const int x = q - y * wp;
  return x < W ? (e * H + y) * W + x : -1;
}

// VAR: 0 = the general kernel; 1 = initial accumulators held as fp32 (a.ai_f32); 2 = EPI_PARTIAL (raw fp32 accumulators
// out).  Separate instantiations: folded into the general one as run-time branches they cost it 75 spilled VGPRs.
// FLAT: any H x W.  The image is taken with one zero column appended to every row (pitch Wp = W + 1; a 1x1 needs none)
// and flattened: position q = y * Wp + x.  A tile is 256 CONSECUTIVE positions of one image, so tap (dy, dx) of every
// position is the position dy * Wp + dx further on - one shifted window of a single run of 256 + 2 Wp + 2 halo rows in
// LDS, whatever the image shape (the pad column is the left AND the right zero border).  The reference's resize gives
// 41 x 73 grids for 16:9 video (vipe/slam/system.py:46-59): 12 tiles of 256 cover the 2993 pixels at 97 %, where 4 x 64
// tiles would run at 53 %.  Wave wn owns positions 64 wn .. 64 wn + 63; pad positions are computed and not stored.
template <int BMC, int KS, bool M16, int VAR = 0, bool FLAT = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_halo32_kernel(ConvArgs a, int gy) {
  // M16: v_mfma_f32_16x16x32_f16 (one instruction per 32-channel step and 16 x 16 tile; the chip holds a higher
  // clock on this shape, MI355X_MICROARCH.md 'DVFS give-back' (7)); otherwise 32x32x16.
  constexpr int TM = BMC >= 128 ? 2 : 1, TN = BMC == 32 ? 1 : 2, BP = HALO_TH * HALO_TW;
  constexpr int MI = TM * 2, NJ = TN * 2;  // 16 x 16 tiles per wave (M16)
  constexpr int WSTAGE = BMC * 64;  // bytes per weight ring slot
  extern __shared__ __align__(16) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = BMC == 32 ? 0 : (wave >> 2), wn = wave & 3;  // cout half, image row of the tile
  const int pxh = BMC == 32 ? (wave >> 2) * 32 : 0;           // BMC = 32: which half of the row
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = L / gy, cout0 = (L % gy) * BMC;
  const int xsegs = a.W / HALO_TW, tiles_per_img = FLAT ? a.f_tiles : (a.H / HALO_TH) * xsegs;
  const int e = tile / tiles_per_img, trem = tile % tiles_per_img;
  const int y0 = (trem / xsegs) * HALO_TH, x0 = (trem % xsegs) * HALO_TW;
  const int q0 = trem * BP;                                   // FLAT: first position of the tile
  const int pitch = FLAT ? a.f_wp : HALO_PITCH;               // halo rows per image row
  const int xbytes = FLAT ? a.f_xbytes : H32_XBYTES;          // bytes of one halo buffer
  const int npieces = FLAT ? a.f_pieces : H32_PIECES;
  // pixel index of tile pixel pl (row pl / 64, column pl % 64; FLAT: position q0 + pl, -1 when it is no pixel)
  auto pix_of = [&](int pl) -> int64_t {
    if constexpr (FLAT) return flat_pixel(q0 + pl, e, a.H, a.W, a.f_wp, a.f_hwp, a.f_magic);
    else return ((int64_t)(e * a.H + y0 + (pl >> 6))) * a.W + x0 + (pl & 63);
  };
  const int cs32 = a.Cin_pad / H32_BK, cs64 = a.Cin_pad / 64;
  const int r16 = lane >> 2, sl = lane & 3;
  const int lrow = lane & 31, lhalf = lane >> 5;

  const unsigned ldsW_a = lds_address(lds), ldsX_a = ldsW_a + 3 * WSTAGE;
  const unsigned sink_a = ldsX_a + 2 * xbytes;  // 1 KiB sink for the count-keeping dummy DMAs
  const unsigned char* ldsW = lds;
  const unsigned char* ldsX = lds + 3 * WSTAGE;
  const half_t* zp = reinterpret_cast<const half_t*>(g_zero_page);

  // ---- DMA descriptors
  constexpr int WPIECES = BMC / 16;  // 1-KiB pieces per weight slot: 8 / 4 / 2 -> waves 0..WPIECES-1 carry one each
  const bool wreal = wave < WPIECES;
  const int wrow = (wave % WPIECES) * 16 + r16;
  const unsigned woff = (unsigned)(wrow * 64 + ((sl ^ swzf<M16>(wrow)) << 3)) * 2u;
  const unsigned wdst = ldsW_a + (wave % WPIECES) * 1024;
  // halo pieces: one VGPR each (pixel index of this lane's halo row, or -1 outside the image); the lane's channel
  // chunk is the same for every piece because (16 * piece) >> 2 is a multiple of 4
  // FLAT: halo row r holds position q0 - (Wp + 1) + r; its pixel index is recomputed at every issue (4 per wave and
  // chunk, ~15 VALU each) instead of being kept in registers the K loop has none to spare of
  int xpix[FLAT ? 1 : H32_XP];
  const int xk = (sl ^ swzf<M16>(r16)) * 8;
  if constexpr (!FLAT) {
#pragma unroll
    for (int i = 0; i < H32_XP; ++i) {
      const int pce = wave + 8 * i;
      const int r = pce * 16 + r16;
      const int hy = r / HALO_PITCH, hx = r % HALO_PITCH;
      const int y = y0 + hy - 1, x = x0 + hx - 1;
      const bool ok = pce < H32_PIECES && r < HALO_ROWS && y >= 0 && y < a.H && x >= 0 && x < a.W;
      xpix[i] = ok ? (e * a.H + y) * a.W + x : -1;
    }
  }
  // a 1x1 reads only the tile's own rows: pieces 4..20 hold halo rows 64..335 (FLAT: rows Wp + 1 .. Wp + 256)
  auto piece_used = [&](int i) {
    const int pce = wave + 8 * i;
    if constexpr (FLAT) return pce < npieces && (KS == 3 || (pce >= a.f_p1lo && pce <= a.f_p1hi));
    else return pce < H32_PIECES && (KS == 3 || (pce >= 4 && pce <= 20));
  };
  auto issueX = [&](int c, int i, int buf) {
    const int c0 = c * H32_BK;
    const bool s0 = c0 < a.split;  // wave-uniform
    const int ctot = s0 ? a.x0_ctot : a.x1_ctot;
    const int cb = (s0 ? a.x0_coff : a.x1_coff - a.split) + c0;
    int xp;
    if constexpr (FLAT) xp = flat_pixel(q0 - pitch - 1 + (wave + 8 * i) * 16 + r16, e, a.H, a.W, a.f_wp, a.f_hwp, a.f_magic);
    else xp = xpix[i];
    const bool ok = xp >= 0 && (c0 + xk < a.Cin);
    const uint64_t off = ((uint64_t)(unsigned)xp * (unsigned)ctot + (unsigned)(cb + xk)) * 2u;  // one v_mad_u64_u32: no 4 GiB limit
    const char* src = reinterpret_cast<const char*>(s0 ? a.x0 : a.x1) + off;
    glds16(ok ? (const void*)src : (const void*)zp, ldsX_a + buf * xbytes + (wave + 8 * i) * 1024);
  };
  // weights of K-step (tap, c): the 64-byte half (c & 1) of rows cout0.. of packed block tap * cs64 + c / 2
  auto wsrc = [&](int tap, int c) {
    return a.w + ((int64_t)(tap * cs64 + (c >> 1)) * a.Cout_pad + cout0) * BK + (c & 1) * H32_BK;
  };
  auto issueW = [&](const half_t* wb, int slot) {
    if (wreal) glds16_off(wb, woff, wdst + slot * WSTAGE);
    else glds16(zp, sink_a);
  };
  // ---- the same two transfers, branch-free (K loop): `on` is wave-uniform
  const uint64_t zp64 = uniform64((uint64_t)zp);
  const uint64_t wbase = uniform64((uint64_t)(a.w + (int64_t)cout0 * BK));  // rows cout0.. of packed block 0
  const unsigned tstride = (unsigned)cs64 * (unsigned)a.Cout_pad * BK * 2u;    // bytes from one tap's packed blocks to the next's
  const unsigned sink_u = __builtin_amdgcn_readfirstlane(sink_a), wdst_u = __builtin_amdgcn_readfirstlane(wdst);
  const unsigned ldsX_u = __builtin_amdgcn_readfirstlane(ldsX_a + wave * 1024);
  auto issueW2 = [&](int tap, int c, int slot, auto ONC, bool more_) {  // weights of K-step (tap, c) into ring slot `slot`
    const unsigned off = (unsigned)tap * tstride + ((unsigned)(c >> 1) * (unsigned)a.Cout_pad * BK + (unsigned)(c & 1) * H32_BK) * 2u;
    if constexpr (false) {
      dma16s(wbase + off, woff, wdst_u + slot * WSTAGE);  // every wave carries a piece and the step exists: nothing to select
    } else {
      const bool on = (decltype(ONC)::value || more_) && wreal;
      dma16s(on ? wbase + off : zp64, on ? woff : 0u, on ? wdst_u + slot * WSTAGE : sink_u);
    }
  };
  auto issueX2 = [&](int c, int i, int buf, bool on) {   // halo piece i of chunk c into halo buffer `buf`
    const int c0 = c * H32_BK;
    const bool s0 = c0 < a.split;  // wave-uniform
    const int ctot = s0 ? a.x0_ctot : a.x1_ctot;
    const int cb = (s0 ? a.x0_coff : a.x1_coff - a.split) + c0;
    int xp;
    if constexpr (FLAT) xp = flat_pixel(q0 - pitch - 1 + (wave + 8 * i) * 16 + r16, e, a.H, a.W, a.f_wp, a.f_hwp, a.f_magic);
    else xp = xpix[i];
    const bool ok = on && xp >= 0 && (c0 + xk < a.Cin);
    const uint64_t off = ((uint64_t)(unsigned)xp * (unsigned)ctot + (unsigned)(cb + xk)) * 2u;  // one v_mad_u64_u32: no 4 GiB limit
    const uint64_t base = (uint64_t)(s0 ? a.x0 : a.x1);
    dma16v(ok ? base + off : zp64, on ? ldsX_u + buf * xbytes + i * 8192 : sink_u);
  };

  float16v acc[M16 ? 1 : TM][M16 ? 1 : TN];
  float4v acc16[M16 ? MI : 1][M16 ? NJ : 1];
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc16[i][j] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
    if constexpr (BMC >= 64) {
      // the accumulators may start from a precomputed partial sum (convolution is linear in its input channels:
      // the part of a GRU gate that only depends on the per-edge context features is computed once per edge)
      if (a.accinit) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int64_t m = pix_of(wn * 64 + j * 16 + (lane & 15));
          if (FLAT && m < 0) continue;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int cg = cout0 + wm * (MI * 16) + i * 16 + 4 * (lane >> 4);
            if constexpr (VAR == 1) {
              const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.accinit) + m * a.ai_ctot +
                                                                a.ai_coff + cg);
              acc16[i][j] = float4v{v.x, v.y, v.z, v.w};
            } else {
              const half4 v = *reinterpret_cast<const half4*>(a.accinit + m * a.ai_ctot + a.ai_coff + cg);
              acc16[i][j] = float4v{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            }
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  }

  // fragment read addresses: weights are fixed per lane; the activation window of tap (dy, dx) starts at halo row
  // (wn + dy + 1) * 66 + pxh + dx + 1.
  const int l16 = lane & 15, lk = lane >> 4;
  const int wa0 = M16 ? swz32<true>(wm * MI * 16 + l16, lk) : swz32<false>(wm * TM * 32 + lrow, lhalf);
  // Rl = this lane's halo row of the (-1, -1) window; the caller passes it through an empty asm once per chunk so
  // that the nine per-tap addresses are recomputed (5 VALU each) instead of being hoisted into 9+ VGPRs (spills)
  const int Rl0 = (FLAT ? wn * HALO_TW : wn * HALO_PITCH) + pxh + (M16 ? l16 : lrow);
  auto mma_step = [&](const unsigned char* bw, const unsigned char* bx, int Rl, int dy, int dx) {
    const int rb = Rl + (dy + 1) * pitch + (dx + 1);
    if constexpr (M16) {
      const int xa0 = swz32<true>(rb, lk);
      half8 wf[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) wf[i] = *reinterpret_cast<const half8*>(bw + (wa0 + i * 1024));
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const half8 xf = *reinterpret_cast<const half8*>(bx + (xa0 + j * 1024));
#pragma unroll
        for (int i = 0; i < MI; ++i)
          acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf, acc16[i][j], 0, 0, 0);
      }
    } else {
      // kk = 1 flips chunk bit 1 (byte offset ^ 32)
      const int xa0 = swz32<false>(rb, lhalf);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        half8 wf[TM], xf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) wf[i] = *reinterpret_cast<const half8*>(bw + ((wa0 ^ (kk << 5)) + i * 2048));
#pragma unroll
        for (int j = 0; j < TN; ++j) xf[j] = *reinterpret_cast<const half8*>(bx + ((xa0 ^ (kk << 5)) + j * 2048));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  // the same step in two halves (16x16x32 form): all eight fragment reads of a tap, then its sixteen matrix instructions -
  // the 3x3 K loop puts the tap's LDS-DMA issue between them, in the shadow of the reads' latency
  auto load_frags = [&](const unsigned char* bw, const unsigned char* bx, int Rl, int dy, int dx, half8 (&wf)[MI], half8 (&xf)[NJ]) {
    const int xa0 = swz32<true>(Rl + (dy + 1) * pitch + (dx + 1), lk);
#pragma unroll
    for (int i = 0; i < MI; ++i) wf[i] = *reinterpret_cast<const half8*>(bw + (wa0 + i * 1024));
#pragma unroll
    for (int j = 0; j < NJ; ++j) xf[j] = *reinterpret_cast<const half8*>(bx + (xa0 + j * 1024));
  };
  auto mma_frags = [&](const half8 (&wf)[MI], const half8 (&xf)[NJ]) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc16[i][j], 0, 0, 0);
  };

  // ---- prologue: halo chunk 0, weights of steps 0 and 1
#pragma unroll
  for (int i = 0; i < H32_XP; ++i)
    if (piece_used(i)) issueX(0, i, 0);
  issueW(wsrc(0, 0), 0);
  if constexpr (KS == 3) issueW(wsrc(1, 0), 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  if constexpr (KS == 3) {
    // 9 taps fully unrolled: ring slot = tap % 3, tap offsets and the per-tap s_waitcnt count are compile-time.
    // Every tap issues one weight DMA (step + 2); taps 0..3 also one halo piece of the next chunk (dummies keep the
    // count when there is nothing to fetch).  At the end of tap t the DMAs younger than W(step + 1) are
    // {X_t (t < 4), W(step + 2), X_{t-1} (1 <= t <= 4)}: vmcnt(2,3,3,3,2,1,1,1,1).  A halo piece issued at tap t <= 3 has
    // landed by the end of tap 5, before the chunk ends.
    for (int c = 0; c < cs32; ++c) {
      const unsigned char* bx = ldsX + (c & 1) * xbytes;
      const bool more = c + 1 < cs32;
      int Rl = Rl0;
      asm volatile("" : "+v"(Rl));
      auto tap_step = [&](auto TAPC) {
        constexpr int TAP = decltype(TAPC)::value;
        constexpr int T2 = TAP + 2;
        if constexpr (M16) {
          half8 wf[MI], xf[NJ];
          load_frags(ldsW + (TAP % 3) * WSTAGE, bx, Rl, TAP / 3 - 1, TAP % 3 - 1, wf, xf);
          if constexpr (T2 < 9) issueW2(T2, c, T2 % 3, std::true_type{}, true);
          else issueW2(T2 - 9, c + 1, T2 % 3, std::false_type{}, more);
          if constexpr (TAP < H32_XP) issueX2(c + 1, TAP, (c + 1) & 1, more && piece_used(TAP));
          mma_frags(wf, xf);
        } else {
          if constexpr (T2 < 9) issueW2(T2, c, T2 % 3, std::true_type{}, true);
          else issueW2(T2 - 9, c + 1, T2 % 3, std::false_type{}, more);
          if constexpr (TAP < H32_XP) issueX2(c + 1, TAP, (c + 1) & 1, more && piece_used(TAP));
          mma_step(ldsW + (TAP % 3) * WSTAGE, bx, Rl, TAP / 3 - 1, TAP % 3 - 1);
        }
        if constexpr (TAP == 0 || TAP == 4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if constexpr (TAP < 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __syncthreads();
      };
      tap_step(std::integral_constant<int, 0>{});
      tap_step(std::integral_constant<int, 1>{});
      tap_step(std::integral_constant<int, 2>{});
      tap_step(std::integral_constant<int, 3>{});
      tap_step(std::integral_constant<int, 4>{});
      tap_step(std::integral_constant<int, 5>{});
      tap_step(std::integral_constant<int, 6>{});
      tap_step(std::integral_constant<int, 7>{});
      tap_step(std::integral_constant<int, 8>{});
    }
  } else {
    for (int c = 0; c < cs32; ++c) {
      const unsigned char* bx = ldsX + (c & 1) * xbytes;
      if (c + 1 < cs32) {
        issueW(wsrc(0, c + 1), (c + 1) & 1);
#pragma unroll
        for (int i = 0; i < H32_XP; ++i)
          if (piece_used(i)) issueX(c + 1, i, (c + 1) & 1);
      }
      mma_step(ldsW + (c & 1) * WSTAGE, bx, Rl0, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue
  constexpr int CPP = BMC / 8;  // 8-channel chunks per pixel
  constexpr int NIT = BP * CPP / 512, PSTEP = 512 / CPP;
  const int ch = (tid % CPP) * 8, co = cout0 + ch, pl0 = tid / CPP;
  if constexpr (BMC >= 64) {
    if constexpr (M16 && VAR == 2) {
      {
        // raw fp32 accumulators (no bias, no activation) for a later launch to start from: lanes lk = 0..3 of a pixel
        // cover 16 consecutive channels = one 64-byte sector per store instruction and pixel
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int64_t m = pix_of(wn * 64 + j * 16 + l16);
          if (FLAT && m < 0) continue;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int cg = cout0 + wm * (MI * 16) + i * 16 + 4 * lk;
            if (cg + 4 <= a.Cout)
              *reinterpret_cast<float4*>(a.fout + m * a.y_ctot + a.y_coff + cg) =
                  make_float4(acc16[i][j][0], acc16[i][j][1], acc16[i][j][2], acc16[i][j][3]);
          }
        }
        return;
      }
    }
    constexpr int PITCH = BMC + 8;  // halves per staged pixel
    half_t* stage = reinterpret_cast<half_t*>(lds);
    auto stage_all = [&](auto ACTC) {
      constexpr int ACT = decltype(ACTC)::value;
      auto put = [&](int cl, int pl, float4 b, float v0, float v1, float v2, float v3) {
        half4 h;
        h[0] = (half_t)act_apply(v0 + b.x, ACT);
        h[1] = (half_t)act_apply(v1 + b.y, ACT);
        h[2] = (half_t)act_apply(v2 + b.z, ACT);
        h[3] = (half_t)act_apply(v3 + b.w, ACT);
        *reinterpret_cast<half4*>(stage + pl * PITCH + cl) = h;
      };
      auto bias4 = [&](int cl) {
        const int cg = cout0 + cl;
        float4 b = *reinterpret_cast<const float4*>(a.bias + cg);
        if (a.extra && cg + 4 <= a.Cout) {
          const float4 x = *reinterpret_cast<const float4*>(a.extra + (int64_t)e * a.extra_stride + a.extra_off + cg);
          b.x += x.x; b.y += x.y; b.z += x.z; b.w += x.w;
        }
        return b;
      };
      if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int cl = wm * (MI * 16) + i * 16 + 4 * lk;
          const float4 b = bias4(cl);
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            put(cl, wn * 64 + j * 16 + l16, b, acc16[i][j][0], acc16[i][j][1], acc16[i][j][2], acc16[i][j][3]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int cl = wm * (TM * 32) + i * 32 + 8 * g + 4 * lhalf;
            const float4 b = bias4(cl);
#pragma unroll
            for (int j = 0; j < TN; ++j)
              put(cl, wn * 64 + j * 32 + lrow, b, acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2],
                  acc[i][j][4 * g + 3]);
          }
      }
    };
    const int actsel = a.epi == EPI_PLAIN ? a.act : (a.epi == EPI_Q ? VIPE_ACT_TANH : VIPE_ACT_SIGMOID);
    if (actsel == VIPE_ACT_RELU) stage_all(std::integral_constant<int, VIPE_ACT_RELU>{});
    else if (actsel == VIPE_ACT_SIGMOID) stage_all(std::integral_constant<int, VIPE_ACT_SIGMOID>{});
    else if (actsel == VIPE_ACT_TANH) stage_all(std::integral_constant<int, VIPE_ACT_TANH>{});
    else stage_all(std::integral_constant<int, VIPE_ACT_NONE>{});
    // global operands of the blend (hidden state, update gate): issued before the barrier
    const bool want_net = a.epi == EPI_GLO || a.epi == EPI_Q || (a.epi == EPI_ZR && co >= 128);
    const bool want_z = a.epi == EPI_Q;
    half8 nvv[NIT], zvv[NIT];
    {
      const int cn = a.epi == EPI_ZR ? co - 128 : co;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int64_t m = pix_of(pl0 + PSTEP * it);
        if (FLAT && m < 0) continue;  // no pixel here: the store loop skips this position as well
        if (want_net) nvv[it] = *reinterpret_cast<const half8*>(a.net + m * a.net_ctot + a.net_coff + cn);
        if (want_z) zvv[it] = *reinterpret_cast<const half8*>(a.zbuf + m * 128 + co);
      }
    }
    __syncthreads();
    float gsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pl = pl0 + PSTEP * it;
      const int64_t m = pix_of(pl);
      if (FLAT && m < 0) continue;
      const half8 sv = *reinterpret_cast<const half8*>(stage + pl * PITCH + ch);
      half8 o = sv;
      half_t* dst = nullptr;
      if (a.epi == EPI_PLAIN) {
        dst = a.y + m * a.y_ctot + a.y_coff + co;
      } else if (a.epi == EPI_GLO) {
#pragma unroll
        for (int q = 0; q < 8; ++q) gsum[q] += (float)sv[q] * (float)nvv[it][q];
      } else if (a.epi == EPI_ZR) {
        if (co < 128) {
          dst = a.y + m * a.y_ctot + a.y_coff + co;
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = (half_t)((float)sv[q] * (float)nvv[it][q]);
          dst = a.y2 + m * a.y2_ctot + a.y2_coff + co - 128;
        }
      } else if (a.epi == EPI_Q) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float z = (float)zvv[it][q];
          o[q] = (half_t)((1.0f - z) * (float)nvv[it][q] + z * (float)sv[q]);  // droid_net.py:399
        }
        dst = a.y + m * a.y_ctot + a.y_coff + co;
      }
      if (dst) {
        if (co + 8 <= a.Cout) {
          *reinterpret_cast<half8*>(dst) = o;
        } else {
          for (int q = 0; q < 8 && co + q < a.Cout; ++q) dst[q] = o[q];
        }
      }
    }
    if (a.epi == EPI_GLO) {
      // threads with equal tid % CPP hold the same 8 channels (different pixels): fold through LDS, then ONE
      // atomic per channel and workgroup
      float* red = reinterpret_cast<float*>(lds);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 8; ++q) red[(tid / CPP) * (BMC + 1) + ch + q] = gsum[q];
      __syncthreads();
      if (tid < BMC) {
        float t = 0.0f;
        for (int r = 0; r < 512 / CPP; ++r) t += red[r * (BMC + 1) + tid];
        if (cout0 + tid < a.Cout) atomicAdd(a.fout + (int64_t)e * a.Cout + cout0 + tid, t);
      }
    }
  } else {
    // BMC = 32 (flow / confidence heads, eta, narrow plain convs): fp32 staging, activation after it
    constexpr int PITCH = BMC + 4;
    float* stage = reinterpret_cast<float*>(lds);
    if constexpr (M16) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          *reinterpret_cast<float4*>(stage + (wn * 64 + pxh + j * 16 + l16) * PITCH + i * 16 + 4 * lk) =
              make_float4(acc16[i][j][0], acc16[i][j][1], acc16[i][j][2], acc16[i][j][3]);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int pl = wn * 64 + pxh + lrow, cs = 8 * g + 4 * lhalf;
        *reinterpret_cast<float4*>(stage + pl * PITCH + cs) =
            make_float4(acc[0][0][4 * g], acc[0][0][4 * g + 1], acc[0][0][4 * g + 2], acc[0][0][4 * g + 3]);
      }
    }
    __syncthreads();
    float bv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      bv[q] = a.bias[co + q];
      if (a.extra && co + q < a.Cout) bv[q] += a.extra[(int64_t)e * a.extra_stride + a.extra_off + co + q];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pl = pl0 + PSTEP * it;
      const int64_t m = pix_of(pl);
      if (FLAT && m < 0) continue;
      const float4 v0 = *reinterpret_cast<const float4*>(stage + pl * PITCH + ch);
      const float4 v1 = *reinterpret_cast<const float4*>(stage + pl * PITCH + ch + 4);
      const float v[8] = {v0.x + bv[0], v0.y + bv[1], v0.z + bv[2], v0.w + bv[3],
                          v1.x + bv[4], v1.y + bv[5], v1.z + bv[6], v1.w + bv[7]};
      if (a.epi == EPI_HEADS) {
        // cout 0,1: delta; cout 2,3: sigmoid -> weight (droid_net.py:486-490); written as float [M,4]
        if (ch == 0)
          *reinterpret_cast<float4*>(a.fout + m * 4) =
              make_float4((float)(half_t)v[0], (float)(half_t)v[1], (float)(half_t)act_apply(v[2], VIPE_ACT_SIGMOID),
                          (float)(half_t)act_apply(v[3], VIPE_ACT_SIGMOID));
      } else if (a.epi == EPI_ETA) {
        if (ch == 0) {  // 0.01 * softplus (droid_net.py:410,429)
          const float sp = v[0] > 20.0f ? v[0] : log1pf(__expf(v[0]));
          a.fout[m] = 0.01f * (float)(half_t)sp;
        }
      } else {
        half_t* dst = a.y + m * a.y_ctot + a.y_coff + co;
        half8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (half_t)act_apply(v[q], a.act);
        if (co + 8 <= a.Cout) {
          *reinterpret_cast<half8*>(dst) = o;
        } else {
          for (int q = 0; q < 8 && co + q < a.Cout; ++q) dst[q] = o[q];
        }
      }
    }
  }
}

// ---- 3x3 with at most 16 output channels (flow / confidence heads: 4, eta: 1).  One 16x16x32 A fragment covers all
// couts, so a K step of one tap is only 2 MFMAs per wave and the per-tap barrier of the general kernel dominates
// (measured 221 us for 15.6 GFLOP).  Here a step is a whole 32-channel chunk: the 9 taps' weights (9 x 1 KiB) and
// the halo chunk are double buffered, one barrier per chunk, 18 MFMAs per wave between barriers.
constexpr int NRW_WBYTES = 9 * 1024;

template <bool FLAT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_halo32_narrow_kernel(ConvArgs a) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, pxh = (wave >> 2) * 32;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int xsegs = a.W / HALO_TW, tiles_per_img = FLAT ? a.f_tiles : (a.H / HALO_TH) * xsegs;
  const int e = tile / tiles_per_img, trem = tile % tiles_per_img;
  const int y0 = (trem / xsegs) * HALO_TH, x0 = (trem % xsegs) * HALO_TW;
  const int q0 = trem * (HALO_TH * HALO_TW);  // FLAT tiling (see conv_halo32_kernel): first position of the tile
  const int pitch = FLAT ? a.f_wp : HALO_PITCH, xbytes = FLAT ? a.f_xbytes : H32_XBYTES;
  const int npieces = FLAT ? a.f_pieces : H32_PIECES;
  auto pix_of = [&](int pl) -> int64_t {
    if constexpr (FLAT) return flat_pixel(q0 + pl, e, a.H, a.W, a.f_wp, a.f_hwp, a.f_magic);
    else return ((int64_t)(e * a.H + y0 + (pl >> 6))) * a.W + x0 + (pl & 63);
  };
  const int cs32 = a.Cin_pad / H32_BK, cs64 = a.Cin_pad / 64;
  const int r16 = lane >> 2, sl = lane & 3, l16 = lane & 15, lk = lane >> 4;
  const unsigned ldsW_a = lds_address(lds), ldsX_a = ldsW_a + 2 * NRW_WBYTES;
  const unsigned char* ldsW = lds;
  const unsigned char* ldsX = lds + 2 * NRW_WBYTES;
  const half_t* zp = reinterpret_cast<const half_t*>(g_zero_page);

  int xpix[H32_XP];
  const int xk = (sl ^ swzf<true>(r16)) * 8;
#pragma unroll
  for (int i = 0; i < H32_XP; ++i) {
    const int pce = wave + 8 * i;
    const int r = pce * 16 + r16;
    if constexpr (FLAT) {
      xpix[i] = pce < npieces ? flat_pixel(q0 - pitch - 1 + r, e, a.H, a.W, a.f_wp, a.f_hwp, a.f_magic) : -1;
    } else {
      const int hy = r / HALO_PITCH, hx = r % HALO_PITCH;
      const int y = y0 + hy - 1, x = x0 + hx - 1;
      const bool ok = pce < H32_PIECES && r < HALO_ROWS && y >= 0 && y < a.H && x >= 0 && x < a.W;
      xpix[i] = ok ? (e * a.H + y) * a.W + x : -1;
    }
  }
  auto issue = [&](int c, int buf) {
    const int c0 = c * H32_BK;
    const bool s0 = c0 < a.split;
    const int ctot = s0 ? a.x0_ctot : a.x1_ctot;
    const int cb = (s0 ? a.x0_coff : a.x1_coff - a.split) + c0;
#pragma unroll
    for (int i = 0; i < H32_XP; ++i) {
      if (wave + 8 * i < npieces) {
        const bool ok = xpix[i] >= 0 && (c0 + xk < a.Cin);
        const uint64_t off = ((uint64_t)(unsigned)xpix[i] * (unsigned)ctot + (unsigned)(cb + xk)) * 2u;
        const char* src = reinterpret_cast<const char*>(s0 ? a.x0 : a.x1) + off;
        glds16(ok ? (const void*)src : (const void*)zp, ldsX_a + buf * xbytes + (wave + 8 * i) * 1024);
      }
    }
    // weights: tap `wave` (and tap 8 on wave 0): rows 0..15 of packed block tap * cs64 + c / 2, 64-byte half c & 1
    const unsigned woff = (unsigned)(r16 * 64 + ((sl ^ swzf<true>(r16)) << 3)) * 2u;
    const half_t* wb = a.w + ((int64_t)(wave * cs64 + (c >> 1)) * a.Cout_pad) * BK + (c & 1) * H32_BK;
    glds16_off(wb, woff, ldsW_a + buf * NRW_WBYTES + wave * 1024);
    if (wave == 0) {
      const half_t* w8 = a.w + ((int64_t)(8 * cs64 + (c >> 1)) * a.Cout_pad) * BK + (c & 1) * H32_BK;
      glds16_off(w8, woff, ldsW_a + buf * NRW_WBYTES + 8 * 1024);
    }
  };

  float4v acc[2] = {float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}};
  const int wa0 = swz32<true>(l16, lk);
  const int Rl0 = (FLAT ? wn * HALO_TW : wn * HALO_PITCH) + pxh + l16;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int c = 0; c < cs32; ++c) {
    if (c + 1 < cs32) issue(c + 1, (c + 1) & 1);
    const unsigned char* bw = ldsW + (c & 1) * NRW_WBYTES;
    const unsigned char* bx = ldsX + (c & 1) * xbytes;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int rb = Rl0 + (tap / 3) * pitch + (tap % 3);
      const int xa0 = swz32<true>(rb, lk);
      const half8 wf = *reinterpret_cast<const half8*>(bw + (wa0 + tap * 1024));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const half8 xf = *reinterpret_cast<const half8*>(bx + (xa0 + j * 1024));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf, acc[j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // ---- epilogue: fp32 tile [256 px][16 couts] through LDS, one pixel per thread
  constexpr int PITCH = 20;
  float* stage = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int j = 0; j < 2; ++j)
    *reinterpret_cast<float4*>(stage + (wn * 64 + pxh + j * 16 + l16) * PITCH + 4 * lk) =
        make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  __syncthreads();
  const int64_t m = tid < HALO_TH * HALO_TW ? pix_of(tid) : -1;
  if (tid < HALO_TH * HALO_TW && (!FLAT || m >= 0)) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t4 = *reinterpret_cast<const float4*>(stage + tid * PITCH + 4 * q);
      v[4 * q] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      v[q] += a.bias[q];
      if (a.extra && q < a.Cout) v[q] += a.extra[(int64_t)e * a.extra_stride + a.extra_off + q];
    }
    if (a.epi == EPI_HEADS) {
      // cout 0,1: delta; cout 2,3: sigmoid -> weight (droid_net.py:486-490); written as float [M,4]
      *reinterpret_cast<float4*>(a.fout + m * 4) =
          make_float4((float)(half_t)v[0], (float)(half_t)v[1], (float)(half_t)act_apply(v[2], VIPE_ACT_SIGMOID),
                      (float)(half_t)act_apply(v[3], VIPE_ACT_SIGMOID));
    } else if (a.epi == EPI_ETA) {
      const float sp = v[0] > 20.0f ? v[0] : log1pf(__expf(v[0]));  // 0.01 * softplus (droid_net.py:410,429)
      a.fout[m] = 0.01f * (float)(half_t)sp;
    } else {
      half_t* dst = a.y + m * a.y_ctot + a.y_coff;
      for (int q = 0; q < a.Cout; ++q) dst[q] = (half_t)act_apply(v[q], a.act);
    }
  }
}

// ---- global-context gate (droid_net.py:392-393): glo[e, c] += sum_p sigmoid(W net[e, p] + b)[c] * net[e, p, c] with
// a 128 -> 128 1x1 convolution.  The general 1x1 path streams `net` through LDS for the MFMA and reads it a second
// time from HBM for the gate product; here the 256-pixel tile [256][128] stays resident in LDS for both uses (217 MB
// of HBM reads per launch instead of 434 MB) and the weights live in registers.
constexpr int GLO_PITCH = 136;  // halves per staged pixel
constexpr size_t GLO_LDS = 256 * GLO_PITCH * 2 + 128 * 4;

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv1x1_glo_kernel(ConvArgs a) {
  extern __shared__ __align__(16) unsigned char lds[];
  half_t* xt = reinterpret_cast<half_t*>(lds);                          // [256 px][136]
  float* gacc = reinterpret_cast<float*>(lds + 256 * GLO_PITCH * 2);    // [128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kg = lane >> 4;
  const int wq = wave & 3, wp = wave >> 2;  // cout quarter (32 couts), pixel half (128 px)
  const int HW = a.H * a.W;
  const int tpi = (HW + 255) / 256;  // tiles per image: a tile never straddles images, the last one may be short
  const int e = blockIdx.x / tpi, p0 = (blockIdx.x % tpi) * 256;
  const int64_t pix0 = (int64_t)e * HW + p0;
  if (tid < 128) gacc[tid] = 0.0f;
  // tile -> LDS (16 B per lane, 256 contiguous bytes per pixel); rows past the image are zero: sigmoid(.) * 0 adds nothing
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int i = tid + 512 * it, px = i >> 4, ck = i & 15;
    half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (p0 + px < HW) v = *reinterpret_cast<const half8*>(a.x0 + (pix0 + px) * a.x0_ctot + a.x0_coff + ck * 8);
    *reinterpret_cast<half8*>(xt + px * GLO_PITCH + ck * 8) = v;
  }
  // weights of this wave: couts 32 wq + 16 i + l16, k = 32 s + 8 kg (packed [k / 64][Cout_pad][64])
  half8 af[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const int k = 32 * s2 + 8 * kg, r = 32 * wq + 16 * i + l16;
      af[i][s2] = *reinterpret_cast<const half8*>(a.w + ((int64_t)(k >> 6) * a.Cout_pad + r) * 64 + (k & 63));
    }
  float4 bv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) bv[i] = *reinterpret_cast<const float4*>(a.bias + 32 * wq + 16 * i + 4 * kg);
  __syncthreads();
  float gs[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 2
  for (int j = 0; j < 8; ++j) {  // 16-pixel groups of this wave's 128 pixels
    const int pxb = wp * 128 + j * 16 + l16;
    float4v acc[2] = {float4v{0.f, 0.f, 0.f, 0.f}, float4v{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const half8 xf = *reinterpret_cast<const half8*>(xt + pxb * GLO_PITCH + 32 * s2 + 8 * kg);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][s2], xf, acc[i], 0, 0, 0);
    }
    // D rows = couts 32 wq + 16 i + 4 kg + r, column = pixel pxb
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const half4 nv = *reinterpret_cast<const half4*>(xt + pxb * GLO_PITCH + 32 * wq + 16 * i + 4 * kg);
      const float b4[4] = {bv[i].x, bv[i].y, bv[i].z, bv[i].w};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        gs[i][r] += (float)(half_t)act_apply(acc[i][r] + b4[r], VIPE_ACT_SIGMOID) * (float)nv[r];
    }
  }
  // sum over the 16 pixel lanes of each kg group (DPP row reductions), then over the workgroup in LDS
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = gs[i][r];
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
      if (l16 == 0) atomicAdd(&gacc[32 * wq + 16 * i + 4 * kg + r], v);
    }
  __syncthreads();
  if (tid < a.Cout) atomicAdd(a.fout + (int64_t)e * a.Cout + tid, gacc[tid]);
}

// ---- 7x7, 4 input channels (the flow encoder's first conv, droid_net.py:447): K = 49 taps x 4 = 196 -> 224.
// The whole packed weight tensor (4 blocks of [128 cout][64 k], 64 KiB) is brought to LDS once by LDS-DMA; the
// (4 + 6) x (64 + 6) pixel halo of the tile (8 B per pixel, 5.5 KiB) is staged through registers.  A 16x16x32
// B fragment is two taps x 4 channels per lane = two ds_read_b64 from the halo; no K pipeline is needed (7 steps).
constexpr int C7_PW = HALO_TW + 6, C7_ROWS = HALO_TH + 6;  // 70 x 10 halo pixels
constexpr int C7_WBYTES = 4 * 128 * 128;                    // 65536
constexpr int C7_XBYTES = ((C7_PW * C7_ROWS * 8 + 15) / 16) * 16;
constexpr int C7_LDS = (C7_WBYTES + C7_XBYTES) > (256 * 136 * 2) ? (C7_WBYTES + C7_XBYTES) : (256 * 136 * 2);

// FLAT: the flat tiling of conv_halo32_kernel with THREE pad columns (pitch Wp = W + 3): tap (ty, tx) of position q is
// position q + (ty - 3) Wp + (tx - 3) of one run of 256 + 6 Wp + 6 halo pixels.
template <bool FLAT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv7x7_c4_kernel(ConvArgs a, int gy) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int tile = L / gy, cout0 = (L % gy) * 128;
  const int xsegs = a.W / HALO_TW, tiles_per_img = FLAT ? a.f_tiles : (a.H / HALO_TH) * xsegs;
  const int e = tile / tiles_per_img, trem = tile % tiles_per_img;
  const int y0 = (trem / xsegs) * HALO_TH, x0 = (trem % xsegs) * HALO_TW;
  const int q0 = trem * (HALO_TH * HALO_TW), pitch = FLAT ? a.f_wp : C7_PW;
  auto pix_of = [&](int pl) -> int64_t {
    if constexpr (FLAT) return flat_pixel(q0 + pl, e, a.H, a.W, a.f_wp, a.f_hwp, a.f_magic);
    else return ((int64_t)(e * a.H + y0 + (pl >> 6))) * a.W + x0 + (pl & 63);
  };
  const int l16 = lane & 15, lk = lane >> 4;
  const unsigned ldsW_a = lds_address(lds);
  unsigned char* ldsX = lds + C7_WBYTES;

  // weights: 64 one-KiB pieces (8 rows x 128 B), 8 per wave; swizzle applied on the source chunk
  {
    const int r8 = lane >> 3, sl = lane & 7;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int pce = wave * 8 + q;          // 0..63: block pce / 16, rows (pce % 16) * 8 ..
      const int blk = pce >> 4, row = (pce & 15) * 8 + r8;
      const half_t* src = a.w + ((int64_t)blk * a.Cout_pad + cout0 + row) * 64 + ((sl ^ ((row >> 1) & 7)) << 3);
      glds16(src, ldsW_a + pce * 1024);
    }
  }
  // halo pixels (zero outside the image)
  if constexpr (FLAT) {
    const int nh = HALO_TH * HALO_TW + 6 * pitch + 6;
    for (int i = tid; i < nh; i += 512) {
      const int m = flat_pixel(q0 - 3 * pitch - 3 + i, e, a.H, a.W, a.f_wp, a.f_hwp, a.f_magic);
      uint2 v = make_uint2(0u, 0u);
      if (m >= 0) v = *reinterpret_cast<const uint2*>(a.x0 + (int64_t)m * a.x0_ctot + a.x0_coff);
      *reinterpret_cast<uint2*>(ldsX + i * 8) = v;
    }
  } else {
    for (int i = tid; i < C7_PW * C7_ROWS; i += 512) {
      const int hy = i / C7_PW, hx = i % C7_PW;
      const int y = y0 + hy - 3, x = x0 + hx - 3;
      uint2 v = make_uint2(0u, 0u);
      if (y >= 0 && y < a.H && x >= 0 && x < a.W)
        v = *reinterpret_cast<const uint2*>(a.x0 + ((int64_t)(e * a.H + y) * a.W + x) * a.x0_ctot + a.x0_coff);
      *reinterpret_cast<uint2*>(ldsX + i * 8) = v;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
  typedef _Float16 half4v __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int st = 0; st < 7; ++st) {
    half8 wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      wf[i] = *reinterpret_cast<const half8*>(lds + (st >> 1) * 16384 + swz(wm * 64 + i * 16 + l16, (st & 1) * 4 + lk));
    // this lane's two taps (k = 8 lk .. 8 lk + 7 of the step); taps >= 49 carry zero weights: read tap 48 (finite)
    const int t0 = min(st * 8 + lk * 2, 48), t1 = min(st * 8 + lk * 2 + 1, 48);
    const int wrow = FLAT ? wn * HALO_TW : wn * C7_PW;  // halo index of this wave's first pixel under tap (0, 0)
    const int o0 = (wrow + (t0 / 7) * pitch + l16 + t0 % 7) * 8, o1 = (wrow + (t1 / 7) * pitch + l16 + t1 % 7) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const half4v lo = *reinterpret_cast<const half4v*>(ldsX + o0 + j * 128);
      const half4v hi = *reinterpret_cast<const half4v*>(ldsX + o1 + j * 128);
      const half8 xf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf, acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();
  // epilogue: bias + activation in the accumulator layout, fp16 tile through LDS, 16-byte NHWC stores
  constexpr int PITCH = 136;
  half_t* stage = reinterpret_cast<half_t*>(lds);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cl = wm * 64 + i * 16 + 4 * lk;
    const float4 b = *reinterpret_cast<const float4*>(a.bias + cout0 + cl);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      half4 h;
      h[0] = (half_t)act_apply(acc[i][j][0] + b.x, a.act);
      h[1] = (half_t)act_apply(acc[i][j][1] + b.y, a.act);
      h[2] = (half_t)act_apply(acc[i][j][2] + b.z, a.act);
      h[3] = (half_t)act_apply(acc[i][j][3] + b.w, a.act);
      *reinterpret_cast<half4*>(stage + (wn * 64 + j * 16 + l16) * PITCH + cl) = h;
    }
  }
  __syncthreads();
  const int ch = (tid & 15) * 8, co = cout0 + ch;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int pl = (tid >> 4) + 32 * it;
    const int64_t m = pix_of(pl);
    if (FLAT && m < 0) continue;
    const half8 o = *reinterpret_cast<const half8*>(stage + pl * PITCH + ch);
    half_t* dst = a.y + m * a.y_ctot + a.y_coff + co;
    if (co + 8 <= a.Cout) {
      *reinterpret_cast<half8*>(dst) = o;
    } else {
      for (int q = 0; q < 8 && co + q < a.Cout; ++q) dst[q] = o[q];
    }
  }
}

// OIHW (fp16 or fp32) -> packed [K_pad/64][Cout_pad][64] fp16, k = tap*Cin_pad + c (generic) or tap*4 + c (Cin == 4)
__global__ void pack_weights_kernel(const void* __restrict__ src, half_t* __restrict__ dst, int Cout, int Cin, int KH,
                                    int KW, int Cout_pad, int Cin_pad, int K_pad, int src_f32, int smallcin) {
  const int64_t total = (int64_t)K_pad * Cout_pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kin = (int)(i % 64);
    const int co = (int)((i / 64) % Cout_pad);
    const int s = (int)(i / (64 * (int64_t)Cout_pad));
    const int k = s * 64 + kin;
    int tap, c;
    if (smallcin) { tap = k / 4; c = k % 4; }
    else { tap = k / Cin_pad; c = k % Cin_pad; }
    float v = 0.0f;
    if (co < Cout && c < Cin && tap < KH * KW) {
      const int64_t o = (((int64_t)co * Cin + c) * KH + tap / KW) * KW + tap % KW;
      v = src_f32 ? reinterpret_cast<const float*>(src)[o] : (float)reinterpret_cast<const half_t*>(src)[o];
    }
    dst[i] = (half_t)v;
  }
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------------

extern "C" {

// Geometry helpers shared with the Python side: padded sizes of the packed weight tensor.
VIPE_EXPORT int vipe_conv_packed_dims(int Cout, int Cin, int KH, int KW, int* cout_pad, int* cin_pad, int* k_pad) {
  if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return VIPE_EINVAL;
  const int cp = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : round_up(Cout, 128));
  int cinp, kp;
  if (Cin == 4) { cinp = 4; kp = round_up(KH * KW * 4, 64); }
  else { cinp = round_up(Cin, 64); kp = KH * KW * cinp; }
  if (cout_pad) *cout_pad = cp;
  if (cin_pad) *cin_pad = cinp;
  if (k_pad) *k_pad = kp;
  return VIPE_OK;
}

VIPE_EXPORT int vipe_conv_pack_weights(const void* d_w_oihw, void* d_w_packed, int Cout, int Cin, int KH, int KW,
                                       int src_dtype, void* stream) {
  VIPE_CHECK_ARG(d_w_oihw && d_w_packed);
  VIPE_CHECK_ARG(src_dtype == VIPE_F16 || src_dtype == VIPE_F32);
  int cp, cinp, kp;
  if (vipe_conv_packed_dims(Cout, Cin, KH, KW, &cp, &cinp, &kp) != VIPE_OK) return VIPE_EINVAL;
  VIPE_CHECK_ARG(Cin == 4 || Cin % 8 == 0);
  const int64_t total = (int64_t)kp * cp;
  pack_weights_kernel<<<(int)std::min<int64_t>((total + 255) / 256, 4096), 256, 0, as_stream(stream)>>>(
      d_w_oihw, (half_t*)d_w_packed, Cout, Cin, KH, KW, cp, cinp, kp, src_dtype == VIPE_F32, Cin == 4);
  return vipe_launch_status();
}

}  // extern "C"

namespace {

// one hipFuncSetAttribute per (kernel, device): the dynamic-LDS ceiling of every kernel that may ask for > 64 KiB
template <typename K>
void allow_lds(K kernel, std::atomic<uint64_t>& seen, size_t bytes) {
  vipe_once_per_device(seen, [&] { (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); });
}
constexpr size_t MAX_LDS = 160 * 1024;

// flat tiling of an H x W image with `pad` zero columns per row (see conv_halo32_kernel): fills a.f_*; false when a
// halo buffer would not fit the kernel's four 1-KiB pieces per wave
bool flat_geometry(ConvArgs& a, int pad, bool is1x1) {
  a.f_wp = a.W + pad;
  a.f_hwp = a.H * a.f_wp;
  a.f_tiles = (a.f_hwp + 255) / 256;
  a.f_magic = (unsigned)(((1ull << 32) + (unsigned)a.f_wp - 1) / (unsigned)a.f_wp);
  const int rows = 256 + 2 * a.f_wp + 2;
  a.f_pieces = (rows + 15) / 16;
  a.f_xbytes = a.f_pieces * 1024;
  a.f_p1lo = (a.f_wp + 1) / 16;
  a.f_p1hi = (a.f_wp + 256) / 16;
  (void)is1x1;
  return a.f_wp >= 2 && a.f_pieces <= 8 * H32_XP && a.f_hwp < (1 << 23);
}

template <int BMC, int KS, int VAR, bool FLAT>
int launch_halo32(const ConvArgs& a, int tiles, int gy, size_t lds, hipStream_t s) {
  static std::atomic<uint64_t> seen{0};
  allow_lds(conv_halo32_kernel<BMC, KS, true, VAR, FLAT>, seen, FLAT ? MAX_LDS : lds);
  conv_halo32_kernel<BMC, KS, true, VAR, FLAT><<<dim3(tiles * gy), 512, lds, s>>>(a, gy);
  return vipe_launch_status();
}

template <bool FLAT>
int launch_halo_family(ConvArgs& a, int cp, int64_t M, hipStream_t s) {
  const bool split_ok = a.split >= a.Cin || a.split % H32_BK == 0;
  const int tiles = FLAT ? a.B * a.f_tiles : (int)(M / (HALO_TH * HALO_TW));
  const size_t xb = FLAT ? (size_t)a.f_xbytes : (size_t)H32_XBYTES;
  if (a.accinit || a.epi == EPI_PARTIAL || a.ai_f32) {
    // initial accumulators / raw partial sums: the 16x16x32 halo kernel with >= 64 output channels (fp32 initial
    // accumulators and partial sums: 3x3, >= 128 output channels)
    const bool ok = cp >= 64 && split_ok && !(a.KH == 3 && a.Cout <= 16) && !(a.epi == EPI_PARTIAL && a.Cout % 4 != 0) &&
                    !(a.ai_f32 && !a.accinit);
    if (!ok) return VIPE_EUNSUPPORTED;
  }
  if (a.KH == 3 && a.Cout <= 16 && cp == 32 && split_ok && (a.epi == EPI_HEADS || a.epi == EPI_ETA || a.epi == EPI_PLAIN)) {
    static std::atomic<uint64_t> seen{0};
    const size_t lds = 2 * NRW_WBYTES + 2 * xb + 1024;
    allow_lds(conv_halo32_narrow_kernel<FLAT>, seen, FLAT ? MAX_LDS : lds);
    conv_halo32_narrow_kernel<FLAT><<<dim3(tiles), 512, lds, s>>>(a);
    return vipe_launch_status();
  }
  if (!split_ok) return VIPE_EUNSUPPORTED;
  const int bmc = cp >= 128 ? 128 : cp;
  const int gy = cp >= 128 ? cp / 128 : 1;
  // K-loop buffers (weight ring + two halo buffers + DMA sink) or the epilogue's staged tile, whichever is larger
  size_t lds = 3 * (size_t)bmc * 64 + 2 * xb + 1024;
  const size_t stage = bmc >= 64 ? (size_t)256 * (bmc + 8) * 2 : (size_t)256 * 36 * 4;
  if (lds < stage) lds = stage;
  if (a.ai_f32 || a.epi == EPI_PARTIAL) {
    // the two variants of the staged z|r gates (vipe_update_gate_state): 3x3, >= 128 output channels
    if (!(a.KH == 3 && bmc == 128 && !(a.ai_f32 && a.epi == EPI_PARTIAL))) return VIPE_EUNSUPPORTED;
    return a.ai_f32 ? launch_halo32<128, 3, 1, FLAT>(a, tiles, gy, lds, s) : launch_halo32<128, 3, 2, FLAT>(a, tiles, gy, lds, s);
  }
  if (a.KH == 3) {
    if (bmc == 128) return launch_halo32<128, 3, 0, FLAT>(a, tiles, gy, lds, s);
    if (bmc == 64) return launch_halo32<64, 3, 0, FLAT>(a, tiles, 1, lds, s);
    return launch_halo32<32, 3, 0, FLAT>(a, tiles, 1, lds, s);
  }
  if (bmc == 128) return launch_halo32<128, 1, 0, FLAT>(a, tiles, gy, lds, s);
  if (bmc == 64) return launch_halo32<64, 1, 0, FLAT>(a, tiles, 1, lds, s);
  return launch_halo32<32, 1, 0, FLAT>(a, tiles, 1, lds, s);
}

int launch_conv(ConvArgs& a, hipStream_t s) {
  int cp, cinp, kp;
  if (vipe_conv_packed_dims(a.Cout, a.Cin, a.KH, a.KW, &cp, &cinp, &kp) != VIPE_OK) return VIPE_EINVAL;
  a.Cout_pad = cp;
  a.Cin_pad = cinp;
  a.nsteps = kp / 64;
  const int64_t M = (int64_t)a.B * a.H * a.W;
  if (M == 0) return VIPE_OK;
  const bool small = a.Cin == 4;
  const int gx = (int)((M + BNP - 1) / BNP);
  // kernel families.  Tile kernels (halo in LDS once per 32-channel chunk): the 4 x 64 tiling when the image is made of
  // such tiles, the FLAT tiling (256 consecutive positions of the row-padded image) for every other shape up to
  // W = 126; wider ragged images take the per-tap gather kernel.
  const bool sq13 = !small && a.KH == a.KW && (a.KH == 1 || a.KH == 3);
  const bool off32 = (int64_t)a.B * a.H * a.W < (1ll << 31);  // pixel indices are ints; byte offsets are formed in 64 bits
  const bool aligned = a.W % HALO_TW == 0 && a.H % HALO_TH == 0;
  const bool halo = sq13 && aligned && off32;
  const bool flat = sq13 && !aligned && off32 && flat_geometry(a, a.KH == 3 ? 1 : 0, a.KH == 1);
  if ((a.accinit || a.epi == EPI_PARTIAL || a.ai_f32) && !halo && !flat) return VIPE_EUNSUPPORTED;
  if (a.epi == EPI_GLO && a.KH == 1 && a.KW == 1 && a.Cin == 128 && a.Cout == 128 && cp == 128 && a.split >= a.Cin &&
      a.x0_ctot % 8 == 0 && a.x0_coff % 8 == 0 && a.net == a.x0 && a.net_ctot == a.x0_ctot && a.net_coff == a.x0_coff &&
      a.extra == nullptr) {
    static std::atomic<uint64_t> seen{0};
    allow_lds(conv1x1_glo_kernel, seen, GLO_LDS);
    conv1x1_glo_kernel<<<dim3((unsigned)(a.B * ((a.H * a.W + 255) / 256))), 512, GLO_LDS, s>>>(a);
    return vipe_launch_status();
  }
  if (halo) {
    const int rc = launch_halo_family<false>(a, cp, M, s);
    if (rc != VIPE_EUNSUPPORTED || a.accinit || a.epi == EPI_PARTIAL || a.ai_f32) return rc;
  } else if (flat) {
    const int rc = launch_halo_family<true>(a, cp, M, s);
    if (rc != VIPE_EUNSUPPORTED || a.accinit || a.epi == EPI_PARTIAL || a.ai_f32) return rc;
  }
  if (small && a.KH == 7 && a.KW == 7 && cp % 128 == 0 && kp == 256 && a.epi == EPI_PLAIN && a.extra == nullptr &&
      (int64_t)a.B * a.H * a.W < (1ll << 31)) {
    const int gy = cp / 128;
    if (aligned) {
      static std::atomic<uint64_t> seen{0};
      allow_lds(conv7x7_c4_kernel<false>, seen, C7_LDS);
      conv7x7_c4_kernel<false><<<dim3((int)(M / (HALO_TH * HALO_TW)) * gy), 512, C7_LDS, s>>>(a, gy);
      return vipe_launch_status();
    }
    flat_geometry(a, 3, false);
    const size_t need = C7_WBYTES + (size_t)(256 + 6 * a.f_wp + 6) * 8;
    if (need <= MAX_LDS && a.f_hwp < (1 << 23) && a.f_wp >= 2) {
      static std::atomic<uint64_t> seen{0};
      allow_lds(conv7x7_c4_kernel<true>, seen, MAX_LDS);
      const size_t lds = need > (size_t)(256 * 136 * 2) ? need : (size_t)(256 * 136 * 2);
      conv7x7_c4_kernel<true><<<dim3(a.B * a.f_tiles * gy), 512, lds, s>>>(a, gy);
      return vipe_launch_status();
    }
  }
  // per-tap gather kernels: any shape
  const bool glds = !small && a.KH * a.KW <= 32;
  static std::atomic<uint64_t> s1{0}, s2{0}, s3{0};
  allow_lds(conv_mfma_kernel<128, 2, 2, false>, s1, 65536);
  allow_lds(conv_mfma_kernel<128, 2, 2, true>, s2, 65536);
  allow_lds(conv_mfma_glds_kernel<128, 2, 2>, s3, 65536);
  if (cp >= 128) {
    const size_t lds = 2 * (128 + BNP) * 128;
    const int gy = cp / 128;
    if (small) conv_mfma_kernel<128, 2, 2, true><<<dim3(gx, gy), 256, lds, s>>>(a);
    else if (glds) conv_mfma_glds_kernel<128, 2, 2><<<dim3(gx * gy), 256, lds, s>>>(a, gy);
    else conv_mfma_kernel<128, 2, 2, false><<<dim3(gx, gy), 256, lds, s>>>(a);
  } else if (cp == 64) {
    const size_t lds = 2 * (64 + BNP) * 128;
    if (small) return VIPE_EUNSUPPORTED;
    if (glds) conv_mfma_glds_kernel<64, 1, 4><<<dim3(gx), 256, lds, s>>>(a, 1);
    else conv_mfma_kernel<64, 1, 4, false><<<dim3(gx, 1), 256, lds, s>>>(a);
  } else {
    const size_t lds = 2 * (32 + BNP) * 128;
    if (small) return VIPE_EUNSUPPORTED;
    if (glds) conv_mfma_glds_kernel<32, 1, 4><<<dim3(gx), 256, lds, s>>>(a, 1);
    else conv_mfma_kernel<32, 1, 4, false><<<dim3(gx, 1), 256, lds, s>>>(a);
  }
  return vipe_launch_status();
}

}  // namespace

extern "C" {

VIPE_EXPORT int vipe_conv2d_nhwc_f16(const void* d_x, const void* d_w_packed, const float* d_bias,
                                     const float* d_extra, void* d_y, int B, int H, int W, int Cin, int cin_total,
                                     int cin_off, int Cout, int cout_total, int cout_off, int KH, int KW, int act,
                                     void* stream) {
  VIPE_CHECK_ARG(d_x && d_w_packed && d_bias && d_y);
  VIPE_CHECK_ARG(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (KH & 1) && (KW & 1));
  VIPE_CHECK_ARG((Cin == 4 || Cin % 8 == 0) && cin_total % 4 == 0 && cin_off % 4 == 0 && cout_total % 4 == 0 &&
                 cout_off % 4 == 0);
  VIPE_CHECK_ARG(Cin == 4 || (cin_total % 8 == 0 && cin_off % 8 == 0));
  VIPE_CHECK_ARG(act >= 0 && act <= 3);
  ConvArgs a = {};
  a.x0 = (const half_t*)d_x; a.x0_ctot = cin_total; a.x0_coff = cin_off;
  a.x1 = a.x0; a.x1_ctot = cin_total; a.x1_coff = cin_off; a.split = Cin;
  a.w = (const half_t*)d_w_packed; a.bias = d_bias; a.extra = d_extra; a.extra_stride = Cout; a.extra_off = 0;
  a.y = (half_t*)d_y; a.y_ctot = cout_total; a.y_coff = cout_off;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW;
  a.act = act; a.epi = EPI_PLAIN;
  return launch_conv(a, as_stream(stream));
}

// [fused] GRU / head epilogues.  `mode`: 1 GLO, 2 ZR, 3 Q, 4 HEADS, 5 ETA (see conv_mfma.hip); two-source input.
VIPE_EXPORT int vipe_conv2d_fused(const void* d_x0, int x0_ctot, int x0_coff, const void* d_x1, int x1_ctot,
                                  int x1_coff, int split, const void* d_w_packed, const float* d_bias,
                                  const float* d_extra, int extra_stride, int extra_off, void* d_y, int y_ctot,
                                  int y_coff, void* d_y2, int y2_ctot, int y2_coff, const void* d_net, int net_ctot,
                                  int net_coff, const void* d_z, float* d_fout, const void* d_accinit, int ai_ctot,
                                  int ai_coff, int B, int H, int W, int Cin, int Cout, int KH, int KW, int act, int mode,
                                  void* stream) {
  VIPE_CHECK_ARG(d_x0 && d_w_packed && d_bias);
  VIPE_CHECK_ARG(!d_accinit || (ai_ctot % 4 == 0 && ai_coff % 4 == 0 && ai_coff + Cout <= ai_ctot));
  VIPE_CHECK_ARG(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (KH & 1) && (KW & 1));
  if (Cin == 4) {
    VIPE_CHECK_ARG(split >= Cin && x0_ctot % 4 == 0 && x0_coff % 4 == 0);
  } else {
    VIPE_CHECK_ARG(Cin % 8 == 0 && x0_ctot % 8 == 0 && x0_coff % 8 == 0);
    VIPE_CHECK_ARG(split >= Cin || (split % 64 == 0 && d_x1 && x1_ctot % 8 == 0 && x1_coff % 8 == 0));
  }
  const int ai_f32 = (mode & VIPE_CONV_ACCINIT_F32) != 0;
  mode &= ~VIPE_CONV_ACCINIT_F32;
  VIPE_CHECK_ARG(mode >= EPI_PLAIN && mode <= EPI_PARTIAL);
  if (mode == EPI_PARTIAL) VIPE_CHECK_ARG(d_fout && y_ctot % 4 == 0 && y_coff % 4 == 0 && y_coff + Cout <= y_ctot && !d_extra);
  if (mode == EPI_PLAIN) VIPE_CHECK_ARG(d_y && y_ctot % 4 == 0 && y_coff % 4 == 0);
  if (mode == EPI_GLO) VIPE_CHECK_ARG(d_net && d_fout);
  if (mode == EPI_ZR) VIPE_CHECK_ARG(d_y && d_y2 && d_net && Cout == 256);
  if (mode == EPI_Q) VIPE_CHECK_ARG(d_y && d_net && d_z && Cout == 128);
  if (mode == EPI_HEADS) VIPE_CHECK_ARG(d_fout && Cout == 4);
  if (mode == EPI_ETA) VIPE_CHECK_ARG(d_fout && Cout == 1);
  ConvArgs a = {};
  a.x0 = (const half_t*)d_x0; a.x0_ctot = x0_ctot; a.x0_coff = x0_coff;
  a.x1 = (const half_t*)(d_x1 ? d_x1 : d_x0); a.x1_ctot = d_x1 ? x1_ctot : x0_ctot; a.x1_coff = d_x1 ? x1_coff : x0_coff;
  a.split = split > Cin ? Cin : split;
  a.w = (const half_t*)d_w_packed; a.bias = d_bias; a.extra = d_extra; a.extra_stride = extra_stride; a.extra_off = extra_off;
  a.y = (half_t*)d_y; a.y_ctot = y_ctot; a.y_coff = y_coff;
  a.y2 = (half_t*)d_y2; a.y2_ctot = y2_ctot; a.y2_coff = y2_coff;
  a.net = (const half_t*)d_net; a.net_ctot = net_ctot; a.net_coff = net_coff;
  a.zbuf = (const half_t*)d_z; a.fout = d_fout;
  a.accinit = (const half_t*)d_accinit; a.ai_ctot = ai_ctot; a.ai_coff = ai_coff; a.ai_f32 = ai_f32;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW;
  a.act = act; a.epi = mode;
  return launch_conv(a, as_stream(stream));
}

}  // extern "C"
