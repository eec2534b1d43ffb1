// CorrBlock.corr + pyramid (vipe/slam/networks/droid_net.py:56-69, 94-102) in ONE kernel:
//   level0[e][p1][p2] = sum_c (f1[e][c][p1] / 4) * (f2[e][c][p2] / 4)        (fp16 in, fp32 accumulate, fp16 out)
//   level i+1 = avg_pool2d(level i, 2, 2) over the target dims (fp32 sum of the four halves, rounded to half -
//   the arithmetic of at::native::avg_pool2d on c10::Half), i = 0..2.
// The reference runs torch.matmul + three avg_pool2d passes that re-read what they just wrote (25 MB/edge written,
// 8.3 MB/edge re-read and 8.3 MB written again by the pools); here every level-0 tile is pooled while it is still in
// LDS / registers, so HBM sees the algorithmic 26.7 MB/edge of writes once.  The kernel is HBM-write bound
// (2.42 GFLOP/edge is ~1 us of MFMA per edge), so the design optimises store shape, not the matrix pipeline.
//
// Mapping.  Workgroup (8 waves) = 64 consecutive source pixels p1 of one edge x ALL target pixels, walked in chunks of
// 2 target rows x 64 columns.  Feature maps stay in the reference's [C][h*w] layout: both MFMA operands are
// K-major in memory, so the LDS images are [channel][pixel] rows filled with coalesced 16-byte loads and the
// 16x16x32 fragments are gathered with ds_read_b64_tr_b16 (hardware transpose read).  Row pitches are 32 bytes past a
// multiple of 256 and a lane group's 4+4 channels are 4 apart (a consistent K permutation of both operands), so the
// two blocks of each 32-lane half cover 8 consecutive channel rows = 8 distinct 32-byte bank segments.
// f2 is the A operand (rows = target pixels): a lane's accumulator registers are then 4 consecutive target pixels of
// one source pixel -> 8-byte staging writes of the [p1][p2] tile.  The 64-pixel f1 tile is held in registers.
// Pooling: the thread that stores columns 8s..8s+7 of both chunk rows of source pixel p1 also produces level-1
// columns 4s..4s+3, keeps them to form level-2 columns 2s, 2s+1 two chunks later and level-3 column s after eight
// target rows - no cross-thread traffic after the level-0 tile.
//
// Two output layouts (template parameter BLK).  Reference layout: level i = [e][p1][h>>i][w>>i], what CorrBlock exposes;
// the chunk is 2 target rows x 64 columns and a source pixel's piece of it is 256 contiguous bytes.  BLOCKED layout
// (the internal layout of the pooled store the update iteration reads, VIPE_PYRAMID_BLOCKED in the header): the chunk
// is 4 target rows x 32 columns and levels 0 / 1 are stored as
//     level0[e][p1 / 64][chunk][p1 % 64][tile 0..3][row 0..3][col 0..7]      chunk = (x / 32) * (h / 4) + y / 4
//     level1[e][p1 / 64][chunk / 2][p1 % 64][tile 0..1][row 0..3][col 0..7]  (level-1 rows 4 (y / 8) .., cols 16 (x / 32) ..)
// i.e. (a) the 64 x 128 tile a workgroup produces per chunk is ONE contiguous 16 KiB run (8 KiB per two chunks at
// level 1) instead of 64 pieces 6 KiB apart - DRAM pages are written whole - and (b) a 64-byte sector holds a
// 4 x 8 patch of a source pixel's slab, so the 8 x 8 lookup window touches 5.2 sectors on average instead of 10
// (8 rows x 1.25).  Levels 2 / 3 (384 / 96 bytes per source pixel) keep the reference layout.
//
// Any grid (template parameter GEN, blocked layout only).  The reference's resize produces 41 x 73 grids for 16:9
// video (vipe/slam/system.py:46-59): an odd pixel count, so the rows of the [C][h*w] maps are not even 4-byte aligned.
// vipe_corr_prep first rewrites every frame's map ONCE into the two operand images the kernel wants, zero padded:
//     A: [C][G * 64]            pixel order, G = ceil(h w / 64) groups of source pixels
//     B: [C][S * R][4 x 32]     chunk order, S = ceil(w / 32) strips, R = 2 ceil(h / 8) row groups of 4
// (1.9 MB per 41 x 73 frame against 27 MB of pyramid per edge), and the kernel runs on them with the aligned kernel's
// thread roles: 16-byte loads, no bounds tests in the chunk loop.  The store keeps the blocked layout on the padded
// dimensions (include/vipe_amd.h); tiles wholly outside the w x h targets are neither written nor ever read, entries of
// the pooled levels outside (h >> i) x (w >> i) are written as zero - F.avg_pool2d floors (droid_net.py:66-68), and the
// lookup treats whatever lies beyond as zero.
#include <algorithm>

#include "common.cuh"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

constexpr int PB_M = 64;                 // source pixels per workgroup
constexpr int PB_N = 128;                // target pixels per chunk (2 rows x 64 columns)
constexpr int PB_C = 128;                // channels
constexpr int PB_PA = PB_N * 2 + 32;     // byte pitch of the f2 chunk image rows (288)
constexpr int PB_PB = PB_M * 2 + 32;     // byte pitch of the f1 tile image rows (160)
constexpr int PB_PS = PB_N + 8;          // halves per staged source pixel (136)
constexpr int PB_IMGA = PB_C * PB_PA;    // 36864
constexpr int PB_STAGE = PB_M * PB_PS * 2;  // 17408: one staged level-0 tile [64 p1][128 p2 + pad] fp16
constexpr int PB_LDS = PB_IMGA + 2 * PB_STAGE;  // f2 chunk image + two stage buffers (the f1 image borrows imgA first)

struct BuildArgs {
  const half_t* f1;      // [*, C, h*w] feature maps: edge e reads map idx1[e] (e itself when idx1 == nullptr)
  const half_t* f2;
  const int64_t* idx1;
  const int64_t* idx2;
  const int* slots;      // edge e is written to slot slots[e] of the level buffers (e itself when nullptr)
  half_t* lv[4];
  int B, h, w, nlev;
  int64_t base;          // GEN: f1 is the prepared store of frames [base, base + n): edge e reads frames idx1[e] - base, ..
  int64_t n_prep;        // GEN: n; an edge with a frame outside the store is skipped (its slot is left as it was)
};

// geometry of the padded blocked layout (shared with corr_lookup.hip through include/vipe_amd.h's description)
struct GenDims {
  int G, S, R, nch, w2p, w3p;
  int64_t fstride;  // halves per prepared frame
  __host__ __device__ GenDims(int h, int w)
      : G((h * w + 63) / 64), S((w + 31) / 32), R(((h + 7) / 8) * 2), nch(S * R), w2p(8 * S), w3p((4 * S + 7) / 8 * 8),
        fstride((int64_t)PB_C * ((int64_t)G * 64 + (int64_t)nch * 128)) {}
};

// [n][C][h*w] -> the prepared operand images A | B of every frame (see the file header); one thread per 8 output halves
__global__ __launch_bounds__(256) void corr_prep_kernel(const half_t* __restrict__ src, half_t* __restrict__ dst, int n, int h,
                                                        int w) {
  const GenDims d(h, w);
  const int P = h * w;
  const int64_t per = d.fstride / 8;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per * n) return;
  const int f = (int)(i / per);
  const int64_t o = (i % per) * 8;
  const int64_t asz = (int64_t)PB_C * d.G * 64;
  const half_t* s = src + (int64_t)f * PB_C * P;
  half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (o < asz) {
    const int c = (int)(o / (d.G * 64)), p = (int)(o % (d.G * 64));
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (p + k < P) v[k] = s[(int64_t)c * P + p + k];
  } else {
    const int64_t ob = o - asz;
    const int c = (int)(ob / (d.nch * 128)), r = (int)(ob % (d.nch * 128));
    const int ch = r >> 7, px = r & 127;
    const int y = (ch % d.R) * 4 + (px >> 5), x = (ch / d.R) * 32 + (px & 31);
    if (y < h) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (x + k < w) v[k] = s[(int64_t)c * P + y * w + x + k];
    }
  }
  *reinterpret_cast<half8*>(dst + (int64_t)f * d.fstride + o) = v;
}

__device__ __forceinline__ half8 tr_frag(const unsigned char* img, int byte_off, int pitch16) {
  // two transposed 4 x 16 blocks, 16 channel rows apart -> the 8 k values of this lane
  const auto p0 = (const __attribute__((address_space(3))) fp16x4*)(img + byte_off);
  const auto p1 = (const __attribute__((address_space(3))) fp16x4*)(img + byte_off + pitch16);
  const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)p0);
  const fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)p1);
  half8 r;
  r[0] = (half_t)a[0]; r[1] = (half_t)a[1]; r[2] = (half_t)a[2]; r[3] = (half_t)a[3];
  r[4] = (half_t)b[0]; r[5] = (half_t)b[1]; r[6] = (half_t)b[2]; r[7] = (half_t)b[3];
  return r;
}

__device__ __forceinline__ half_t pool4(half_t a, half_t b, half_t c, half_t d) {
  // at::native avg_pool2d<c10::Half, float>: fp32 sum in window order, divide, round to half
  return (half_t)((((float)a + (float)b) + (float)c + (float)d) / 4.0f);
}

template <bool BLK, bool GEN = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void corr_pyramid_build_kernel(BuildArgs a) {
  static_assert(BLK || !GEN, "the general-shape kernel writes the blocked layout");
  extern __shared__ __align__(16) unsigned char lds[];
  unsigned char* imgA = lds;             // f2 chunk  [128 c][128 p2]; before the first chunk: the f1 tile [128 c][64 p1]
  unsigned char* imgB = lds;             // (f1 tile image, 20480 B <= imgA)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const GenDims gd(a.h, a.w);
  const int P = GEN ? gd.G * 64 : a.h * a.w;   // GEN: source pixels padded to whole groups of 64 (slabs exist for all)
  const int tiles = P / PB_M;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int e = L / tiles, p1_0 = (L % tiles) * PB_M;
  if (GEN) {  // whole workgroups leave (e is uniform): a frame index outside the prepared store would be a wild read
    const int64_t r1 = (a.idx1 ? a.idx1[e] : (int64_t)e) - a.base, r2 = (a.idx2 ? a.idx2[e] : (int64_t)e) - a.base;
    if (r1 < 0 || r1 >= a.n_prep || r2 < 0 || r2 >= a.n_prep) return;
  }
  // operand images: rows of the [C][h*w] maps, or (GEN) of the prepared A / B images of the two frames
  const int pitch1 = P, pitch2 = GEN ? gd.nch * 128 : P;
  const half_t* f1 = GEN ? a.f1 + ((a.idx1 ? a.idx1[e] : (int64_t)e) - a.base) * gd.fstride
                         : a.f1 + (a.idx1 ? a.idx1[e] : (int64_t)e) * PB_C * P;
  const half_t* f2 = GEN ? a.f2 + ((a.idx2 ? a.idx2[e] : (int64_t)e) - a.base) * gd.fstride + (int64_t)PB_C * P
                         : a.f2 + (a.idx2 ? a.idx2[e] : (int64_t)e) * PB_C * P;
  const int64_t es = a.slots ? a.slots[e] : e;

  // ---- f1 tile -> LDS -> registers (B operand: columns = source pixels)
  for (int i = tid; i < PB_C * 8; i += 512) {
    const int c = i >> 3, s = i & 7;
    *reinterpret_cast<half8*>(imgB + c * PB_PB + s * 16) = *reinterpret_cast<const half8*>(f1 + (int64_t)c * pitch1 + p1_0 + s * 8);
  }
  __syncthreads();
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, l16 = lane & 15;
  const int wn = wave & 3, wp = wave >> 2;  // target-pixel quarter of the chunk, source-pixel half of the tile
  half8 bf[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      bf[j][kk] = tr_frag(imgB, (kk * 32 + g * 4 + q) * PB_PB + (wp * 32 + j * 16 + 4 * pp) * 2, 16 * PB_PB);
  const int a_off = (g * 4 + q) * PB_PA + (wn * 32 + 4 * pp) * 2;

  // ---- chunk walk.  Reference layout: (8-row group, 64-column segment, row pair) of 2 x 64 chunks.  Blocked layout:
  // 4 x 32 chunks, down the rows of one 32-column strip, then the next strip (vertically adjacent chunks are
  // consecutive: pairs of them complete the level-1 tiles and the level-3 entries)
  const int csegs = a.w / 64;
  const int nchunks = GEN ? gd.nch : (a.h / 2) * csegs;  // = (h / 4) * (w / 32)
  const int rgs = GEN ? gd.R : a.h / 4;
  auto chunk_origin = [&](int ch, int& y, int& x0) {
    if (BLK) {
      y = (ch % rgs) * 4;
      x0 = (ch / rgs) * 32;
    } else {
      const int rp = ch & 3, t = ch >> 2;
      y = (t / csegs) * 8 + rp * 2;
      x0 = (t % csegs) * 64;
    }
  };
  // this thread's 4 sixteen-byte pieces of a chunk: channel rows tid/16 + 32 k, piece tid % 16 (chunk-local pixels
  // 8 piece ..: row piece / 8 of a 2 x 64 chunk, row piece / 4 of a 4 x 32 chunk)
  half8 pre[4];
  auto fetch = [&](int ch) {
    int y, x0;
    chunk_origin(ch, y, x0);
    const int pc = tid & 15;
    const int off = GEN ? ch * 128 + pc * 8  // the prepared B image is in chunk order
                        : (BLK ? (y + (pc >> 2)) * a.w + x0 + (pc & 3) * 8 : (y + (pc >> 3)) * a.w + x0 + (pc & 7) * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = (tid >> 4) + 32 * k;
      pre[k] = *reinterpret_cast<const half8*>(f2 + (int64_t)c * pitch2 + off);
    }
  };
  auto commit = [&]() {
    const int pc = tid & 15;
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<half8*>(imgA + ((tid >> 4) + 32 * k) * PB_PA + pc * 16) = pre[k];
  };
  fetch(0);
  const int sp1 = tid >> 3, sseg = tid & 7;  // output role: source pixel of the tile, 8-column segment
  const int64_t p1g = es * P + p1_0 + sp1;
  const int64_t eg = es * tiles + (p1_0 / PB_M);  // blocked layout: (slot, group of 64 source pixels)
  half4 l1prev = {0, 0, 0, 0};
  half_t l2prev[2] = {0, 0};
  half_t* stage0 = reinterpret_cast<half_t*>(lds + PB_IMGA);
  // Software pipeline over the chunks: the level-0 tile of chunk ch is staged in buffer ch & 1 and leaves for HBM one
  // iteration later, BEFORE the loads of the next f2 chunk are issued - loads and stores share the in-order vmcnt
  // counter, so a wait for the f2 loads then never has to sit through the HBM write latency of younger stores (it
  // did: 4.3 us per chunk).
  auto output = [&](int ch, const half_t* stage) {
    int y, x0;
    chunk_origin(ch, y, x0);
    if (BLK) {
      // Two roles per thread.  STORE: 16-byte pieces sseg and sseg + 8 of the source pixel's 256-byte run (piece q = row
      // q % 4 of tile q / 4), so that each store instruction covers 128 contiguous bytes per source pixel and the
      // workgroup's two instructions one 16 KiB run.  POOL: (tile sseg / 2, row pair sseg % 2) - the 2 x 8 patch whose
      // level-1 entries this thread forms (re-read from the staged tile: LDS bandwidth is not the limit here).
      const int tile = sseg >> 1, hf = sseg & 1, rg = y >> 2;
      half_t* dst = a.lv[0] + ((eg * nchunks + ch) * PB_M + sp1) * 128;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int q = sseg + 8 * s2;
        if (GEN && (y >= a.h || x0 + (q >> 2) * 8 >= a.w)) continue;  // a tile wholly outside the targets: never read
        *reinterpret_cast<half8*>(dst + q * 8) =
            *reinterpret_cast<const half8*>(stage + sp1 * PB_PS + (q & 3) * 32 + (q >> 2) * 8);
      }
      const half8 r0 = *reinterpret_cast<const half8*>(stage + sp1 * PB_PS + (2 * hf) * 32 + tile * 8);
      const half8 r1 = *reinterpret_cast<const half8*>(stage + sp1 * PB_PS + (2 * hf + 1) * 32 + tile * 8);
      if (a.nlev <= 1) return;
      half4 l1;  // level-1 row hf of the chunk (rows 2 (rg & 1) + hf of the level-1 tile), columns 4 tile .. + 3
#pragma unroll
      for (int k = 0; k < 4; ++k) l1[k] = pool4(r0[2 * k], r0[2 * k + 1], r1[2 * k], r1[2 * k + 1]);
      if (GEN) {  // entries outside the floored (h >> 1) x (w >> 1) level pooled zero padding with real rows / columns
        const int r1l = 2 * rg + hf, c1l = (x0 >> 1) + tile * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (r1l >= (a.h >> 1) || c1l + k >= (a.w >> 1)) l1[k] = (half_t)0;
      }
      if ((rg & 1) && !(GEN && ((x0 >> 1) + (tile >> 1) * 8 >= (a.w >> 1) || 2 * (rg - 1) >= (a.h >> 1)))) {
        // second chunk of the pair: both level-1 rows of this thread leave together (8 KiB run per workgroup)
        half_t* d1 = a.lv[1] + ((eg * (nchunks >> 1) + (ch >> 1)) * PB_M + sp1) * 64 + (tile >> 1) * 32 + (tile & 1) * 4;
        *reinterpret_cast<half4*>(d1 + hf * 8) = l1prev;
        *reinterpret_cast<half4*>(d1 + (2 + hf) * 8) = l1;
      }
      if (a.nlev > 2) {
        // level 2 pools the two level-1 rows of the chunk: the other row lives in the neighbour lane (sseg ^ 1)
        uint2 mine = __builtin_bit_cast(uint2, l1), oth;
        oth.x = __shfl_xor(mine.x, 1, WAVE);
        oth.y = __shfl_xor(mine.y, 1, WAVE);
        const half4 lo = hf ? __builtin_bit_cast(half4, oth) : l1, hi = hf ? l1 : __builtin_bit_cast(half4, oth);
        half_t l2[2];
        l2[0] = pool4(lo[0], lo[1], hi[0], hi[1]);
        l2[1] = pool4(lo[2], lo[3], hi[2], hi[3]);
        if (GEN) {  // rows of R / R/2 entries with pitches 8 S / 4 S rounded up to 8: all of it written, zero outside the level
          const int c2 = (x0 >> 2) + tile * 2;
          if (rg >= (a.h >> 2) || c2 >= (a.w >> 2)) l2[0] = (half_t)0;
          if (rg >= (a.h >> 2) || c2 + 1 >= (a.w >> 2)) l2[1] = (half_t)0;
          if (hf == 0) {
            half_t* d2 = a.lv[2] + (p1g * gd.R + rg) * gd.w2p + c2;
            d2[0] = l2[0];
            d2[1] = l2[1];
            if (a.nlev > 3 && (rg & 1)) {
              const int c3 = (x0 >> 3) + tile;
              half_t* d3 = a.lv[3] + (p1g * (gd.R >> 1) + (rg >> 1)) * gd.w3p + c3;
              const bool in = (rg >> 1) < (a.h >> 3) && c3 < (a.w >> 3);
              d3[0] = in ? pool4(l2prev[0], l2prev[1], l2[0], l2[1]) : (half_t)0;
              if (c3 + 1 == 4 * gd.S)  // the last column any thread owns: the pitch's padding is its to clear
                for (int k = c3 + 1; k < gd.w3p; ++k) d3[k - c3] = (half_t)0;
            }
          }
        } else if (hf == 0) {
          half_t* d2 = a.lv[2] + (p1g * (a.h >> 2) + rg) * (a.w >> 2) + (x0 >> 2) + tile * 2;
          d2[0] = l2[0];
          d2[1] = l2[1];
          if (a.nlev > 3 && (rg & 1))
            a.lv[3][(p1g * (a.h >> 3) + (rg >> 1)) * (a.w >> 3) + (x0 >> 3) + tile] = pool4(l2prev[0], l2prev[1], l2[0], l2[1]);
        }
        l2prev[0] = l2[0];
        l2prev[1] = l2[1];
      }
      l1prev = l1;
      return;
    }
    // level 0 + level 1: the thread reads columns 8s..8s+7 of both chunk rows once; its two 16-byte stores are parts
    // of two fully covered 128-byte row segments (8 lanes per source pixel and row)
    const half8 r0 = *reinterpret_cast<const half8*>(stage + sp1 * PB_PS + sseg * 8);
    const half8 r1 = *reinterpret_cast<const half8*>(stage + sp1 * PB_PS + 64 + sseg * 8);
    half_t* dst = a.lv[0] + (p1g * a.h + y) * a.w + x0 + sseg * 8;
    *reinterpret_cast<half8*>(dst) = r0;
    *reinterpret_cast<half8*>(dst + a.w) = r1;
    if (a.nlev <= 1) return;
    half4 l1;
#pragma unroll
    for (int k = 0; k < 4; ++k) l1[k] = pool4(r0[2 * k], r0[2 * k + 1], r1[2 * k], r1[2 * k + 1]);
    *reinterpret_cast<half4*>(a.lv[1] + (p1g * (a.h >> 1) + (y >> 1)) * (a.w >> 1) + (x0 >> 1) + sseg * 4) = l1;
    if (a.nlev > 2 && (ch & 1)) {
      half_t l2[2];
      l2[0] = pool4(l1prev[0], l1prev[1], l1[0], l1[1]);
      l2[1] = pool4(l1prev[2], l1prev[3], l1[2], l1[3]);
      half_t* d2 = a.lv[2] + (p1g * (a.h >> 2) + (y >> 2)) * (a.w >> 2) + (x0 >> 2) + sseg * 2;
      d2[0] = l2[0];
      d2[1] = l2[1];
      if (a.nlev > 3 && (ch & 3) == 3)
        a.lv[3][(p1g * (a.h >> 3) + (y >> 3)) * (a.w >> 3) + (x0 >> 3) + sseg] = pool4(l2prev[0], l2prev[1], l2[0], l2[1]);
      l2prev[0] = l2[0];
      l2prev[1] = l2[1];
    }
    l1prev = l1;
  };
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();  // every wave is done reading imgA (previous chunk's fragments, or the f1 tile)
    commit();
    __syncthreads();  // imgA ready; the previous chunk's staged tile is complete
    if (ch > 0) output(ch - 1, stage0 + ((ch - 1) & 1) * (PB_STAGE / 2));
    if (ch + 1 < nchunks) fetch(ch + 1);
    float4v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      half8 af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = tr_frag(imgA, a_off + kk * 32 * PB_PA + i * 32, 16 * PB_PA);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j][kk], acc[i][j], 0, 0, 0);
    }
    // D: rows = target pixels wn*32 + i*16 + 4g + r, column = source pixel wp*32 + j*16 + l16
    half_t* stage = stage0 + (ch & 1) * (PB_STAGE / 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        half4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = (half_t)(acc[i][j][r] * 0.0625f);
        *reinterpret_cast<half4*>(stage + (wp * 32 + j * 16 + l16) * PB_PS + wn * 32 + i * 16 + 4 * g) = hv;
      }
  }
  __syncthreads();
  output(nchunks - 1, stage0 + ((nchunks - 1) & 1) * (PB_STAGE / 2));
}

}  // namespace

// ---- plain kernels for what the MFMA kernel does not take (fp32 / fp64 maps, channel counts other than 128): the
// reference's `CorrBlock.corr` is dtype-generic (droid_net.py:94-102).  Off the update iteration's path (the SLAM maps are
// fp16 x 128 channels); kept simple: 16 x 16 output tile per workgroup, operands staged through LDS 16 channels at a time.
template <typename T, typename ACC>
__global__ __launch_bounds__(256) void corr_volume_plain_kernel(const T* __restrict__ f1, const T* __restrict__ f2,
                                                                T* __restrict__ vol, int C, int P) {
  __shared__ ACC t1[16][17], t2[16][17];
  const int e = blockIdx.z, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int p1 = blockIdx.y * 16 + ty, p2 = blockIdx.x * 16 + tx;
  const T* a = f1 + (int64_t)e * C * P;
  const T* b = f2 + (int64_t)e * C * P;
  ACC acc = 0;
  for (int c0 = 0; c0 < C; c0 += 16) {
    // t1[c][p1 of the tile], t2[c][p2 of the tile]; the reference divides both maps by 4 before the product
    const int c = c0 + ty, q1 = blockIdx.y * 16 + tx, q2 = blockIdx.x * 16 + tx;
    t1[ty][tx] = (c < C && q1 < P) ? (ACC)(T)((ACC)a[(int64_t)c * P + q1] / (ACC)4) : (ACC)0;
    t2[ty][tx] = (c < C && q2 < P) ? (ACC)(T)((ACC)b[(int64_t)c * P + q2] / (ACC)4) : (ACC)0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc += t1[k][ty] * t2[k][tx];
    __syncthreads();
  }
  if (p1 < P && p2 < P) vol[((int64_t)e * P + p1) * P + p2] = (T)acc;
}

// avg_pool2d(x, 2, stride = 2) over the last two dims of [n, h, w] (floor), at::native arithmetic: sum in window order in
// the accumulation type, divided by 4, rounded to T
template <typename T, typename ACC>
__global__ __launch_bounds__(256) void avg_pool2x2_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, int h, int w) {
  const int ho = h >> 1, wo = w >> 1;
  const int64_t total = n * ho * wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xo = (int)(i % wo), yo = (int)((i / wo) % ho);
    const T* s = x + ((i / wo / ho) * h + 2 * yo) * (int64_t)w + 2 * xo;
    y[i] = (T)((((ACC)s[0] + (ACC)s[1]) + (ACC)s[w] + (ACC)s[w + 1]) / (ACC)4);
  }
}

static int build_impl(const void* d_f1, const void* d_f2, const int64_t* d_idx1, const int64_t* d_idx2, const int* d_slots,
                      void* const* h_levels, int B, int C, int h, int w, int num_levels, int layout, bool prepared,
                      int64_t base, int64_t n_prep, void* stream) {
  VIPE_CHECK_ARG(h_levels && num_levels >= 1 && num_levels <= 4 && B >= 0 && C > 0 && h > 0 && w > 0);
  VIPE_CHECK_ARG(layout == VIPE_PYRAMID_REFERENCE || layout == VIPE_PYRAMID_BLOCKED);
  if (B == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_f1 && d_f2);
  for (int i = 0; i < num_levels; ++i) VIPE_CHECK_ARG(h_levels[i]);
  // tilings of this kernel: 128 channels; straight from the [C][h*w] maps for 64-column segments and 8-row groups (the
  // DROID feature map at 1/8 of 512 x 384 and multiples), from the prepared operand images (vipe_corr_prep) for any grid
  if (C != PB_C) return VIPE_EUNSUPPORTED;
  if (prepared) {
    if (layout != VIPE_PYRAMID_BLOCKED || (h >> (num_levels - 1)) < 1 || (w >> (num_levels - 1)) < 1) return VIPE_EUNSUPPORTED;
  } else if (w % 64 != 0 || h % 8 != 0) {
    return VIPE_EUNSUPPORTED;
  }
  const GenDims gd(h, w);
  const int64_t nwg = prepared ? (int64_t)B * gd.G : (int64_t)B * h * w / PB_M;
  if (nwg > 0x7fffffff) return VIPE_EINVAL;
  BuildArgs a;
  a.f1 = (const half_t*)d_f1;
  a.f2 = (const half_t*)d_f2;
  a.idx1 = d_idx1; a.idx2 = d_idx2; a.slots = d_slots;
  for (int i = 0; i < 4; ++i) a.lv[i] = i < num_levels ? (half_t*)h_levels[i] : nullptr;
  a.B = B; a.h = h; a.w = w; a.nlev = num_levels; a.base = base; a.n_prep = n_prep;
  static std::atomic<uint64_t> attr{0};  // bit d: set on device d
  vipe_once_per_device(attr, [] {
    (void)hipFuncSetAttribute((const void*)corr_pyramid_build_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS);
    (void)hipFuncSetAttribute((const void*)corr_pyramid_build_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS);
    (void)hipFuncSetAttribute((const void*)corr_pyramid_build_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS);
  });
  const dim3 grid((unsigned)nwg);
  if (prepared) corr_pyramid_build_kernel<true, true><<<grid, 512, PB_LDS, as_stream(stream)>>>(a);
  else if (layout == VIPE_PYRAMID_BLOCKED) corr_pyramid_build_kernel<true><<<grid, 512, PB_LDS, as_stream(stream)>>>(a);
  else corr_pyramid_build_kernel<false><<<grid, 512, PB_LDS, as_stream(stream)>>>(a);
  return vipe_launch_status();
}

extern "C" {

VIPE_EXPORT int vipe_corr_pyramid_build(const void* d_fmap1, const void* d_fmap2, void* const* h_levels, int B, int C,
                                        int h, int w, int num_levels, void* stream) {
  return build_impl(d_fmap1, d_fmap2, nullptr, nullptr, nullptr, h_levels, B, C, h, w, num_levels, VIPE_PYRAMID_REFERENCE,
                    false, 0, 0, stream);
}

VIPE_EXPORT int vipe_corr_pyramid_build_indexed(const void* d_fmaps, const int64_t* d_idx1, const int64_t* d_idx2,
                                                const int* d_slots, void* const* h_levels, int B, int C, int h, int w,
                                                int num_levels, int layout, void* stream) {
  VIPE_CHECK_ARG(B == 0 || (d_idx1 && d_idx2));
  return build_impl(d_fmaps, d_fmaps, d_idx1, d_idx2, d_slots, h_levels, B, C, h, w, num_levels, layout, false, 0, 0, stream);
}

VIPE_EXPORT int vipe_corr_blocked_dims(int h, int w, int* dims6) {
  VIPE_CHECK_ARG(h > 0 && w > 0 && dims6);
  const GenDims d(h, w);
  dims6[0] = d.G; dims6[1] = d.S; dims6[2] = d.R; dims6[3] = d.w2p; dims6[4] = d.w3p;
  dims6[5] = (w % 64 == 0 && h % 8 == 0) ? 1 : 0;  // 1: vipe_corr_pyramid_build_indexed tiles this grid without vipe_corr_prep
  return VIPE_OK;
}

VIPE_EXPORT int64_t vipe_corr_prep_halves(int C, int h, int w) {
  if (C != PB_C || h <= 0 || w <= 0) return 0;
  return GenDims(h, w).fstride;
}

VIPE_EXPORT int vipe_corr_prep(const void* d_fmaps, void* d_prep, int n, int C, int h, int w, void* stream) {
  VIPE_CHECK_ARG(n >= 0 && h > 0 && w > 0);
  if (C != PB_C) return VIPE_EUNSUPPORTED;
  if (n == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_fmaps && d_prep);
  const int64_t threads = GenDims(h, w).fstride / 8 * n;
  VIPE_CHECK_ARG((threads + 255) / 256 <= 0x7fffffff);
  corr_prep_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, as_stream(stream)>>>((const half_t*)d_fmaps, (half_t*)d_prep, n, h, w);
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_corr_pyramid_build_prepared(const void* d_prep, int64_t frame_base, int64_t n_prepared, const int64_t* d_idx1,
                                                 const int64_t* d_idx2, const int* d_slots, void* const* h_levels, int B,
                                                 int C, int h, int w, int num_levels, void* stream) {
  VIPE_CHECK_ARG((B == 0 || (d_idx1 && d_idx2)) && n_prepared >= 0);
  return build_impl(d_prep, d_prep, d_idx1, d_idx2, d_slots, h_levels, B, C, h, w, num_levels, VIPE_PYRAMID_BLOCKED, true,
                    frame_base, n_prepared, stream);
}

VIPE_EXPORT int vipe_corr_volume(const void* d_fmap1, const void* d_fmap2, void* d_volume, int B, int C, int P, int dtype,
                                 void* stream) {
  VIPE_CHECK_ARG(B >= 0 && C > 0 && P > 0 && B <= 65535);
  if (B == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_fmap1 && d_fmap2 && d_volume);
  const dim3 grid((P + 15) / 16, (P + 15) / 16, B);
  VIPE_CHECK_ARG(grid.y <= 65535);
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case VIPE_F16: corr_volume_plain_kernel<half_t, float><<<grid, 256, 0, s>>>((const half_t*)d_fmap1, (const half_t*)d_fmap2, (half_t*)d_volume, C, P); break;
    case VIPE_F32: corr_volume_plain_kernel<float, float><<<grid, 256, 0, s>>>((const float*)d_fmap1, (const float*)d_fmap2, (float*)d_volume, C, P); break;
    case VIPE_F64: corr_volume_plain_kernel<double, double><<<grid, 256, 0, s>>>((const double*)d_fmap1, (const double*)d_fmap2, (double*)d_volume, C, P); break;
    default: return VIPE_EINVAL;
  }
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_avg_pool2x2(const void* d_x, void* d_y, int64_t n, int h, int w, int dtype, void* stream) {
  VIPE_CHECK_ARG(n >= 0 && h >= 2 && w >= 2);
  if (n == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_x && d_y);
  const int64_t total = n * (h >> 1) * (w >> 1);
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20);
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case VIPE_F16: avg_pool2x2_kernel<half_t, float><<<blocks, 256, 0, s>>>((const half_t*)d_x, (half_t*)d_y, n, h, w); break;
    case VIPE_F32: avg_pool2x2_kernel<float, float><<<blocks, 256, 0, s>>>((const float*)d_x, (float*)d_y, n, h, w); break;
    case VIPE_F64: avg_pool2x2_kernel<double, double><<<blocks, 256, 0, s>>>((const double*)d_x, (double*)d_y, n, h, w); break;
    default: return VIPE_EINVAL;
  }
  return vipe_launch_status();
}

}  // extern "C"
