// Per-edge state of a factor graph whose edges come and go every keyframe (factor_graph.py:147-202 of the reference):
// `add_factors` concatenates - i.e. copies - every per-edge tensor, `rm_factors` compacts each of them with a boolean
// mask (a device-to-host sync per tensor).  Here the tensors live in stores with spare capacity and the two operations
// are ONE launch each:
//   vipe_rows_gather           compaction: the surviving rows of up to 8 tensors move into their other backing buffers
//   vipe_gather_nchw_to_nhwc   append: hidden state / context features of the new edges' source frames, read from the
//                              keyframe buffer ([N, C, h*w], buffer.py:142-170) and written channels-last into the tail
//                              of the stores (what `nets[ii].permute(...)` + `torch.cat` do in four launches per tensor)
#include "common.cuh"

namespace {

constexpr int MAX_JOBS = 8;

struct RowsArgs {
  vipe_rows_job job[MAX_JOBS];
  int64_t chunks[MAX_JOBS];  // copy units of job j
  int unit[MAX_JOBS];        // bytes per unit: 16, 8 or 4
};

typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

template <typename U>
__device__ __forceinline__ void rows_gather_job(const vipe_rows_job& j, int64_t total) {
  constexpr int SH = sizeof(U) == 16 ? 4 : (sizeof(U) == 8 ? 3 : 2);
  const int64_t per_seg = j.seg_bytes >> SH, per_row = per_seg * j.n_seg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / per_row, w = i % per_row;
    const int64_t seg = w / per_seg, c = w % per_seg;
    const int64_t sr = j.idx ? j.idx[r] : r;
    const char* src = (const char*)j.src + sr * j.src_row_pitch + seg * j.seg_pitch + (c << SH);
    char* dst = (char*)j.dst + (r + j.dst_row0) * j.dst_row_pitch + seg * j.seg_pitch + (c << SH);
    *reinterpret_cast<U*>(dst) = *reinterpret_cast<const U*>(src);
  }
}

// grid (x, n_jobs): grid-stride over the chunks of job blockIdx.y (16-, 8- or 4-byte units, whatever the job's sizes,
// pitches and addresses are all multiples of: a [h, w, 2] f32 map of an odd pixel count is not a multiple of 16 bytes)
__global__ __launch_bounds__(256) void rows_gather_kernel(RowsArgs a) {
  const vipe_rows_job& j = a.job[blockIdx.y];
  const int u = a.unit[blockIdx.y];
  if (u == 16) rows_gather_job<uint4v>(j, a.chunks[blockIdx.y]);
  else if (u == 8) rows_gather_job<unsigned long long>(j, a.chunks[blockIdx.y]);
  else rows_gather_job<unsigned int>(j, a.chunks[blockIdx.y]);
}

struct NhwcArgs {
  vipe_nhwc_job job[MAX_JOBS];
  int n_rows, C, P;
};

// grid (ceil(P / 64), n_rows, n_jobs); block 256: a 64-pixel x C-channel tile through LDS (C <= 128)
__global__ __launch_bounds__(256) void gather_nchw_to_nhwc_kernel(NhwcArgs a) {
  __shared__ half_t tile[128][64 + 2];
  const vipe_nhwc_job& j = a.job[blockIdx.z];
  const int r = blockIdx.y, p0 = blockIdx.x * 64, tid = threadIdx.x;
  const int64_t f = j.frame ? j.frame[r] : r;
  const half_t* src = (const half_t*)j.src + f * (int64_t)a.C * a.P;
  // read: 64 consecutive pixels of one channel = 128 contiguous bytes; a wave reads one channel row
  for (int i = tid; i < a.C * 64; i += 256) {
    const int c = i >> 6, pp = i & 63;
    tile[c][pp] = (p0 + pp < a.P) ? src[(int64_t)c * a.P + p0 + pp] : (half_t)0;
  }
  __syncthreads();
  half_t* dst = (half_t*)j.dst + (int64_t)(r + j.dst_row0) * j.dst_row_pitch;
  // write: C consecutive channels of one pixel (2 C contiguous bytes)
  for (int i = tid; i < a.C * 64; i += 256) {
    const int pp = i / a.C, c = i % a.C;
    if (p0 + pp < a.P) dst[(int64_t)(p0 + pp) * j.dst_ctot + j.dst_coff + c] = tile[c][pp];
  }
}

}  // namespace

VIPE_EXPORT int vipe_rows_gather(const vipe_rows_job* jobs, int n_jobs, void* stream) {
  VIPE_CHECK_ARG(n_jobs >= 0 && n_jobs <= MAX_JOBS && (n_jobs == 0 || jobs));
  RowsArgs a;
  int live = 0;
  int64_t most = 0;
  for (int k = 0; k < n_jobs; ++k) {
    const vipe_rows_job& j = jobs[k];
    VIPE_CHECK_ARG(j.n_rows >= 0 && j.dst_row0 >= 0 && j.n_seg >= 1 && j.seg_bytes > 0);
    if (j.n_rows == 0) continue;
    VIPE_CHECK_ARG(j.src && j.dst && j.src != j.dst);
    const uint64_t all = (uint64_t)j.seg_bytes | (uint64_t)j.seg_pitch | (uint64_t)j.src_row_pitch | (uint64_t)j.dst_row_pitch |
                         (uint64_t)(uintptr_t)j.src | (uint64_t)(uintptr_t)j.dst;
    VIPE_CHECK_ARG((all & 3) == 0);
    const int unit = (all & 15) == 0 ? 16 : ((all & 7) == 0 ? 8 : 4);
    a.job[live] = j;
    a.unit[live] = unit;
    a.chunks[live] = (int64_t)j.n_rows * j.n_seg * (j.seg_bytes / unit);
    most = a.chunks[live] > most ? a.chunks[live] : most;
    ++live;
  }
  if (live == 0) return VIPE_OK;
  const int64_t blocks = (most + 255) / 256;
  const dim3 grid((unsigned)(blocks < 4096 ? blocks : 4096), live);
  rows_gather_kernel<<<grid, 256, 0, as_stream(stream)>>>(a);
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_gather_nchw_to_nhwc_f16(const vipe_nhwc_job* jobs, int n_jobs, int n_rows, int C, int P, void* stream) {
  VIPE_CHECK_ARG(n_jobs >= 0 && n_jobs <= MAX_JOBS && n_rows >= 0 && C > 0 && P > 0 && (n_jobs == 0 || jobs));
  if (C > 128) return VIPE_EUNSUPPORTED;
  if (n_jobs == 0 || n_rows == 0) return VIPE_OK;
  NhwcArgs a;
  for (int k = 0; k < n_jobs; ++k) {
    VIPE_CHECK_ARG(jobs[k].src && jobs[k].dst && jobs[k].dst_ctot >= C && jobs[k].dst_coff >= 0 &&
                   jobs[k].dst_coff + C <= jobs[k].dst_ctot && jobs[k].dst_row0 >= 0);
    a.job[k] = jobs[k];
  }
  a.n_rows = n_rows; a.C = C; a.P = P;
  VIPE_CHECK_ARG(n_rows <= 65535);
  gather_nchw_to_nhwc_kernel<<<dim3((P + 63) / 64, n_rows, n_jobs), 256, 0, as_stream(stream)>>>(a);
  return vipe_launch_status();
}
