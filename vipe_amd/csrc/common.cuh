// Shared helpers for the gfx950 kernels of libvipe_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/vipe_amd.h"

#define VIPE_EXPORT extern "C" __attribute__((visibility("default")))

#define VIPE_CHECK_ARG(cond) \
  do {                       \
    if (!(cond)) return VIPE_EINVAL; \
  } while (0)

static inline int vipe_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VIPE_OK : (int)e;
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

constexpr int WAVE = 64;

// ---- wave-level reductions (64 lanes). DPP row/bank shuffles are emitted by the compiler for the
// constant-offset __shfl_xor pattern.
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// half <-> float with round-to-nearest-even, the arithmetic c10::Half performs (float op, round to half)
using half_t = _Float16;
__device__ __forceinline__ float h2f(half_t h) { return (float)h; }
__device__ __forceinline__ half_t f2h(float f) { return (half_t)f; }

// XCD-aware block remap (cdna_hip_programming.md T1, bijective form): consecutive logical blocks
// land on the same XCD so neighbouring tiles share its L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int q = nwg / nx, r = nwg % nx;
  int xcd = bid % nx;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + bid / nx;
}
