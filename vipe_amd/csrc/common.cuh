// Shared helpers for the gfx950 kernels of libvipe_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

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

// Runs `fn` exactly once per (call site, device) and returns only when it HAS run: hipFuncSetAttribute is a per-device
// setting (a process may drive several devices; one process per GPU is the deployment model, but the library must not
// depend on it), and a second host thread must not launch a kernel that needs > 64 KiB of dynamic LDS before the first
// one has raised the limit - the bit is published after `fn`, under a lock.
template <typename F>
static inline void vipe_once_per_device(std::atomic<uint64_t>& done, F&& fn) {
  int d = 0;
  (void)hipGetDevice(&d);
  const uint64_t bit = 1ull << (d & 63);
  if (done.load(std::memory_order_acquire) & bit) return;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (done.load(std::memory_order_relaxed) & bit) return;
  fn();
  done.fetch_or(bit, std::memory_order_release);
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

// Wave-wide float sum on the DPP path (one VALU instruction per step instead of a ds_bpermute round trip):
// xor-1 / xor-2 quad permutes, half-row and row mirrors give every lane its 16-lane row sum, row_bcast15/31 carry
// the row sums forward so that LANE 63 holds the total (other lanes hold partial sums).
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
#define VIPE_DPP_ADD(ctrl, rmask)                                                                          \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, true))
  VIPE_DPP_ADD(0xB1, 0xf);   // quad_perm [1,0,3,2]
  VIPE_DPP_ADD(0x4E, 0xf);   // quad_perm [2,3,0,1]
  VIPE_DPP_ADD(0x141, 0xf);  // row_half_mirror
  VIPE_DPP_ADD(0x140, 0xf);  // row_mirror
  VIPE_DPP_ADD(0x142, 0xa);  // row_bcast15 -> rows 1, 3
  VIPE_DPP_ADD(0x143, 0xc);  // row_bcast31 -> rows 2, 3
#undef VIPE_DPP_ADD
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
