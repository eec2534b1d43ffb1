// experiment: a stream whose queue may only use a subset of the CUs (hipExtStreamCreateWithCUMask)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
extern "C" int sm_create(int n_cus, const uint32_t* mask_words, int n_words, void** out) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, n_words, mask_words);
  if (e != hipSuccess) { fprintf(stderr, "cu mask stream: %s\n", hipGetErrorString(e)); return 1; }
  *out = (void*)s;
  return 0;
}
__global__ void where_kernel(unsigned* out) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  // stay resident a little so that the grid spreads over every CU the queue may use
  unsigned long long t0 = clock64();
  while (clock64() - t0 < 200000) {}
}
extern "C" int sm_where(void* stream, unsigned* d_out, int n_blocks) {
  where_kernel<<<n_blocks, 64, 65536, (hipStream_t)stream>>>(d_out);
  return (int)hipGetLastError();
}
