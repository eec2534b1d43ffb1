// precision of v_rsq_f64 on gfx950 and of one / two Newton steps on top of it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r = __builtin_amdgcn_rsq(v);
  r0[i] = r;
  const double hx = 0.5 * v;
  r = r * __builtin_fma(-hx * r, r, 1.5);
  r1[i] = r;
  r = r * __builtin_fma(-hx * r, r, 1.5);
  r2[i] = r;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), a(n), b(n), c(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(-30.0, 30.0);
  for (auto& v : x) v = std::exp2(u(g)) * (1.0 + 0.999 * (double)(g() >> 11) / 9007199254740992.0);
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / sqrtl((long double)x[i]);
    e0 = fmax(e0, (double)fabsl((a[i] - t) / t));
    e1 = fmax(e1, (double)fabsl((b[i] - t) / t));
    e2 = fmax(e2, (double)fabsl((c[i] - t) / t));
  }
  printf("max rel err: v_rsq_f64 %.3e (%.1f bits), +1 Newton %.3e (%.1f bits), +2 Newton %.3e (%.1f bits)\n", e0, -log2(e0), e1,
         -log2(e1), e2, -log2(e2));
  return 0;
}
