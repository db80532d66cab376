// Is sincos(x) bit-identical to (sin(x), cos(x)) on gfx950's device libm?  (K10 evaluates both values of
// several arguments; one call instead of two would remove two of the nine function-class evaluations.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
__global__ void k(const double* x, long n, unsigned long long* bad_s, unsigned long long* bad_c) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  sincos(x[i], &s, &c);
  if (__double_as_longlong(s) != __double_as_longlong(sin(x[i]))) atomicAdd(bad_s, 1ull);
  if (__double_as_longlong(c) != __double_as_longlong(cos(x[i]))) atomicAdd(bad_c, 1ull);
}
int main() {
  const long n = 1 << 24;
  double* h = (double*)malloc(n * sizeof(double));
  srand(1);
  for (long i = 0; i < n; ++i) {
    double u = (double)rand() / RAND_MAX;
    double scale = (i % 4 == 0) ? 1e-3 : (i % 4 == 1) ? 3.2 : (i % 4 == 2) ? 60.0 : 1e6;
    h[i] = (2 * u - 1) * scale;
  }
  double* d; unsigned long long *bs, *bc, hs = 0, hc = 0;
  hipMalloc(&d, n * sizeof(double)); hipMalloc(&bs, 8); hipMalloc(&bc, 8);
  hipMemcpy(d, h, n * sizeof(double), hipMemcpyHostToDevice);
  hipMemset(bs, 0, 8); hipMemset(bc, 0, 8);
  k<<<(n + 255) / 256, 256>>>(d, n, bs, bc);
  hipMemcpy(&hs, bs, 8, hipMemcpyDeviceToHost); hipMemcpy(&hc, bc, 8, hipMemcpyDeviceToHost);
  printf("n=%ld sin mismatches=%llu cos mismatches=%llu\n", n, hs, hc);
  return 0;
}
