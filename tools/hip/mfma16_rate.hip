// Issue rate of v_mfma_f32_16x16x4_f32 in the shape K13's forward waves use it: four independent accumulators, per
// group of 16 MFMAs one 16-byte LDS read (A fragments) and four 16-byte global loads (the weight stream, L2-resident),
// three groups ahead.  Variants: MFMAs only / + LDS reads / + LDS reads and the weight stream; one or two waves per
// SIMD.  Prints s_memtime ticks per MFMA (the counter runs at the shader clock).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma16_rate tools/hip/mfma16_rate.hip && /tmp/mfma16_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>     // 0: MFMAs only; 1: + LDS A fragments; 2: + weight stream
__global__ __launch_bounds__(512) void rate_kernel(const float4* __restrict__ w, int groups, int reps, float* out,
                                                   unsigned long long* ticks) {
  __shared__ float4 img[16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 64; i += blockDim.x) img[i] = make_float4(1.f, 0.5f, 0.25f, 0.125f);
  __syncthreads();
  f32x4 acc[4] = {{0}, {0}, {0}, {0}};
  float4 b[4][4];
  float4 a[2];
  const float4* wp[4];
  for (int t = 0; t < 4; ++t) wp[t] = w + ((size_t)(wave * 4 + t) * groups) * 64;
  for (int d = 0; d < 4; ++d)
    for (int t = 0; t < 4; ++t) b[d][t] = make_float4(1.f, 1.f, 1.f, 1.f);
  a[0] = a[1] = make_float4(1.f, 1.f, 1.f, 1.f);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (MODE >= 2)
      for (int d = 0; d < 3; ++d)
        for (int t = 0; t < 4; ++t) b[d][t] = wp[t][(size_t)d * 64 + lane];
    if (MODE >= 1) a[0] = img[lane];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if (MODE >= 2 && g + 3 < 16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) b[(g + 3) % 4][t] = wp[t][(size_t)(g + 3) * 64 + lane];
      }
      if (MODE >= 1 && g + 1 < 16) a[(g + 1) & 1] = img[(g + 1) * 64 + lane];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float aq = q == 0 ? a[g & 1].x : q == 1 ? a[g & 1].y : q == 2 ? a[g & 1].z : a[g & 1].w;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float4 bb = b[g % 4][t];
          const float bq = q == 0 ? bb.x : q == 1 ? bb.y : q == 2 ? bb.z : bb.w;
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bq, acc[t], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) ticks[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

int main() {
  const int groups = 16, reps = 400;
  float4* w;
  float* out;
  unsigned long long* ticks;
  const size_t wn = (size_t)8 * 4 * groups * 64;
  hipMalloc(&w, wn * sizeof(float4));
  hipMemset(w, 0, wn * sizeof(float4));
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&ticks, 256 * 8 * sizeof(unsigned long long));
  for (int threads : {256, 512}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int it = 0; it < 2; ++it) {
        if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(256), dim3(threads), 0, 0, w, groups, reps, out, ticks);
        if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(256), dim3(threads), 0, 0, w, groups, reps, out, ticks);
        if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(256), dim3(threads), 0, 0, w, groups, reps, out, ticks);
        hipDeviceSynchronize();
      }
      std::vector<unsigned long long> h(256 * 8);
      hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      const int nw = 256 * threads / 64;
      double sum = 0, mx = 0;
      for (int i = 0; i < nw; ++i) { sum += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; }
      const double per = sum / nw / ((double)reps * groups * 16);
      printf("{\"waves_per_simd\": %d, \"mode\": %d, \"ticks_per_mfma_per_wave\": %.2f, \"ticks_per_mfma_per_simd\": %.2f, \"max_over_mean\": %.3f}\n",
             threads / 256, mode, per, per / (threads / 256), mx / (sum / nw));
    }
  }
  return 0;
}
