// What a 16-row layer costs in K14's shape: 8 waves (two per SIMD), every wave NT dependent accumulator chains over G
// groups of four k-steps (128 MFMAs per wave and layer either way), per group one 16-byte LDS read (A fragments) and NT
// 16-byte buffer loads (the weight stream, L2-resident, two groups ahead), a workgroup barrier between layers.
// Variants switch the pieces off / move the loads of waves 4-7 to the middle of a group.  Prints shader-clock cycles
// per layer (issue floor: 2 waves x 128 MFMAs x 32 cycles = 8192).
//   hipcc --offload-arch=gfx950 -O3 -o tools/hip/mfma16_pairs.bin tools/hip/mfma16_pairs.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sel(const u32x4& v, int q) {
  return __uint_as_float(q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w);
}

// NT: chains per wave; LOADW: weight stream on; LOADA: LDS A fragments on; SHIFT: waves 4-7 issue their loads after the
// group's first half
template <int NT, bool LOADW, bool LOADA, bool SHIFT, int LAYERS = 1, int RT = 1>
__global__ __launch_bounds__(512, 2) void layer_kernel(const float* __restrict__ w, int reps, float* out, unsigned long long* ticks) {
  // LAYERS > 1: one long stream (the start-up round trip amortised); RT row tiles share every weight fragment (a 16 RT-row tile)
  constexpr int G = LAYERS * 128 / (4 * NT * RT);
  __shared__ float4 img[32 * 64];              // A fragments: the group index wraps
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) img[i] = make_float4(1.f, 0.5f, 0.25f, 0.125f);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 8 * 4 * 128 * 1024, 0x00020000);
  const unsigned voff = lane * 16;
  unsigned so[NT];
  for (int t = 0; t < NT; ++t) so[t] = (unsigned)((wave * NT + t) * G) * 1024u;
  f32x4 acc[RT][NT];
  u32x4 b[3][NT];
  float4 a[2][RT];
  const bool late = SHIFT && wave >= 4;
  auto loads = [&](int g) {
    if (LOADW && g + 2 < G) {
#pragma unroll
      for (int t = 0; t < NT; ++t) b[(g + 2) % 3][t] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so[t] + (unsigned)(g + 2) * 1024u, 0);
    }
    if (LOADA && g + 1 < G)
#pragma unroll
      for (int r = 0; r < RT; ++r) a[(g + 1) & 1][r] = img[(((g + 1) * RT + r) & 31) * 64 + lane];
  };
  float keep = 0.f;
  unsigned long long t0 = 0;
  for (int r = 0; r < reps + 1; ++r) {
    if (r == 1) t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (LOADW) b[d][t] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so[t] + (unsigned)d * 1024u, 0);
        else b[d][t] = u32x4{0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};
      }
    if (!LOADW)
#pragma unroll
      for (int t = 0; t < NT; ++t) b[2][t] = u32x4{0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      a[0][r] = LOADA ? img[r * 64 + lane] : make_float4(1.f, 1.f + r, 1.f, 1.f);
      a[1][r] = a[0][r];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (!late) loads(g);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          const float4 av = a[g & 1][r];
          const float aq = q == 0 ? av.x : q == 1 ? av.y : q == 2 ? av.z : av.w;
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, sel(b[g % 3][t], q), acc[r][t], 0, 0, 0);
        }
        if (q == 1) {
          __builtin_amdgcn_sched_barrier(0);
          if (late) loads(g);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < RT; ++r) keep += acc[r][t][0] + acc[r][t][3];
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = keep;
  if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NT, bool LOADW, bool LOADA, bool SHIFT, int LAYERS = 1, int RT = 1>
void run(const char* name, const float* w, float* out, unsigned long long* ticks) {
  const int reps = 200;
  for (int it = 0; it < 2; ++it) {
    hipLaunchKernelGGL((layer_kernel<NT, LOADW, LOADA, SHIFT, LAYERS, RT>), dim3(256), dim3(512), 0, 0, w, reps, out, ticks);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(256 * 8);
  hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double sum = 0, mx = 0;
  for (auto v : h) { sum += (double)v; if ((double)v > mx) mx = (double)v; }
  printf("{\"variant\": \"%s\", \"chains_per_wave\": %d, \"layers_per_stream\": %d, \"row_tiles\": %d, \"cycles_per_layer_mean\": %.0f, \"max\": %.0f, \"floor\": 8192}\n", name, NT,
         LAYERS, RT, sum / h.size() / reps / LAYERS, mx / reps / LAYERS);
}

int main() {
  float* w;
  float* out;
  unsigned long long* ticks;
  hipMalloc(&w, 8 * 4 * 128 * 1024);
  hipMemset(w, 0, 8 * 4 * 128 * 1024);
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&ticks, 256 * 8 * sizeof(unsigned long long));
  run<2, false, false, false>("mfma only", w, out, ticks);
  run<2, false, true, false>("+ LDS A fragments", w, out, ticks);
  run<2, true, false, false>("+ weight stream", w, out, ticks);
  run<2, true, true, false>("both (K14's layer)", w, out, ticks);
  run<2, true, true, true>("both, waves 4-7 load mid-group", w, out, ticks);
  run<4, false, false, false>("mfma only", w, out, ticks);
  run<4, true, true, false>("both", w, out, ticks);
  run<4, true, true, true>("both, waves 4-7 load mid-group", w, out, ticks);
  run<2, true, true, false, 4>("both, one stream of four layers", w, out, ticks);
  run<2, true, true, true, 4>("both, one stream of four layers, waves 4-7 load mid-group", w, out, ticks);
  run<2, true, false, false, 4>("weight stream only, four layers", w, out, ticks);
  run<4, true, true, false, 4>("both, one stream of four layers", w, out, ticks);
  run<2, true, true, false, 4, 2>("both, four layers, two row tiles per weight fragment", w, out, ticks);
  run<2, true, true, false, 1, 2>("both, two row tiles per weight fragment", w, out, ticks);
  return 0;
}
