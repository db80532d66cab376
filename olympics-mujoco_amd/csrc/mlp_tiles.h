// Shared pieces of the fused actor / critic MLP forward (K11, csrc/k11_mlp.hip) that the persistent rollout
// kernel (K13, csrc/k13_rollout.hip) runs inside its step loop: the packed-weight layout and the 32-row x
// 32-column f32 MFMA tile.  Numerics: see k11_mlp.hip (exact f32 fma chains, k ascending).
#pragma once
#include "oly_common.h"

namespace oly_mlp {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int HID = 256;        // hidden width (both layers)
constexpr int RT = 32;          // rows per workgroup
constexpr int LDP = 33;         // LDS row pitch of the [k][row] activation images (conflict-free)
constexpr int MAX_IN = 64;
constexpr int MAX_OUT = 32;
constexpr int THREADS = 512;     // 8 waves
constexpr int KSPLIT = 8;        // output layer: k split over the waves
constexpr int G1 = 8;            // layer 1: k zero-padded to MAX_IN = 8 groups of four k-steps

struct PackLayout {
  int in_dim, out_dim, g1;      // g1: groups of four k-steps in layer 1
  size_t w1, b1, w2, b2, w3, b3, mean, std, total;
};

__host__ __device__ inline PackLayout pack_layout(int in_dim, int out_dim) {
  PackLayout L;
  L.in_dim = in_dim;
  L.out_dim = out_dim;
  L.g1 = G1;
  L.w1 = 0;
  L.b1 = L.w1 + (size_t)8 * L.g1 * 256;
  L.w2 = L.b1 + HID;
  L.b2 = L.w2 + (size_t)8 * 32 * 256;
  L.w3 = L.b2 + HID;
  L.b3 = L.w3 + (size_t)32 * 256;
  L.mean = L.b3 + MAX_OUT;
  L.std = L.mean + MAX_IN;
  L.total = L.std + MAX_IN;
  return L;
}

// one 32-row x 32-column tile of  A W  (k = 0 .. 8 G - 1 in order): A fragments from the [k][row] LDS
// image (one group ahead), W from the packed stream (two groups ahead), fully unrolled so that every
// load is in flight behind the 64-cycle MFMAs of the groups before it.
template <int G>
__device__ __forceinline__ void layer_tile(const float* __restrict__ aT, const float4* __restrict__ w, int lane,
                                           f32x16& acc) {
  const int r = lane & 31, h = lane >> 5;
  float4 b[3];
  float a[2][4];
  b[0] = w[lane];
  if (G > 1) b[1] = w[64 + lane];
  {
    const float* ap = aT + (size_t)h * LDP + r;
    a[0][0] = ap[0]; a[0][1] = ap[2 * LDP]; a[0][2] = ap[4 * LDP]; a[0][3] = ap[6 * LDP];
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g + 2 < G) b[(g + 2) % 3] = w[(size_t)(g + 2) * 64 + lane];
    if (g + 1 < G) {
      const float* ap = aT + (size_t)(8 * (g + 1) + h) * LDP + r;
      a[(g + 1) & 1][0] = ap[0]; a[(g + 1) & 1][1] = ap[2 * LDP];
      a[(g + 1) & 1][2] = ap[4 * LDP]; a[(g + 1) & 1][3] = ap[6 * LDP];
    }
    // keep the loads above ahead of this group's MFMAs (hipcc otherwise sinks each load to just before
    // its first use and waits for it there)
    __builtin_amdgcn_sched_barrier(0);
    const float4 bb = b[g % 3];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][0], bb.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][1], bb.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][2], bb.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][3], bb.w, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// bias + ReLU of an accumulator tile into the [k][row] image of the next layer
__device__ __forceinline__ void store_relu(const f32x16& acc, const float* __restrict__ bias, int col0, int lane,
                                           float* __restrict__ hT) {
  const int col = col0 + (lane & 31), h = lane >> 5;
  const float b = bias[col];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    const float v = acc[i] + b;
    hT[(size_t)col * LDP + row] = (v > 0.f || v != v) ? v : 0.f;     // relu, NaN kept like torch
  }
}

}  // namespace oly_mlp
