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

// The packed stream of one network holds every weight matrix TWICE: as the B-operand stream of
// v_mfma_f32_32x32x2_f32 (32-row tiles: K11) and as that of v_mfma_f32_16x16x4_f32 (16-row tiles: K13, which needs
// twice as many workgroups); biases and the input normalisation once.
constexpr int G1N = 4;           // 16-wide layout, layer 1: k zero-padded to MAX_IN = 4 groups of four k-steps (16 k each)
constexpr int T3N = MAX_OUT / 16;   // 16-wide layout, output layer: column tiles

struct PackLayout {
  int in_dim, out_dim, g1;      // g1: groups of four k-steps in layer 1
  size_t w1, b1, w2, b2, w3, b3, mean, std;
  size_t w1n, w2n, w3n;         // the 16-column-tile streams
  size_t w2t, w3t;              // transposed-role streams of the update kernel (K14): the B operand of dH = dZ W
  size_t total;
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
  L.w1n = L.std + MAX_IN;
  L.w2n = L.w1n + (size_t)(HID / 16) * G1N * 256;
  L.w3n = L.w2n + (size_t)(HID / 16) * (HID / 16) * 256;
  // K14 (csrc/k14_ppo_update.hip): the data gradients dH1 = dZ2 W2 and dH2 = dZ3 W3 sum over the layer's OUTPUT
  // index n, so the B operand is  PT[tile][group g][lane][q] = W[n = 16 g + 4 q + (lane >> 4)][k = 16 tile + (lane & 15)]
  // (W3: n >= out_dim zero; T3N groups of n)
  L.w2t = L.w3n + (size_t)T3N * (HID / 16) * 256;
  L.w3t = L.w2t + (size_t)(HID / 16) * (HID / 16) * 256;
  L.total = L.w3t + (size_t)(HID / 16) * T3N * 256;
  return L;
}

// ---------------------------------------------------------------------------------------------------------------
// 16-row tiles on v_mfma_f32_16x16x4_f32 (lane l: A[row l & 15][k = l >> 4], B[k = l >> 4][col l & 15]; C/D: col =
// l & 15, row = 4 (l >> 4) + register).  Same numerics as the 32-row form: every output is the f32 fma chain over k
// ascending (k = 16 g + 4 q + (l >> 4) for group g, k-step q).
//
// Packed weights:  P16[tile][group g][lane][q] = W[n = 16 tile + (lane & 15)][k = 16 g + 4 q + (lane >> 4)]
//                  (one 16-byte load per lane feeds four k-steps of one 16-column tile)
// Activations in LDS, as float4:  image[g * 64 + lane] = { act[k = 16 g + 4 q + (lane >> 4)][row = lane & 15] : q = 0..3 }
//                  i.e. element (k, row) lives at float index act16_index(k, row): the A fragments of a group are ONE
//                  conflict-free 16-byte read per lane.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int act16_index(int k, int row) {
  return (((k >> 4) * 4 + (k & 3)) * 16 + row) * 4 + ((k >> 2) & 3);
}

// NT 16-column tiles of  A W  that share the A operand, G groups (k = 0 .. 16 G - 1 in order): weights two groups
// ahead, activations one group ahead, fully unrolled; NT >= 2 independent accumulators cover the 40-cycle dependent
// latency of the 32-cycle instruction.
// PRE: the caller has requested the first two groups already (preload16, in front of the barrier that ends the layer
// before: weights do not depend on it, and behind it the first MFMA waits out an L2 round trip with nothing to hide it
// when a workgroup is alone on its CU).
template <int G, int NT>
__device__ __forceinline__ void preload16(const float4* const (&w)[NT], int lane, float4 (&b)[3][NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    b[0][t] = w[t][lane];
    if (G > 1) b[1][t] = w[t][64 + lane];
  }
}

template <int G, int NT, bool PRE = false>
__device__ __forceinline__ void layer_tiles16(const float4* __restrict__ a4, const float4* const (&w)[NT], int lane,
                                              f32x4 (&acc)[NT], float4 (&b)[3][NT]) {
  float4 a[2];
  if (!PRE) preload16<G, NT>(w, lane, b);
  a[0] = a4[lane];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g + 2 < G) {
#pragma unroll
      for (int t = 0; t < NT; ++t) b[(g + 2) % 3][t] = w[t][(size_t)(g + 2) * 64 + lane];
    }
    if (g + 1 < G) a[(g + 1) & 1] = a4[(g + 1) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);      // keep the loads above ahead of this group's MFMAs
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float aq = q == 0 ? a[g & 1].x : q == 1 ? a[g & 1].y : q == 2 ? a[g & 1].z : a[g & 1].w;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float bq = q == 0 ? b[g % 3][t].x : q == 1 ? b[g % 3][t].y : q == 2 ? b[g % 3][t].z : b[g % 3][t].w;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bq, acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int G, int NT>
__device__ __forceinline__ void layer_tiles16(const float4* __restrict__ a4, const float4* const (&w)[NT], int lane,
                                              f32x4 (&acc)[NT]) {
  float4 b[3][NT];
  layer_tiles16<G, NT, false>(a4, w, lane, acc, b);
}

// The same tiles for a wave that is alone on its SIMD (K13's forward waves): nothing else hides an L2 round trip
// (~1200 cycles), so the weight stream runs D groups ahead in a ring of D + 1 register sets that the CALLER owns and
// that is never drained between layers: while a layer's last D groups are multiplied, the NEXT layer's first groups
// (wn, NGN of them) are already requested (weights do not depend on the barrier between the layers).  RB is the ring
// slot of the layer's group 0; the next layer's is (RB + GT) % (D + 1).  The call runs groups [GB, GE) of the
// GT-group layer: a layer may be cut in two by a barrier that the wave does not itself need (accumulators and ring
// stay in registers).  Same MFMAs in the same order as layer_tiles16.
template <int NT, int D, int RB>
__device__ __forceinline__ void preload_tiles16(const float4* const (&w)[NT], int lane, int groups,
                                                float4 (&b)[D + 1][NT]) {
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < groups) {
#pragma unroll
      for (int t = 0; t < NT; ++t) b[(RB + d) % (D + 1)][t] = w[t][(size_t)d * 64 + lane];
    }
}

template <int GB, int GE, int GT, int NT, int D, int RB, int NGN>
__device__ __forceinline__ void layer_tiles16p(const float4* __restrict__ a4, const float4* const (&w)[NT],
                                               const float4* const (&wn)[NT], int lane, f32x4 (&acc)[NT],
                                               float4 (&b)[D + 1][NT]) {
  float4 a[2];
  a[GB & 1] = a4[GB * 64 + lane];
#pragma unroll
  for (int g = GB; g < GE; ++g) {
    if (g + D < GT) {
#pragma unroll
      for (int t = 0; t < NT; ++t) b[(RB + g + D) % (D + 1)][t] = w[t][(size_t)(g + D) * 64 + lane];
    } else if (g + D - GT < NGN) {
#pragma unroll
      for (int t = 0; t < NT; ++t) b[(RB + g + D) % (D + 1)][t] = wn[t][(size_t)(g + D - GT) * 64 + lane];
    }
    if (g + 1 < GE) a[(g + 1) & 1] = a4[(g + 1) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);      // keep the loads above ahead of this group's MFMAs
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float aq = q == 0 ? a[g & 1].x : q == 1 ? a[g & 1].y : q == 2 ? a[g & 1].z : a[g & 1].w;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 bb = b[(RB + g) % (D + 1)][t];
        const float bq = q == 0 ? bb.x : q == 1 ? bb.y : q == 2 ? bb.z : bb.w;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bq, acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// bias + ReLU of a 16 x 16 accumulator tile (columns 16 tile ..) into the activation image of the next layer
__device__ __forceinline__ void store_relu16(const f32x4& acc, const float* __restrict__ bias, int tile, int lane,
                                             float* __restrict__ img) {
  const int c = lane & 15, h2 = lane >> 4;
  const float bv = bias[16 * tile + c];
  // k = 16 tile + c of the next layer: group = tile, k-step q = c >> 2, lane group c & 3
  float* dst = img + ((tile * 4 + (c & 3)) * 16 + 4 * h2) * 4 + (c >> 2);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = acc[i] + bv;
    dst[4 * i] = (v > 0.f || v != v) ? v : 0.f;     // relu, NaN kept like torch
  }
}

// the same with the bias value of the lane's column already in a register (a wave that owns the same column tiles
// step after step loads its biases once)
__device__ __forceinline__ void store_relu16v(const f32x4& acc, float bv, int tile, int lane, float* __restrict__ img) {
  const int c = lane & 15, h2 = lane >> 4;
  float* dst = img + ((tile * 4 + (c & 3)) * 16 + 4 * h2) * 4 + (c >> 2);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = acc[i] + bv;
    dst[4 * i] = (v > 0.f || v != v) ? v : 0.f;     // relu, NaN kept like torch
  }
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
