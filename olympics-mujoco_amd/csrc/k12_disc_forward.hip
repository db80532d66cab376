// K12: the whole VAIL discriminator reward in ONE launch on the f32 matrix cores.
//
//   GAIL.make_discrim_reward      imitation_lib/imitation/gail_TRPO.py:320-327
//   prepare_discrim_inputs        gail_TRPO.py:297-313 (state mask)
//   VAIL.discrim_output           imitation_lib/imitation/vail_TRPO.py:18-21
//   VariationalNet.forward        imitation_lib/utils/networks.py:258-284
//   Standardizer.forward          networks.py:68-74 (fp32 batch, fp64 statistics, .float() afterwards)
//   reparameterize                networks.py:21-24
//   network shape                 examples/imitation_learning/utils.py:151-163 (in -> 256 -> 128, mu / logvar
//                                 128 -> z = 128, decoder z -> 1; the same for every robot of confs.yaml)
//
// Before: column statistics, standardise, five library GEMMs, two ReLU launches, reparameterisation and reward
// = about ten launches with every activation round-tripping HBM (0.25 ms per 4096 samples, launch-bound).
// Here a 256-thread workgroup (4 waves, two workgroups per CU) carries 32 samples through the whole chain:
// activations stay in LDS as [k][row] images (pitch 33), weights stream from L2 in the MFMA B-operand layout
// (packed once per discriminator update by disc_pack_kernel), the reparameterisation happens on the
// accumulator registers (wave w owns the same 32 columns of mu AND logvar), the decoder's 128-long dot product
// and the reward are finished by wave 0 while the other waves already stage the next tile.  The running-statistics
// update of Standardizer.forward (networks.py:76-81) stays its own launch (oly_col_stats): it is a
// grid-wide reduction that must complete before the first standardised value exists.
//
// Numerics (restated bit for bit by the oracle, oly_disc_forward_cpu):
//   xs   = f32((f64(x[mask]) - mean) / std)
//   h1_j = relu(fma-chain_k(xs_k W0_jk; 0) + b0_j)        k ascending   (v_mfma_f32_32x32x2_f32 is that chain)
//   h2_j = relu(fma-chain_k(h1_k W1_jk; 0) + b1_j)
//   mu_j, lv_j likewise over k < 128, bias after the chain
//   z_j  = mu_j + exp32(lv_j / 2) * eps_j      exp32: the Cody-Waite + degree-5 polynomial below, f32 fma only
//   d    = (chain_{k<64}(z_k wd_k) + chain_{64<=k<128}(z_k wd_k)) + bd
//   r    = -f32(log(f64(1 - 1/(1 + f32(exp(f64(-d)))) + 1e-8)))     float32 steps as numpy takes them, each
//          transcendental correctly rounded through fp64 (one per sample: free next to 148 kFLOP)
#include <cstdlib>

#include "mlp_tiles.h"
#include "oly_common.h"

using oly_mlp::act16_index;
using oly_mlp::f32x4;
using oly_mlp::layer_tiles16;
using oly_mlp::preload16;
using oly_mlp::store_relu16;

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int H1 = 256;          // encoder hidden width
constexpr int H2 = 128;          // encoder output width
constexpr int ZD = 128;          // latent width
constexpr int RT = 32;           // rows (samples) per tile
constexpr int LDP = 33;          // LDS pitch of the [k][row] images: conflict-free A-operand reads and tile write-backs
constexpr int MAX_IN = 64;
constexpr int THREADS = 256;     // 4 waves: one per SIMD; two workgroups per CU interleave on the matrix pipes

struct DiscLayout {
  int in_dim, g1;                // g1: groups of four k-steps (8 k values) in layer 1: 4 (in <= 32) or 8
  size_t w0, b0, w1, b1, wmu, bmu, wlv, blv, wd, bd;
  // the same matrices as 16-column-tile streams (mlp_tiles.h: P16[tile][group of 16 k][lane][4]) for the 16-row
  // kernel that small batches take
  size_t w0n, w1n, wmun, wlvn;
  size_t total;
};
constexpr int G1N16 = MAX_IN / 16;   // 16-wide layout, layer 1: 4 groups of 16 k (zero beyond in_dim)

__host__ __device__ inline DiscLayout disc_layout(int in_dim) {
  DiscLayout L;
  L.in_dim = in_dim;
  L.g1 = in_dim <= 32 ? 4 : 8;
  L.w0 = 0;
  L.b0 = L.w0 + (size_t)(H1 / 32) * L.g1 * 256;
  L.w1 = L.b0 + H1;
  L.b1 = L.w1 + (size_t)(H2 / 32) * (H1 / 8) * 256;
  L.wmu = L.b1 + H2;
  L.bmu = L.wmu + (size_t)(ZD / 32) * (H2 / 8) * 256;
  L.wlv = L.bmu + ZD;
  L.blv = L.wlv + (size_t)(ZD / 32) * (H2 / 8) * 256;
  L.wd = L.blv + ZD;
  L.bd = L.wd + ZD;
  L.w0n = L.bd + 4;
  L.w1n = L.w0n + (size_t)(H1 / 16) * G1N16 * 256;
  L.wmun = L.w1n + (size_t)(H2 / 16) * (H1 / 16) * 256;
  L.wlvn = L.wmun + (size_t)(ZD / 16) * (H2 / 16) * 256;
  L.total = L.wlvn + (size_t)(ZD / 16) * (H2 / 16) * 256;
  return L;
}

// P16[tile][group g][lane][q] = W[n = 16 tile + (lane & 15)][k = 16 g + 4 q + (lane >> 4)]     (0 beyond K)
__device__ __forceinline__ float packed_weight16(const float* __restrict__ W, int K, int groups, size_t r) {
  const int q = r & 3, lane = (r >> 2) & 63;
  const int g = (int)((r >> 8) % groups), tile = (int)((r >> 8) / groups);
  const int k = 16 * g + 4 * q + (lane >> 4), n = 16 * tile + (lane & 15);
  return k < K ? W[(size_t)n * K + k] : 0.f;
}

// B operand of v_mfma_f32_32x32x2_f32: lane l holds B[k = l >> 5][n = l & 31].  One 16-byte load per lane feeds
// four consecutive k-steps of one 32-column tile:
//   P[tile][group g][lane][q] = W[n = 32 tile + (lane & 31)][k = 2 (4 g + q) + (lane >> 5)]     (0 beyond K)
__device__ __forceinline__ float packed_weight(const float* __restrict__ W, int K, int groups, size_t r) {
  const int q = r & 3, lane = (r >> 2) & 63;
  const int g = (int)((r >> 8) % groups), tile = (int)((r >> 8) / groups);
  const int k = 2 * (4 * g + q) + (lane >> 5), n = 32 * tile + (lane & 31);
  return k < K ? W[(size_t)n * K + k] : 0.f;
}

__global__ void disc_pack_kernel(DiscLayout L, const float* __restrict__ W0, const float* __restrict__ B0,
                                 const float* __restrict__ W1, const float* __restrict__ B1,
                                 const float* __restrict__ Wmu, const float* __restrict__ Bmu,
                                 const float* __restrict__ Wlv, const float* __restrict__ Blv,
                                 const float* __restrict__ Wd, const float* __restrict__ Bd, float* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < L.total; e += stride) {
    float v;
    if (e < L.b0) v = packed_weight(W0, L.in_dim, L.g1, e - L.w0);
    else if (e < L.w1) v = B0[e - L.b0];
    else if (e < L.b1) v = packed_weight(W1, H1, H1 / 8, e - L.w1);
    else if (e < L.wmu) v = B1[e - L.b1];
    else if (e < L.bmu) v = packed_weight(Wmu, H2, H2 / 8, e - L.wmu);
    else if (e < L.wlv) v = Bmu[e - L.bmu];
    else if (e < L.blv) v = packed_weight(Wlv, H2, H2 / 8, e - L.wlv);
    else if (e < L.wd) v = Blv[e - L.blv];
    else if (e < L.bd) v = Wd[e - L.wd];
    else if (e < L.w0n) v = (e == L.bd) ? Bd[0] : 0.f;
    else if (e < L.w1n) v = packed_weight16(W0, L.in_dim, G1N16, e - L.w0n);
    else if (e < L.wmun) v = packed_weight16(W1, H1, H1 / 16, e - L.w1n);
    else if (e < L.wlvn) v = packed_weight16(Wmu, H2, H2 / 16, e - L.wmun);
    else v = packed_weight16(Wlv, H2, H2 / 16, e - L.wlvn);
    out[e] = v;
  }
}

// exp in float32 from fma / rint / exponent arithmetic only, so that the oracle's copy returns the same bits:
// n = rint(x log2 e), r = x - n ln2 (two-constant Cody-Waite), e^r = 1 + (r + r^2 P(r)) with a degree-5 minimax
// P, result scaled by 2^n in two exact steps.  Within 1 ulp of exp over the whole float range (checked
// against fp64 exp by tests/test_oracle_golden.py), the error class of torch's own float32 exp.
__device__ __forceinline__ float pow2i(int e) { return __int_as_float((e + 127) << 23); }
__device__ __forceinline__ float exp32(float x) {
  if (x != x) return x;
  if (x > 88.72283935546875f) return __int_as_float(0x7f800000);
  if (x < -103.97208404541016f) return 0.f;
  const float n = rintf(x * 1.4426950408889634f);
  float r = fmaf(n, -0.693145751953125f, x);
  r = fmaf(n, -1.428606765330187045e-06f, r);
  float u = 0.000198527617612853646278381f;
  u = fmaf(u, r, 0.00139304355252534151077271f);
  u = fmaf(u, r, 0.00833336077630519866943359f);
  u = fmaf(u, r, 0.0416664853692054748535156f);
  u = fmaf(u, r, 0.166666671633720397949219f);
  u = fmaf(u, r, 0.5f);
  u = 1.0f + fmaf(r * r, u, r);
  const int q = (int)n, q1 = q >> 1;
  return (u * pow2i(q1)) * pow2i(q - q1);
}

struct DiscArgs {
  long B;
  int Dx, D, ntiles;
  const float* x;
  const int* mask;
  const double *mean, *sd, *colstats;
  const float* packed;
  const float* eps;
  float *reward, *logits, *mu, *logvar;
};

// NACC 32-row x 32-column tiles of  A W  that share the A operand (k = 0 .. 8 G - 1 in order): A fragments from
// the [k][row] LDS image one group ahead, W from the packed streams two groups ahead, fully unrolled so that
// every load is in flight behind the 64-cycle MFMAs of the groups before it.
template <int G, int NACC>
__device__ __forceinline__ void layer_tiles(const float* __restrict__ aT, const float4* const (&w)[NACC], int lane,
                                            f32x16 (&acc)[NACC]) {
  const int r = lane & 31, h = lane >> 5;
  float4 b[3][NACC];
  float a[2][4];
#pragma unroll
  for (int t = 0; t < NACC; ++t) {
    b[0][t] = w[t][lane];
    if (G > 1) b[1][t] = w[t][64 + lane];
  }
  {
    const float* ap = aT + (size_t)h * LDP + r;
    a[0][0] = ap[0]; a[0][1] = ap[2 * LDP]; a[0][2] = ap[4 * LDP]; a[0][3] = ap[6 * LDP];
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g + 2 < G) {
#pragma unroll
      for (int t = 0; t < NACC; ++t) b[(g + 2) % 3][t] = w[t][(size_t)(g + 2) * 64 + lane];
    }
    if (g + 1 < G) {
      const float* ap = aT + (size_t)(8 * (g + 1) + h) * LDP + r;
      a[(g + 1) & 1][0] = ap[0]; a[(g + 1) & 1][1] = ap[2 * LDP];
      a[(g + 1) & 1][2] = ap[4 * LDP]; a[(g + 1) & 1][3] = ap[6 * LDP];
    }
    // keep the loads above ahead of this group's MFMAs (hipcc otherwise sinks each load to its first use)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        const float bq = q == 0 ? b[g % 3][t].x : q == 1 ? b[g % 3][t].y : q == 2 ? b[g % 3][t].z : b[g % 3][t].w;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][q], bq, acc[t], 0, 0, 0);
      }
    }
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

__device__ __forceinline__ void store_relu_v(const f32x16& acc, float b, int col0, int lane, float* __restrict__ hT) {
  const int col = col0 + (lane & 31), h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    const float v = acc[i] + b;
    hT[(size_t)col * LDP + row] = (v > 0.f || v != v) ? v : 0.f;     // relu, NaN kept like torch
  }
}

// G1: groups of layer 1 (4: in <= 32, 8: in <= 64).  XPT = 8 G1 * 32 / 256 input elements per thread.
template <int G1>
__global__ __launch_bounds__(THREADS, 2) void disc_forward_kernel(DiscArgs p) {
  constexpr int KIN = 8 * G1;               // zero-padded input width
  constexpr int XPT = KIN * RT / THREADS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xT = lds;                          // [KIN][LDP]   standardised input, k-major
  float* hA = xT + MAX_IN * LDP;            // [H1][LDP]    layer-1 output; later z ([ZD][LDP])
  float* hB = hA + H1 * LDP;                // [H2][LDP]    layer-2 output
  float* wd = hB + H2 * LDP;                // [ZD + 4]     decoder row + bias
  double* st = reinterpret_cast<double*>(wd + ZD + 4);   // [2][MAX_IN]  mean, std of the standardiser
  const DiscLayout L = disc_layout(p.D);
  const float* Pbase = p.packed;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // uniform: the weight streams get scalar bases
  const int r = lane & 31, h = lane >> 5;

  if (tid < ZD + 4) wd[tid] = Pbase[L.wd + tid];
  const bool standardise = p.mean || p.colstats;
  if (tid < p.D && standardise) {
    double mean, sd;
    if (p.colstats) {
      // Standardizer.update_mean_std (networks.py:76-81) from the running (count, sum, sumsq) rows:
      // _count and _sumsq start at 1e-2, the variance is floored at 1e-2
      const double cnt = p.colstats[tid] + 1e-2;
      mean = p.colstats[p.D + tid] / cnt;
      sd = sqrt(fmax((p.colstats[2 * p.D + tid] + 1e-2) / cnt - mean * mean, 1e-2));
    } else {
      mean = p.mean[tid];
      sd = p.sd[tid];
    }
    st[tid] = mean;
    st[MAX_IN + tid] = sd;
  }
  __syncthreads();

  // input element i of this thread: row m = e / KIN, column k = e % KIN of the tile (consecutive threads:
  // consecutive k of one row, so a row is one contiguous read when there is no mask)
  float xr[XPT];
  auto load_x = [&](long tile) {
    const long row0 = tile * RT;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int e = tid + THREADS * i, m = e / KIN, k = e - m * KIN;
      float v = 0.f;
      if (row0 + m < p.B && k < p.D) v = p.x[(size_t)(row0 + m) * p.Dx + (p.mask ? p.mask[k] : k)];
      xr[i] = v;
    }
  };
  auto stage_x = [&](long tile) {
    const long row0 = tile * RT;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int e = tid + THREADS * i, m = e / KIN, k = e - m * KIN;
      float v = xr[i];
      if (standardise && row0 + m < p.B && k < p.D) v = (float)(((double)v - st[k]) / st[MAX_IN + k]);
      xT[k * LDP + m] = v;
    }
  };

  // this wave's bias values (its columns do not change from tile to tile): a global load in front of every
  // accumulator write-back otherwise
  const float bias0a = Pbase[L.b0 + 64 * wave + r], bias0b = Pbase[L.b0 + 64 * wave + 32 + r];
  const float bias1 = Pbase[L.b1 + 32 * wave + r];
  const float bias_mu = Pbase[L.bmu + 32 * wave + r], bias_lv = Pbase[L.blv + 32 * wave + r];

  long tile = blockIdx.x;
  if (tile < p.ntiles) {
    load_x(tile);
    stage_x(tile);
  }
  __syncthreads();
  for (; tile < p.ntiles; tile += gridDim.x) {
    const long row0 = tile * RT;
    const long next = tile + gridDim.x;
    // the weight stream is the same for every tile: without this the compiler hoists ALL its loads out of the
    // tile loop (they are loop-invariant) and spills them
    // (an opaque zero offset, not an opaque pointer: the loads must stay global_load, not flat_load)
    int opaque0 = 0;
    asm volatile("" : "+s"(opaque0));
    const float* P = Pbase + opaque0;
    const float4* P4 = reinterpret_cast<const float4*>(P);
    {  // ---- layer 1: [32, in] x [in, 256]; wave w owns columns [64 w, 64 w + 64)
      f32x16 acc[2] = {{0}, {0}};
      const float4* const w[2] = {P4 + (L.w0 >> 2) + (size_t)(2 * wave) * G1 * 64,
                                  P4 + (L.w0 >> 2) + (size_t)(2 * wave + 1) * G1 * 64};
      layer_tiles<G1, 2>(xT, w, lane, acc);
      store_relu_v(acc[0], bias0a, 64 * wave, lane, hA);
      store_relu_v(acc[1], bias0b, 64 * wave + 32, lane, hA);
    }
    __syncthreads();
    {  // ---- layer 2: [32, 256] x [256, 128]; wave w owns columns [32 w, 32 w + 32), one chain over all k
      f32x16 acc[1] = {{0}};
      const float4* const w[1] = {P4 + (L.w1 >> 2) + (size_t)wave * (H1 / 8) * 64};
      layer_tiles<H1 / 8, 1>(hA, w, lane, acc);
      store_relu_v(acc[0], bias1, 32 * wave, lane, hB);
    }
    if (next < p.ntiles) load_x(next);       // in flight behind layer 3
    // the reparameterisation noise of this wave's columns, in flight behind layer 3
    float ev[16];
    const int col = 32 * wave + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      ev[i] = (p.eps && row < p.B) ? p.eps[(size_t)row * ZD + col] : 0.f;
    }
    __syncthreads();
    {  // ---- mu and logvar: [32, 128] x [128, 128] each; wave w owns columns [32 w, 32 w + 32) of BOTH
      f32x16 acc[2] = {{0}, {0}};
      const float4* const w[2] = {P4 + (L.wmu >> 2) + (size_t)wave * (H2 / 8) * 64,
                                  P4 + (L.wlv >> 2) + (size_t)wave * (H2 / 8) * 64};
      layer_tiles<H2 / 8, 2>(hB, w, lane, acc);
      const float bmu = bias_mu, blv = bias_lv;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (i & 3) + 8 * (i >> 2) + 4 * h;
        const float mu = acc[0][i] + bmu, lv = acc[1][i] + blv;
        float z = mu;
        if (p.eps) z = mu + exp32(lv / 2.0f) * ev[i];
        hA[(size_t)col * LDP + m] = z;
        if (row0 + m < p.B) {
          if (p.mu) p.mu[(size_t)(row0 + m) * ZD + col] = mu;
          if (p.logvar) p.logvar[(size_t)(row0 + m) * ZD + col] = lv;
        }
      }
    }
    __syncthreads();
    if (wave == 0) {  // ---- decoder + reward: lane (row r, half h) runs the chain over k in [64 h, 64 h + 64)
      float s = 0.f;
      const float* zp = hA + (size_t)(64 * h) * LDP + r;
#pragma unroll 16
      for (int k = 0; k < 64; ++k) s = fmaf(zp[(size_t)k * LDP], wd[64 * h + k], s);
      const float o = __shfl_xor(s, 32, 64);
      if (h == 0 && row0 + r < p.B) {
        const float d = (s + o) + wd[ZD];
        if (p.logits) p.logits[row0 + r] = d;
        if (p.reward) {
          const float e = (float)exp(-(double)d);
          const float pr = 1.0f / (1.0f + e);
          const float q = 1.0f - pr + 1e-8f;
          p.reward[row0 + r] = -(float)log((double)q);
        }
      }
    }
    if (next < p.ntiles) stage_x(next);      // xT was consumed by layer 1; z (hA) is read by wave 0 only
    __syncthreads();
  }
}

// The same chain for 16 samples per workgroup on v_mfma_f32_16x16x4_f32: twice as many tiles, half the matrix work per
// tile.  A batch of 4096 (the size of one discriminator fit) is 128 of the 32-row tiles on 256 CUs: half the chip idle
// and 7.7 us of dependent MFMAs per tile; as 256 tiles of 16 rows every CU works and a tile's chain is 3.8 us.  Same
// arithmetic, value for value (k-ascending fma chains; the 16-row instruction is the same chain): both kernels are
// bit-exact against the one oracle.  G1: groups of 16 inputs in layer 1 (2: in <= 32, 4: in <= 64).
constexpr int RT16 = 16;
template <int G1>
__global__ __launch_bounds__(THREADS, 2) void disc_forward16_kernel(DiscArgs p) {
  constexpr int KIN = 16 * G1;               // zero-padded input width
  constexpr int XPT = KIN * RT16 / THREADS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xT = lds;                           // [MAX_IN x 16]  standardised input image (act16 layout)
  float* hA = xT + MAX_IN * RT16;            // [H1 x 16]      layer-1 image; later z ([ZD x 16])
  float* hB = hA + H1 * RT16;                // [H2 x 16]      layer-2 image
  float* wd = hB + H2 * RT16;                // [ZD + 4]       decoder row + bias
  double* st = reinterpret_cast<double*>(wd + ZD + 4);   // [2][MAX_IN]  mean, std of the standardiser
  const DiscLayout L = disc_layout(p.D);
  const float* Pbase = p.packed;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, h2 = lane >> 4;

  // This kernel serves SMALL batches: one tile per workgroup, one workgroup per CU, nothing else on the CU to hide a
  // memory round trip.  So every load that does not depend on a barrier is requested before the barrier in front of its
  // use: the first tile's rows and layer-1 weights here, ahead of the statistics; each later layer's first two weight
  // groups ahead of the barrier that ends the layer before (they were five exposed L2 / HBM round trips per tile).
  float xr[XPT];
  auto load_x = [&](long tile) {
    const long row0 = tile * RT16;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int e = tid + THREADS * i, m = e / KIN, k = e - m * KIN;
      float v = 0.f;
      if (row0 + m < p.B && k < p.D) v = p.x[(size_t)(row0 + m) * p.Dx + (p.mask ? p.mask[k] : k)];
      xr[i] = v;
    }
  };
  auto w1_of = [&](const float4* P4, const float4* (&w)[4]) {
    const float4* base = P4 + (L.w0n >> 2) + (size_t)(4 * wave) * G1N16 * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t) w[t] = base + (size_t)t * G1N16 * 64;
  };
  float4 b1[3][4], b2[3][2], b3[3][4];       // the three layers' weight rings (first two groups requested early)
  long tile = blockIdx.x;
  if (tile < p.ntiles) {
    load_x(tile);
    const float4* w[4];
    w1_of(reinterpret_cast<const float4*>(Pbase), w);
    preload16<G1, 4>(w, lane, b1);
  }

  if (tid < ZD + 4) wd[tid] = Pbase[L.wd + tid];
  const bool standardise = p.mean || p.colstats;
  if (tid < p.D && standardise) {
    double mean, sd;
    if (p.colstats) {
      const double cnt = p.colstats[tid] + 1e-2;       // Standardizer.update_mean_std (networks.py:76-81)
      mean = p.colstats[p.D + tid] / cnt;
      sd = sqrt(fmax((p.colstats[2 * p.D + tid] + 1e-2) / cnt - mean * mean, 1e-2));
    } else {
      mean = p.mean[tid];
      sd = p.sd[tid];
    }
    st[tid] = mean;
    st[MAX_IN + tid] = sd;
  }
  __syncthreads();

  auto stage_x = [&](long tile) {
    const long row0 = tile * RT16;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int e = tid + THREADS * i, m = e / KIN, k = e - m * KIN;
      float v = xr[i];
      if (standardise && row0 + m < p.B && k < p.D) v = (float)(((double)v - st[k]) / st[MAX_IN + k]);
      xT[act16_index(k, m)] = v;
    }
  };

  if (tile < p.ntiles) stage_x(tile);
  __syncthreads();
  for (; tile < p.ntiles; tile += gridDim.x) {
    const long row0 = tile * RT16;
    const long next = tile + gridDim.x;
    int opaque0 = 0;                          // (see disc_forward_kernel: keeps the weight loads inside the tile loop)
    asm volatile("" : "+s"(opaque0));
    const float* P = Pbase + opaque0;
    const float4* P4 = reinterpret_cast<const float4*>(P);
    const float4* const base2 = P4 + (L.w1n >> 2) + (size_t)(2 * wave) * (H1 / 16) * 64;
    const float4* const w2[2] = {base2, base2 + (H1 / 16) * 64};
    const float4* const bm = P4 + (L.wmun >> 2) + (size_t)(2 * wave) * (H2 / 16) * 64;
    const float4* const bl = P4 + (L.wlvn >> 2) + (size_t)(2 * wave) * (H2 / 16) * 64;
    const float4* const w3[4] = {bm, bm + (H2 / 16) * 64, bl, bl + (H2 / 16) * 64};
    {  // ---- layer 1: [16, in] x [in, 256]; wave w owns column tiles 4 w .. 4 w + 3
      f32x4 acc[4] = {{0}, {0}, {0}, {0}};
      const float4* w[4];
      w1_of(P4, w);
      const float4* const wc[4] = {w[0], w[1], w[2], w[3]};
      layer_tiles16<G1, 4, true>(reinterpret_cast<const float4*>(xT), wc, lane, acc, b1);
      preload16<H1 / 16, 2>(w2, lane, b2);
#pragma unroll
      for (int t = 0; t < 4; ++t) store_relu16(acc[t], P + L.b0, 4 * wave + t, lane, hA);
    }
    __syncthreads();
    {  // ---- layer 2: [16, 256] x [256, 128]; wave w owns column tiles 2 w, 2 w + 1
      f32x4 acc[2] = {{0}, {0}};
      layer_tiles16<H1 / 16, 2, true>(reinterpret_cast<const float4*>(hA), w2, lane, acc, b2);
      preload16<H2 / 16, 4>(w3, lane, b3);
#pragma unroll
      for (int t = 0; t < 2; ++t) store_relu16(acc[t], P + L.b1, 2 * wave + t, lane, hB);
    }
    if (next < p.ntiles) load_x(next);       // in flight behind layer 3
    // the reparameterisation noise of this wave's columns (tiles 2 w, 2 w + 1; rows 4 h2 .. 4 h2 + 3)
    float ev[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long row = row0 + 4 * h2 + i;
        ev[t][i] = (p.eps && row < p.B) ? p.eps[(size_t)row * ZD + 16 * (2 * wave + t) + c] : 0.f;
      }
    __syncthreads();
    {  // ---- mu and logvar: [16, 128] x [128, 128] each; wave w owns column tiles 2 w, 2 w + 1 of BOTH
      f32x4 acc[4] = {{0}, {0}, {0}, {0}};
      layer_tiles16<H2 / 16, 4, true>(reinterpret_cast<const float4*>(hB), w3, lane, acc, b3);
      if (next < p.ntiles) {                 // the next tile's layer-1 weights (the same stream)
        const float4* w[4];
        w1_of(P4, w);
        const float4* const wc[4] = {w[0], w[1], w[2], w[3]};
        preload16<G1, 4>(wc, lane, b1);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int col = 16 * (2 * wave + t) + c;
        const float bmu = P[L.bmu + col], blv = P[L.blv + col];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = 4 * h2 + i;
          const float mu = acc[t][i] + bmu, lv = acc[2 + t][i] + blv;
          float z = mu;
          if (p.eps) z = mu + exp32(lv / 2.0f) * ev[t][i];
          hA[act16_index(col, m)] = z;
          if (row0 + m < p.B) {
            if (p.mu) p.mu[(size_t)(row0 + m) * ZD + col] = mu;
            if (p.logvar) p.logvar[(size_t)(row0 + m) * ZD + col] = lv;
          }
        }
      }
    }
    __syncthreads();
    if (wave == 0) {  // ---- decoder + reward: lane (row = lane & 15, half = (lane >> 4) & 1) runs k in [64 half, 64 half + 64)
      const int r = lane & 15, half = (lane >> 4) & 1;
      float s = 0.f;
#pragma unroll 16
      for (int k = 0; k < 64; ++k) s = fmaf(hA[act16_index(64 * half + k, r)], wd[64 * half + k], s);
      const float o = __shfl_xor(s, 16, 64);
      if (lane < 16 && row0 + r < p.B) {
        const float d = (s + o) + wd[ZD];
        if (p.logits) p.logits[row0 + r] = d;
        if (p.reward) {
          const float e = (float)exp(-(double)d);
          const float pr = 1.0f / (1.0f + e);
          const float q = 1.0f - pr + 1e-8f;
          p.reward[row0 + r] = -(float)log((double)q);
        }
      }
    }
    if (next < p.ntiles) stage_x(next);      // xT was consumed by layer 1; z (hA) is read by wave 0 only
    __syncthreads();
  }
}

constexpr size_t DISC16_LDS = sizeof(float) * ((MAX_IN + H1 + H2) * RT16 + ZD + 4) + sizeof(double) * 2 * MAX_IN;
static_assert(((MAX_IN + H1 + H2) * RT16 + ZD + 4) % 2 == 0, "the fp64 statistics must be 8-byte aligned");

constexpr size_t DISC_LDS = sizeof(float) * ((MAX_IN + H1 + H2) * LDP + ZD + 4) + sizeof(double) * 2 * MAX_IN;
static_assert(((MAX_IN + H1 + H2) * LDP + ZD + 4) % 2 == 0, "the fp64 statistics must be 8-byte aligned");

bool disc_shape_ok(int in_dim, int hidden, int enc_out, int z) {
  return in_dim > 0 && in_dim <= MAX_IN && hidden == H1 && enc_out == H2 && z == ZD;
}
}  // namespace

extern "C" int64_t oly_disc_packed_floats(int in_dim, int hidden, int enc_out, int z_size) {
  if (!disc_shape_ok(in_dim, hidden, enc_out, z_size)) return -1;
  return (int64_t)disc_layout(in_dim).total;
}

extern "C" int oly_disc_pack(oly_ctx* ctx, int in_dim, int hidden, int enc_out, int z_size, const float* enc_w0,
                             const float* enc_b0, const float* enc_w1, const float* enc_b1, const float* mu_w,
                             const float* mu_b, const float* lv_w, const float* lv_b, const float* dec_w,
                             const float* dec_b, float* packed, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!disc_shape_ok(in_dim, hidden, enc_out, z_size))
    OLY_FAIL(ctx, OLY_ERANGE, "oly_disc_pack: supported shape is in <= %d -> %d -> %d -> (mu, logvar) %d -> 1 (got %d, %d, %d, %d)",
             MAX_IN, H1, H2, ZD, in_dim, hidden, enc_out, z_size);
  if (!enc_w0 || !enc_b0 || !enc_w1 || !enc_b1 || !mu_w || !mu_b || !lv_w || !lv_b || !dec_w || !dec_b || !packed)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_disc_pack: NULL pointer");
  if ((reinterpret_cast<uintptr_t>(packed) & 15) != 0) OLY_FAIL(ctx, OLY_EINVAL, "oly_disc_pack: packed must be 16-byte aligned");
  hipLaunchKernelGGL(disc_pack_kernel, dim3(128), dim3(256), 0, oly_s(stream), disc_layout(in_dim), enc_w0, enc_b0,
                     enc_w1, enc_b1, mu_w, mu_b, lv_w, lv_b, dec_w, dec_b, packed);
  OLY_LAUNCH_CHECK(ctx, "disc_pack_kernel");
  return OLY_OK;
}

extern "C" int oly_disc_forward(oly_ctx* ctx, int64_t B, int Dx, int D, const float* x, const int32_t* mask,
                                const double* mean, const double* sd, const double* colstats, const float* packed,
                                const float* eps, float* reward, float* logits, float* mu, float* logvar,
                                oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (B < 0 || Dx <= 0 || D <= 0 || D > MAX_IN || (!mask && D != Dx) || (mean == nullptr) != (sd == nullptr) ||
      (mean && colstats))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_disc_forward: bad shape (B %ld, Dx %d, D %d), only one of mean / std, or both mean / std and colstats given",
             (long)B, Dx, D);
  if (B == 0) return OLY_OK;
  if (!x || !packed || (!reward && !logits && !mu && !logvar)) OLY_FAIL(ctx, OLY_EINVAL, "oly_disc_forward: NULL input or no output");
  if ((reinterpret_cast<uintptr_t>(packed) & 15) != 0) OLY_FAIL(ctx, OLY_EINVAL, "oly_disc_forward: packed must be 16-byte aligned");
  long ntiles = (B + RT - 1) / RT;
  if (ntiles > 0x7fffffffL) OLY_FAIL(ctx, OLY_ERANGE, "oly_disc_forward: B too large");
  const long slots = 2L * (ctx->num_cu > 0 ? ctx->num_cu : 256);      // two resident workgroups per CU
  // small batches: 16-row tiles fill the chip (OLY_K12_ROWS = 16 / 32 forces either kernel, for the tests)
  static const int force_rows = [] { const char* e = getenv("OLY_K12_ROWS"); return e ? atoi(e) : 0; }();
  const bool rows16 = force_rows == 16 || (force_rows != 32 && ntiles < slots);
  if (rows16) ntiles = (B + RT16 - 1) / RT16;
  if (!ctx->disc_attr_done) {
    OLY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(disc_forward16_kernel<2>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)DISC16_LDS));
    OLY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(disc_forward16_kernel<4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)DISC16_LDS));
    OLY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(disc_forward_kernel<4>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)DISC_LDS));
    OLY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(disc_forward_kernel<8>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)DISC_LDS));
    ctx->disc_attr_done = true;
  }
  DiscArgs a{(long)B, Dx, D, (int)ntiles, x, mask, mean, sd, colstats, packed, eps, reward, logits, mu, logvar};
  const dim3 grid((unsigned)(ntiles < slots ? ntiles : slots));
  if (rows16 && D <= 32) hipLaunchKernelGGL(disc_forward16_kernel<2>, grid, dim3(THREADS), DISC16_LDS, oly_s(stream), a);
  else if (rows16) hipLaunchKernelGGL(disc_forward16_kernel<4>, grid, dim3(THREADS), DISC16_LDS, oly_s(stream), a);
  else if (D <= 32) hipLaunchKernelGGL(disc_forward_kernel<4>, grid, dim3(THREADS), DISC_LDS, oly_s(stream), a);
  else hipLaunchKernelGGL(disc_forward_kernel<8>, grid, dim3(THREADS), DISC_LDS, oly_s(stream), a);
  OLY_LAUNCH_CHECK(ctx, "disc_forward_kernel");
  return OLY_OK;
}

// make_discrim_reward's whole device path in ONE call: (optionally) the re-pack of the current weights, then
// Standardizer.update_mean_std on the batch (networks.py:70,76-81: the running sums take the batch in BEFORE it is
// standardised) and the fused forward on those statistics: four launches issued from C (from Python the separate
// ctypes calls cost more host time than the kernels take at B = 4096).
extern "C" int oly_disc_reward_step(oly_ctx* ctx, int64_t B, int D, const float* x, double* colstats, int accumulate,
                                    const float* const* weights, float* packed, const float* eps, float* reward,
                                    float* logits, float* mu, float* logvar, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (B < 0 || B > 0x7fffffffL || !colstats || !packed) OLY_FAIL(ctx, OLY_EINVAL, "oly_disc_reward_step: bad B or NULL colstats / packed");
  int rc;
  if (weights) {
    rc = oly_disc_pack(ctx, D, H1, H2, ZD, weights[0], weights[1], weights[2], weights[3], weights[4], weights[5], weights[6],
                       weights[7], weights[8], weights[9], packed, stream);
    if (rc != OLY_OK) return rc;
  }
  rc = oly_col_stats(ctx, (int)B, D, x, colstats, accumulate, stream);
  if (rc != OLY_OK) return rc;
  return oly_disc_forward(ctx, B, D, D, x, nullptr, nullptr, nullptr, colstats, packed, eps, reward, logits, mu, logvar, stream);
}
