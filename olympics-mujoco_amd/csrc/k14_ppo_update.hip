// K14: the gradients of ONE PPO minibatch update (PPO.update_policy + the two backward() calls of PPO.train,
// rl/algos/ppo.py:232-282,396-410) in one launch on the f32 matrix cores: actor and critic forward, the loss terms
// of K9 (clipped surrogate, value loss, mirror-symmetry loss), and the backward pass down to the parameter
// gradients torch's autograd would hand to the optimiser.
//
// Why: at the config-3 batch (4096 environments x 400 steps, minibatches of 65536 rows) the update phase is 93 % of
// an iteration, and as library GEMMs + ~40 small torch launches per minibatch it runs at ~20 % of the f32 MFMA peak
// (hipBLASLt picks 32 x 64 macro tiles for the [256, B] x [B, 256] weight-gradient shapes).  Here a 512-thread
// workgroup owns 16-row tiles of ONE network and carries a tile through
//   forward   X -> H1 -> H2 -> out          (the tiles of K11 / K13: mlp_tiles.h, same numerics, so mu / value equal K11's)
//   loss      d out                         (one wave; K9's arithmetic per row)
//   backward  dH2 = dZ3 W3, dZ2 = dH2 [H2 > 0], dH1 = dZ2 W2, dZ1 = dH1 [H1 > 0]
//             dW3 += dZ3^T H2, dW2 += dZ2^T H1, dW1 += dZ1^T X, db += column sums
// with every activation in LDS and the WEIGHT GRADIENTS IN REGISTERS for the whole launch: wave w owns the 32 rows
// {32 w .. 32 w + 31} of dW2 / dW1 (128 + 8 KT1 accumulator registers) and 32 columns of dW3.  The matrix cores' k
// index of a weight-gradient product is the ROW of the tile, and both of its operands are accumulator tiles of
// earlier products in the C/D register layout (lane l: column l & 15, rows 4 (l >> 4) + i), so dZ is used straight
// from registers and H from a float4-per-lane LDS image: no transposes.  A workgroup ("part") walks tiles part,
// part + parts, ...; at the end it stores its partial gradients in parameter order and a finishing launch adds the
// parts (fp64, four contiguous groups in order, the group sums in order).  256 workgroups = one per CU, split between the networks by their work.
//
// Mirror-symmetry loss (ppo.py:261-268): an actor tile then holds EIGHT rows of the minibatch (tile rows 0-7) and their
// mirrored observations (tile rows 8-15), so one forward gives policy(obs) and policy(mirror_obs) of the same rows, the
// loss wave pairs row m with row m + 8 (d mirror / d det joins d mu of row m, d mirror / d mir becomes the output gradient
// of row m + 8), and ONE backward carries both.  The mirrored observations are an input array (the env's own
// mirror_clock_observation, rl/envs/wrappers.py:59-72, evaluated once per iteration); mirror_action is its (index, sign)
// table.  (Round 4's first form ran three sub-passes per tile: mirrored forward, rows forward + backward, mirrored forward
// again + backward: 101 K cycles per 16 rows against 2 x 46 K.)
//
// Numerics (restated bit for bit by the oracle twin, oly_ppo_update_cpu): forward as K11; dH chains run over
// the layer's output index ascending; a weight-gradient element is ONE f32 fma chain over the part's rows in
// tile order, inside a tile in the order 0,4,8,12,1,5,9,13,...; bias gradients are per-(column, row-group) f32
// sums combined as (s0 + s1) + (s2 + s3); exp is K12's exp32; log(std) is an input.  Against torch autograd the
// difference is summation order (tolerance in tests/test_gpu_update.py).
#include <cstdlib>

#include "oly_common.h"
#include "mlp_tiles.h"

using namespace oly_mlp;
namespace {

constexpr int UT = 512;          // 8 waves, two per SIMD
constexpr int UR = 16;           // rows per tile
constexpr int PP = 20;           // pitch of the output layer's partial tiles (rows 16-byte aligned: four columns per read)
constexpr int XI = MAX_IN * UR;  // floats of one input image
constexpr int HI = HID * UR;     // floats of one hidden image
constexpr float LOG_SQRT_2PI = 0.9189385332046727f;

// element (row, col) of a [16 rows] x [16 T cols] block in the C/D register layout of v_mfma_f32_16x16x4_f32:
// image4[tile * 64 + lane] = { value[row = 4 (lane >> 4) + i][col = 16 tile + (lane & 15)] : i = 0..3 }
__device__ __forceinline__ int c16_index(int col, int row) {
  return (((col >> 4) * 4 + (row >> 2)) * 16 + (col & 15)) * 4 + (row & 3);
}

__device__ __forceinline__ float pow2i(int e) { return __int_as_float((e + 127) << 23); }
// K12's exp32 (k12_disc_forward.hip): f32 fma / rint / exponent arithmetic only, the oracle's copy returns the same bits
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

struct UpdNet {
  const float* packed;
  float* partials;        // [parts][pstride]: a part's gradients in parameter order; pstride = grad_floats rounded up to 4
  int out_dim, normalize, parts, grad_floats, pstride, ntiles;   // ntiles: 16-row tiles of this network's work
};
struct UpdArgs {
  int B, in_dim, act_dim, pad_;
  const float *obs, *mir_obs, *action, *adv, *ret, *old_mu;
  const int* idx;
  UpdNet net[2];          // actor, critic
  const float *sd, *log_sd, *old_sd, *old_log_sd;
  float clip, vf_coeff, mirror_coeff, mirror_gscale, inv_b;
  const int* act_src;
  const float* act_sign;
  double* stat_partials;  // [parts_actor + parts_critic][NSTAT]
#ifdef OLY_DIAG
  unsigned long long* stamps;   // diagnostic build: s_memtime at every phase boundary, [block < 2][wave < 8][item < 64][16]
#endif
};
constexpr int NSTAT = 6;   // surrogate, kl, clipped, mirror, critic, rows

__device__ __forceinline__ float relu_keep_nan(float v) { return (v > 0.f || v != v) ? v : 0.f; }

// store a C/D-layout value tile (columns 16 tile ..) as the A-operand image of the next product (mlp_tiles.h: act16)
__device__ __forceinline__ void store_act16(const f32x4& v, int tile, int lane, float* __restrict__ img) {
  const int c = lane & 15, h2 = lane >> 4;
  float* dst = img + ((tile * 4 + (c & 3)) * 16 + 4 * h2) * 4 + (c >> 2);
#pragma unroll
  for (int i = 0; i < 4; ++i) dst[4 * i] = v[i];
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// mlp_tiles.h's layer_tiles16 with the weight stream read by buffer loads: ONE per-lane offset register (16 lane) for
// every stream of the kernel and a wave-uniform byte offset per (tile, group) in SGPRs / the instruction's immediate.
// (As 64-bit global addresses hipcc keeps a VGPR pair per four groups of every stream alive across the whole item
// loop, ~60 registers that then spill.)  Same MFMAs in the same order: NT 16-column tiles that share the A operand,
// G groups of 16 k, weights two groups ahead, activations one group ahead.
#ifdef OLY_K14_GLOBAL     // diagnostic build: the same loads as plain global loads
__device__ const char* g_k14_base;
#define OLY_K14_LOAD(rs, voff, so) (*reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(pbase) + (voff) + (so)))
#else
#define OLY_K14_LOAD(rs, voff, so) __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so, 0)
#endif
constexpr int RD = 2;    // weight groups in flight ahead of the MFMAs (ring of RD + 1 register sets per tile)

// the first RD weight groups of a layer's NT tiles: issued BEFORE the barrier in front of the layer (weights do not depend
// on it), so that the layer does not open with an exposed L2 round trip on all eight waves at once
template <int G, int NT>
__device__ __forceinline__ void preload16b(__amdgpu_buffer_rsrc_t rs, unsigned voff, const unsigned (&soff)[NT],
                                           u32x4 (&b)[RD + 1][NT], const float* pbase = nullptr) {
#pragma unroll
  for (int d = 0; d < RD; ++d) {
    if (d < G) {
#pragma unroll
      for (int t = 0; t < NT; ++t) b[d][t] = OLY_K14_LOAD(rs, voff, soff[t] + (unsigned)d * 1024u);
    }
  }
}

template <int G, int NT>
__device__ __forceinline__ void layer_tiles16b(const float4* __restrict__ a4, __amdgpu_buffer_rsrc_t rs, unsigned voff,
                                               const unsigned (&soff)[NT], int lane, f32x4 (&acc)[NT], u32x4 (&b)[RD + 1][NT],
                                               const float* pbase = nullptr) {
  float4 a[2];
  a[0] = a4[lane];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g + RD < G) {
#pragma unroll
      for (int t = 0; t < NT; ++t) b[(g + RD) % (RD + 1)][t] = OLY_K14_LOAD(rs, voff, soff[t] + (unsigned)(g + RD) * 1024u);
    }
    if (g + 1 < G) a[(g + 1) & 1] = a4[(g + 1) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);      // keep the loads above ahead of this group's MFMAs
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float aq = q == 0 ? a[g & 1].x : q == 1 ? a[g & 1].y : q == 2 ? a[g & 1].z : a[g & 1].w;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const u32x4 bb = b[g % (RD + 1)][t];
        const float bq = __uint_as_float(q == 0 ? bb.x : q == 1 ? bb.y : q == 2 ? bb.z : bb.w);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bq, acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

#ifdef OLY_DIAG
#define OLY_STAMP(i)                                                                                      \
  do {                                                                                                    \
    if (p.stamps && (blockIdx.x == 0 || blockIdx.x == (unsigned)p.net[0].parts) && lane == 0 && it < 64)  \
      p.stamps[(((blockIdx.x ? 1 : 0) * 8 + wave) * 64 + it) * 16 + (i)] = __builtin_amdgcn_s_memtime();  \
  } while (0)
#else
#define OLY_STAMP(i)
#endif

// values the optimiser must treat as new (see the loss block)
__device__ __forceinline__ int opaque_v(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ float opaque_s(float x) { asm volatile("" : "+s"(x)); return x; }

template <int KT1>      // groups of 16 inputs
__global__ __launch_bounds__(UT, 2) void ppo_update_kernel(UpdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xA = lds;                 // [2][XI]   input rows as the A operand of layer 1 (double-buffered over work items)
  float* xC = xA + 2 * XI;         // [2][XI]   the same rows in the C/D layout: B operand of dW1
  float* h1A = xC + 2 * XI;        // [HI]      layer-1 activations, A operand of layer 2
  float* h1C = h1A + HI;           // [HI]      the same in the C/D layout: B operand of dW2, ReLU mask of dH1
  float* h2A = h1C + HI;           // [HI]      layer-2 activations, A operand of the output layer
  float* dz2A = h2A;               //           dZ2, A operand of dH1 (the output layer has read h2A two barriers earlier)
  float* part = h2A + HI;          // [8][16][PP] output layer: the eight partial chains
  float* mirL = part + 8 * UR * PP;  // [16][16] policy(mirror_obs) of the tile
  float* dmirL = mirL + UR * 16;   // [16][16]  d loss / d policy(mirror_obs)
  float* dz3A = dmirL + UR * 16;   // [256]     d loss / d out as an A operand (one group of 16)
  float* dz3C = dz3A + 256;        // [256]     the same in the C/D layout: A operand of dW3
  float* cstL = dz3C + 256;        // [8][16]   per-column constants of the loss (wave 0 reads them per tile)
  float* lossL = cstL + 128;       // [10][64]  wave 0: the loss inputs of the next tile (action 4, old mean 4, adv, ret per lane)
  float* termL = lossL + 640;      // [2][16][16] wave 0: per-(row, column) log-prob terms of the new / old policy
  double* stL = reinterpret_cast<double*>(termL + 512);  // [NSTAT][64] wave 0's per-lane statistics
  float4* parkL = reinterpret_cast<float4*>(h2A);   // [16][64] wave 0 parks half of its dW2 accumulators here while it runs the
                                   //           loss (the loss needs ~70 registers, 40 are free); h2A is dead between the
                                   //           output layer and dZ2
  float4* dW1L = reinterpret_cast<float4*>(stL + NSTAT * 64);   // [16][KT1][64] dW1 in the C/D layout: the accumulators
                                   //           of a tile are read, run through its four MFMAs and written back

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, j = lane >> 4;
  const bool critic = (int)blockIdx.x >= p.net[0].parts;
  const UpdNet net = p.net[critic ? 1 : 0];
  const int part_id = (int)blockIdx.x - (critic ? p.net[0].parts : 0);
  const int parts = net.parts, out_dim = net.out_dim, in_dim = p.in_dim, B = p.B;
  const PackLayout L = pack_layout(in_dim, out_dim);
  const float* __restrict__ P = net.packed;
  const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, (int)(L.total * sizeof(float)), 0x00020000);
  const unsigned voff = 16u * (unsigned)lane;
  const bool mirror = !critic && p.mir_obs != nullptr;
  const int ta = 2 * wave;       // this wave's hidden column tiles: ta, ta + 1

  f32x4 dW2[2][16], dW3[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int k = 0; k < 16; ++k) dW2[t][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KT1; ++k) dW1L[((ta + t) * KT1 + k) * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    dW3[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float db2[2] = {0.f, 0.f}, db1[2] = {0.f, 0.f}, db3 = 0.f;
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < NSTAT; ++q) stL[q * 64 + lane] = 0.0;
  }

  float bias1[2], bias2[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    bias1[t] = P[L.b1 + 16 * (ta + t) + c];
    bias2[t] = P[L.b2 + 16 * (ta + t) + c];
  }
  // staging role of this thread: input column k of rows m0 and m0 + 8
  const int sk = tid & 63, sm = tid >> 6;
  float nmean = 0.f, nstd = 1.f;
  if (net.normalize && sk < in_dim) { nmean = P[L.mean + sk]; nstd = P[L.std + sk]; }
  // per-column constants of the loss: cstL[f][col], f = bias, sd^2, 2 sd^2, 2 old_sd^2, log sd, log old_sd, mirror sign, mirror index
  if (tid < 16) {
    const int col = tid;
    const bool on = col < out_dim;
    float v[8] = {on ? P[L.b3 + col] : 0.f, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f};
    if (!critic && on) {
      const float s = p.sd[col], os = p.old_sd[col];
      v[1] = s * s;
      v[2] = 2.0f * (s * s);
      v[3] = 2.0f * (os * os);
      v[4] = p.log_sd[col];
      v[5] = p.old_log_sd[col];
      if (mirror) { v[6] = p.act_sign[col]; v[7] = __int_as_float(p.act_src[col]); }
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) cstL[f * 16 + col] = v[f];
  }

  // A tile is 16 rows of the network's forward / backward.  With the mirror loss an actor tile holds EIGHT rows of the
  // minibatch (tile rows 0-7) and their mirrored observations (tile rows 8-15: row m + 8 mirrors row m), so that one
  // forward gives policy(obs) and policy(mirror_obs) of the same rows, the loss sees both, and one backward carries
  // d loss / d mu and d loss / d mirror together (ppo.py:261-268).
  const int rpt = mirror ? 8 : UR;                 // rows of the minibatch per tile
  const int n_items = (net.ntiles - part_id + parts - 1) / parts;

  // the input rows of work item `it`: thread's two elements (tile rows sm and sm + 8), normalised, zero-padded
  auto load_x = [&](int it, float (&v)[2]) {
    const int tile = part_id + it * parts;
    const int sk_here = opaque_v(sk);              // (as a loop invariant the per-thread base address was spilled)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = mirror ? tile * 8 + sm : tile * UR + sm + 8 * i;
      const float* __restrict__ src = (mirror && i == 1) ? p.mir_obs : p.obs;
      float x = 0.f;
      if (row < B && sk < in_dim) {
        const long r = p.idx ? (long)p.idx[row] : (long)row;
        x = src[r * in_dim + sk_here];
        if (net.normalize) x = (x - nmean) / nstd;
      }
      v[i] = x;
    }
  };
  auto store_x = [&](int buf, const float (&v)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = sm + 8 * i;
      xA[buf * XI + act16_index(sk, m)] = v[i];
      xC[buf * XI + c16_index(sk, m)] = v[i];
    }
  };

  // byte offsets of this wave's weight streams
  const unsigned so1[2] = {(unsigned)(L.w1n * 4) + (unsigned)ta * (G1N * 1024u), (unsigned)(L.w1n * 4) + (unsigned)(ta + 1) * (G1N * 1024u)};
  const unsigned so2[2] = {(unsigned)(L.w2n * 4) + (unsigned)ta * (HID / 16 * 1024u), (unsigned)(L.w2n * 4) + (unsigned)(ta + 1) * (HID / 16 * 1024u)};
  const unsigned so3[1] = {(unsigned)(L.w3n * 4) + (unsigned)(2 * wave) * 1024u};
  const unsigned so3t[2] = {(unsigned)(L.w3t * 4) + (unsigned)ta * (T3N * 1024u), (unsigned)(L.w3t * 4) + (unsigned)(ta + 1) * (T3N * 1024u)};
  const unsigned so2t[2] = {(unsigned)(L.w2t * 4) + (unsigned)ta * (HID / 16 * 1024u), (unsigned)(L.w2t * 4) + (unsigned)(ta + 1) * (HID / 16 * 1024u)};
  u32x4 wb[RD + 1][2];     // the weight ring of the two-tile layers: its first RD groups are requested before the barrier

  // wave 0's loss inputs (lane: row c, columns 4 j ..) of a tile: fetched one tile ahead into registers behind the
  // backward phases, parked in LDS at the item's end (a random row of HBM costs ~2 us: in front of the loss they were
  // 4 K of a tile's 47 K cycles on every wave)
  // the buffer row behind a tile's loss lane (-1: none).  Resolved at the TOP of the item before: as the first step of
  // loss_fetch the index load was a memory round trip on wave 0 between its loss and the barrier the other seven
  // waves were already waiting at (1.5 K of a tile's 46 K cycles)
  auto loss_row = [&](int tile_) -> int {
    const int row = tile_ * rpt + c;
    if (tile_ < net.ntiles && c < rpt && row < B) return p.idx ? p.idx[row] : row;
    return -1;
  };
  auto loss_fetch = [&](int r32, float (&v)[10]) {
#pragma unroll
    for (int q = 0; q < 10; ++q) v[q] = 0.f;
    if (r32 >= 0) {
      const long r = r32;
      const int j = opaque_v(lane) >> 4;          // (the hoisted per-lane base addresses came back as scratch reloads)
      if (critic) {
        v[9] = p.ret[r];
      } else {
        v[8] = p.adv[r];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int col = 4 * j + cc;
          if (col < out_dim) { v[cc] = p.action[r * out_dim + col]; v[4 + cc] = p.old_mu[r * out_dim + col]; }
        }
      }
    }
  };
  auto loss_park = [&](const float (&v)[10]) {
#pragma unroll
    for (int q = 0; q < 10; ++q) lossL[q * 64 + lane] = v[q];
  };

  // dW2 (own rows) += dZ2^T H1 of one tile: 128 MFMAs per wave, the two tiles' chains alternating (consecutive MFMAs on one
  // accumulator wait on the 40-cycle dependent latency of the 32-cycle instruction)
  auto dW2_accumulate = [&](const f32x4 (&dz)[2], const float4* __restrict__ h1c) {
    float4 hb[2];
    hb[0] = h1c[lane];
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) {
      if (kt + 1 < 16) hb[(kt + 1) & 1] = h1c[(kt + 1) * 64 + lane];
      const float4 b = hb[kt & 1];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float bq = q == 0 ? b.x : q == 1 ? b.y : q == 2 ? b.z : b.w;
        dW2[0][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dz[0][q], bq, dW2[0][kt], 0, 0, 0);
        dW2[1][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dz[1][q], bq, dW2[1][kt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  float xn[2];
  if (n_items > 0) {
    // two memory round trips in front of the first tile, not four: the index behind the loss lane goes out with the
    // indices behind the input rows, the loss inputs with the rows themselves (at the reference's minibatch of 64 a
    // launch is one tile per workgroup and these were 3 of its 26 us)
    const int lr = wave == 0 ? loss_row(part_id) : -1;
    load_x(0, xn);
    float lv[10];
    if (wave == 0) loss_fetch(lr, lv);
    preload16b<KT1, 2>(rsP, voff, so1, wb, P);
    store_x(0, xn);
    if (wave == 0) loss_park(lv);
  }
  __syncthreads();

  for (int it = 0; it < n_items; ++it) {
    const int tile = part_id + it * parts;
    const int pb = it & 1;
    const bool more = it + 1 < n_items;
    if (more) load_x(it + 1, xn);           // in flight behind this item's layers
    const int lrow_next = (wave == 0 && more) ? loss_row(tile + parts) : -1;
    const float4* xA4 = reinterpret_cast<const float4*>(xA + pb * XI);
    const float4* xC4 = reinterpret_cast<const float4*>(xC + pb * XI);
    float4* h1C4 = reinterpret_cast<float4*>(h1C);
    OLY_STAMP(0);

    {  // ---- layer 1
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      layer_tiles16b<KT1, 2>(xA4, rsP, voff, so1, lane, acc, wb, P);
      preload16b<HID / 16, 2>(rsP, voff, so2, wb, P);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = relu_keep_nan(acc[t][i] + bias1[t]);
        store_act16(v, ta + t, lane, h1A);
        h1C4[(ta + t) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    OLY_STAMP(1);
    __syncthreads();
    OLY_STAMP(2);
    f32x4 h2own[2];
    u32x4 wb1[RD + 1][1];                   // the output layer's two weight groups: requested in front of barrier 2 (behind
                                            // it they were an exposed L2 round trip in a phase of eight MFMAs)
    {  // ---- layer 2
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      layer_tiles16b<HID / 16, 2>(reinterpret_cast<const float4*>(h1A), rsP, voff, so2, lane, acc, wb, P);
      preload16b<2, 1>(rsP, voff, so3, wb1, P);

#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) h2own[t][i] = relu_keep_nan(acc[t][i] + bias2[t]);
        store_act16(h2own[t], ta + t, lane, h2A);
      }
    }
    OLY_STAMP(3);
    __syncthreads();
    OLY_STAMP(4);
    {  // ---- output layer: chain `wave` of the eight partial chains (k in [32 wave, 32 wave + 32))
      f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
      layer_tiles16b<2, 1>(reinterpret_cast<const float4*>(h2A) + (size_t)(2 * wave) * 64, rsP, voff, so3, lane, acc, wb1, P);
      preload16b<1, 2>(rsP, voff, so3t, wb, P);
#pragma unroll
      for (int i = 0; i < 4; ++i) part[(wave * UR + 4 * j + i) * PP + c] = acc[0][i];
    }
    OLY_STAMP(5);
    __syncthreads();
    OLY_STAMP(6);
    if (wave == 0) {  // ---- out = partial chains in order + bias; loss terms; d loss / d out   (lane: row c, columns 4 j ..)
      // Everything this block derives from the lane number or from launch constants is derived HERE, from copies the
      // compiler cannot trace back (an empty asm): as loop invariants they were hoisted in front of the tile loop, could not
      // keep a register through the matrix phases, and came back as ~17 scratch reloads, each waited for on the spot, in
      // the one phase seven waves wait for.
      const int lane_here = opaque_v(lane);
      const float clip_here = opaque_s(p.clip), vf_here = opaque_s(p.vf_coeff);
      const int lane = lane_here, c = lane_here & 15, j = lane_here >> 4;
      const float lo = 1.0f - clip_here, hi = 1.0f + clip_here;
      const float zero_here = __int_as_float(opaque_v(0));
      const bool valid = c < rpt && tile * rpt + c < B;      // the loss lane: row c of the tile (c >= rpt: a mirrored row)
#pragma unroll
      for (int k = 0; k < 16; ++k) parkL[k * 64 + lane] = make_float4(dW2[0][k][0], dW2[0][k][1], dW2[0][k][2], dW2[0][k][3]);
      float o[4], g[4] = {0.f, 0.f, 0.f, 0.f};
      const float act[4] = {lossL[lane], lossL[64 + lane], lossL[128 + lane], lossL[192 + lane]};
      const float omu[4] = {lossL[256 + lane], lossL[320 + lane], lossL[384 + lane], lossL[448 + lane]};
      const float advv = lossL[512 + lane], retv = lossL[576 + lane];
      const float4* cst4 = reinterpret_cast<const float4*>(cstL);
      const float4 b3q = cst4[j];
      const float b3v[4] = {b3q.x, b3q.y, b3q.z, b3q.w};
      {
        float4 pq[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) pq[w] = reinterpret_cast<const float4*>(part + (w * UR + c) * PP)[j];
        float4 sq = pq[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) { sq.x += pq[w].x; sq.y += pq[w].y; sq.z += pq[w].z; sq.w += pq[w].w; }
#pragma unroll
        for (int w = 0; w < 4; ++w) pq[w] = reinterpret_cast<const float4*>(part + ((w + 4) * UR + c) * PP)[j];
#pragma unroll
        for (int w = 0; w < 4; ++w) { sq.x += pq[w].x; sq.y += pq[w].y; sq.z += pq[w].z; sq.w += pq[w].w; }
        o[0] = sq.x + b3v[0]; o[1] = sq.y + b3v[1]; o[2] = sq.z + b3v[2]; o[3] = sq.w + b3v[3];
      }
      if (mirror) {      // every row's means where the partner row's lanes can read them; the mirrored rows' gradients start at 0
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          mirL[c * 16 + 4 * j + cc] = o[cc];
          dmirL[c * 16 + 4 * j + cc] = zero_here;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (critic) {
        // critic_loss = vf_coeff * mse(ret, value)   (ppo.py:256)
        const float v = o[0];
        if (j == 0 && valid) {
          const float dv = retv - v;
          stL[4 * 64 + lane] += (double)(dv * dv);
          g[0] = vf_here * 2.0f * (v - retv) * p.inv_b;
        }
      } else {
        float t[4], lt[4], olt[4];
        const float4 q1 = cst4[4 + j], q2 = cst4[8 + j], q3 = cst4[12 + j], q4 = cst4[16 + j], q5 = cst4[20 + j],
                     q6 = cst4[24 + j], q7 = cst4[28 + j];
        const float sd2[4] = {q1.x, q1.y, q1.z, q1.w}, sdv2[4] = {q2.x, q2.y, q2.z, q2.w}, osdv2[4] = {q3.x, q3.y, q3.z, q3.w};
        const float lsd[4] = {q4.x, q4.y, q4.z, q4.w}, olsd[4] = {q5.x, q5.y, q5.z, q5.w}, msg[4] = {q6.x, q6.y, q6.z, q6.w};
        const int msrc[4] = {__float_as_int(q7.x), __float_as_int(q7.y), __float_as_int(q7.z), __float_as_int(q7.w)};
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const bool on = 4 * j + cc < out_dim;
          t[cc] = act[cc] - o[cc];
          const float ot = act[cc] - omu[cc];
          // Normal.log_prob: -((v - loc)**2) / (2*var) - log(scale) - log(sqrt(2 pi))
          lt[cc] = on ? -(t[cc] * t[cc]) / sdv2[cc] - lsd[cc] - LOG_SQRT_2PI : 0.f;
          olt[cc] = on ? -(ot * ot) / osdv2[cc] - olsd[cc] - LOG_SQRT_2PI : 0.f;
        }
        // sums over the action dimension in order, as torch's .sum(-1) does: the terms of a row go through LDS so that
        // every lane of the row adds all sixteen in sequence (columns >= act_dim hold exact zeros: x + 0 = x)
        float4* t4 = reinterpret_cast<float4*>(termL);
        t4[c * 4 + j] = make_float4(lt[0], lt[1], lt[2], lt[3]);
        t4[64 + c * 4 + j] = make_float4(olt[0], olt[1], olt[2], olt[3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float lp = 0.f, olp = 0.f;
        {
          float4 a4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) a4[q] = t4[c * 4 + q];
#pragma unroll
          for (int q = 0; q < 4; ++q) { lp += a4[q].x; lp += a4[q].y; lp += a4[q].z; lp += a4[q].w; }
#pragma unroll
          for (int q = 0; q < 4; ++q) a4[q] = t4[64 + c * 4 + q];
#pragma unroll
          for (int q = 0; q < 4; ++q) { olp += a4[q].x; olp += a4[q].y; olp += a4[q].z; olp += a4[q].w; }
        }
        const float log_ratio = lp - olp;
        const float ratio = exp32(log_ratio);
        const float cpi = ratio * advv;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float cl = rc * advv;
        if (j == 0 && valid) {
          stL[0 * 64 + lane] += (double)fminf(cpi, cl);
          stL[1 * 64 + lane] += (double)((ratio - 1.0f) - log_ratio);
          stL[2 * 64 + lane] += (fabsf(ratio - 1.0f) > p.clip) ? 1.0 : 0.0;
          stL[5 * 64 + lane] += 1.0;
        }
        // torch.min splits the gradient on ties; clamp passes it inside [lo, hi] (inclusive)
        const float inr = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
        const float wgt = cpi < cl ? 1.0f : (cpi == cl ? 0.5f + 0.5f * inr : inr);
        const float g_lp = -p.inv_b * advv * wgt * ratio;
        double mir_sq = 0.0;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int col = 4 * j + cc;
          if (col < out_dim && valid) {
            float gm = g_lp * t[cc] / sd2[cc];
            if (mirror) {
              // (det - mirror_action(policy(mirror_obs)))^2 mean   (ppo.py:261-268); det = mu; the mirrored row is tile row c + 8
              const float d = o[cc] - msg[cc] * mirL[(c + 8) * 16 + msrc[cc]];
              mir_sq += (double)(d * d);
              const float gg = p.mirror_gscale * d;
              gm = gm + p.mirror_coeff * gg;
              dmirL[(c + 8) * 16 + msrc[cc]] = p.mirror_coeff * (-msg[cc] * gg);
            }
            g[cc] = gm;
          }
        }
        if (mirror) {
          stL[3 * 64 + lane] += mir_sq;
          // the mirrored rows' lanes pick up d loss / d policy(mirror_obs) their partner rows have just left
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (c >= 8) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) g[cc] = (4 * j + cc < out_dim) ? dmirL[c * 16 + 4 * j + cc] : 0.f;
          }
        }
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int col = 4 * j + cc;
        dz3A[act16_index(col, c)] = g[cc];
        dz3C[c16_index(col, c)] = g[cc];
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float4 v = parkL[k * 64 + lane];
        dW2[0][k] = f32x4{v.x, v.y, v.z, v.w};
      }
    }
    OLY_STAMP(7);
    // wave 0: the next tile's loss inputs, requested behind this item's backward phases (the item before a tile's loss)
    float lnext[10];
    const bool fetch_loss = wave == 0 && more;
    if (fetch_loss) loss_fetch(lrow_next, lnext);
    {
      __syncthreads();
      OLY_STAMP(8);
      // ---- dH2 (own tiles) = dZ3 W3; dZ2 = dH2 [H2 > 0]; dW3 (own columns) += dZ3^T H2; dW2 (own rows) += dZ2^T H1
      f32x4 dz2[2];
      {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        layer_tiles16b<1, 2>(reinterpret_cast<const float4*>(dz3A), rsP, voff, so3t, lane, acc, wb, P);
        preload16b<HID / 16, 2>(rsP, voff, so2t, wb, P);
        const float4 z3 = reinterpret_cast<const float4*>(dz3C)[lane];
        if (wave == 0) {
          db3 += z3.x;
          db3 += z3.y;
          db3 += z3.z;
          db3 += z3.w;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            dz2[t][i] = h2own[t][i] > 0.f ? acc[t][i] : 0.f;
            db2[t] += dz2[t][i];
          }
          store_act16(dz2[t], ta + t, lane, dz2A);
          dW3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(z3.x, h2own[t][0], dW3[t], 0, 0, 0);
          dW3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(z3.y, h2own[t][1], dW3[t], 0, 0, 0);
          dW3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(z3.z, h2own[t][2], dW3[t], 0, 0, 0);
          dW3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(z3.w, h2own[t][3], dW3[t], 0, 0, 0);
        }
      }
      dW2_accumulate(dz2, h1C4);
      OLY_STAMP(9);
      __syncthreads();
      OLY_STAMP(10);
      {  // ---- dH1 (own tiles) = dZ2 W2; dZ1 = dH1 [H1 > 0]; dW1 (own rows) += dZ1^T X
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        layer_tiles16b<HID / 16, 2>(reinterpret_cast<const float4*>(dz2A), rsP, voff, so2t, lane, acc, wb, P);
        f32x4 dz1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float4 h = h1C4[(ta + t) * 64 + lane];
          dz1[t][0] = h.x > 0.f ? acc[t][0] : 0.f;
          dz1[t][1] = h.y > 0.f ? acc[t][1] : 0.f;
          dz1[t][2] = h.z > 0.f ? acc[t][2] : 0.f;
          dz1[t][3] = h.w > 0.f ? acc[t][3] : 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) db1[t] += dz1[t][i];
        }
        // dW1's accumulators live in LDS: all 2 KT1 tiles are read first, then their MFMAs run as independent chains
        // (one tile at a time is four DEPENDENT MFMAs behind an LDS round trip), then all go back
        f32x4 a1[2][KT1];
        float4 xb[KT1];
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt) {
          xb[kt] = xC4[kt * 64 + lane];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float4 a0 = dW1L[((ta + t) * KT1 + kt) * 64 + lane];
            a1[t][kt] = f32x4{a0.x, a0.y, a0.z, a0.w};
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int kt = 0; kt < KT1; ++kt) {
            const float bq = q == 0 ? xb[kt].x : q == 1 ? xb[kt].y : q == 2 ? xb[kt].z : xb[kt].w;
#pragma unroll
            for (int t = 0; t < 2; ++t) a1[t][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dz1[t][q], bq, a1[t][kt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
            dW1L[((ta + t) * KT1 + kt) * 64 + lane] = make_float4(a1[t][kt][0], a1[t][kt][1], a1[t][kt][2], a1[t][kt][3]);
        }
      }
    }
    if (more) {
      store_x(pb ^ 1, xn);
      preload16b<KT1, 2>(rsP, voff, so1, wb, P);
    }
    if (fetch_loss) loss_park(lnext);
    OLY_STAMP(11);
    __syncthreads();
    OLY_STAMP(12);
  }

  // ---- this part's partial gradients, in parameter order: W1 [256, in] | b1 | W2 [256, 256] | b2 | W3 [out, 256] | b3
  float* __restrict__ G = net.partials + (size_t)part_id * net.pstride;
  const size_t oW1 = 0, ob1 = (size_t)HID * in_dim, oW2 = ob1 + HID, ob2 = oW2 + (size_t)HID * HID, oW3 = ob2 + HID,
               ob3 = oW3 + (size_t)out_dim * HID;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n0 = 16 * (ta + t) + 4 * j;        // rows n0 + i of dW2 / dW1
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) G[oW2 + (size_t)(n0 + i) * HID + 16 * kt + c] = dW2[t][kt][i];
    }
#pragma unroll
    for (int kt = 0; kt < KT1; ++kt) {
      const int k = 16 * kt + c;
      if (k < in_dim) {
        const float4 a = dW1L[((ta + t) * KT1 + kt) * 64 + lane];
        G[oW1 + (size_t)(n0 + 0) * in_dim + k] = a.x;
        G[oW1 + (size_t)(n0 + 1) * in_dim + k] = a.y;
        G[oW1 + (size_t)(n0 + 2) * in_dim + k] = a.z;
        G[oW1 + (size_t)(n0 + 3) * in_dim + k] = a.w;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (4 * j + i < out_dim) G[oW3 + (size_t)(4 * j + i) * HID + 16 * (ta + t) + c] = dW3[t][i];
    }
    // bias gradients: (s0 + s1) + (s2 + s3) over the four row groups
    float s2 = db2[t], s1 = db1[t];
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    if (j == 0) {
      G[ob2 + 16 * (ta + t) + c] = s2;
      G[ob1 + 16 * (ta + t) + c] = s1;
    }
  }
  if (wave == 0) {
    float s3 = db3;
    s3 += __shfl_xor(s3, 16, 64);
    s3 += __shfl_xor(s3, 32, 64);
    if (j == 0 && c < out_dim) G[ob3 + c] = s3;
    double* S = p.stat_partials + (size_t)blockIdx.x * NSTAT;
#pragma unroll
    for (int q = 0; q < NSTAT; ++q) {
      const double s = wave_sum(stL[q * 64 + lane]);
      if (lane == 0) S[q] = s;
    }
  }
}

constexpr int ADAM_MAX_BLOCKS = 512;   // 256-element blocks of a network's squared-norm partials

struct FinArgs {
  UpdNet net[2];
  float* grad[2];
  const double* stat_partials;
  double* scal_out;
  double* gnorm;          // [2][ADAM_MAX_BLOCKS] block partials of sum g^2 (for oly_ppo_adam_step) or NULL
  const float* log_sd;
  int B, act_dim, mirror;
  float vf_coeff;
};

// grad[e] = the parts added in fp64 and rounded once; block 0 also finishes the six scalars.  A block owns 256 consecutive
// elements of one network, a lane four of them (one 16-byte load per part: the rows are padded to a multiple of four
// floats); wave g adds the parts of group g (four contiguous groups of ceil(parts / 4), each in order, eight loads in
// flight per lane), the four group sums are combined as ((S0 + S1) + S2) + S3.  (One thread per element over all 128
// parts ran at 0.8 TB/s: 1228 waves on the chip, each waiting out sixteen dependent HBM round trips.)
__global__ __launch_bounds__(256) void ppo_update_finish_kernel(FinArgs f, int blocks_a) {
  __shared__ double sh[4][64][4];
  const int n = (int)blockIdx.x >= blocks_a;
  const long gf = f.net[n].grad_floats, ps = f.net[n].pstride;
  const int parts = f.net[n].parts;
  const float* __restrict__ src = f.net[n].partials;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long e = 256L * ((int)blockIdx.x - (n ? blocks_a : 0)) + 4 * lane;
  const bool on = e < ps;                     // whole float4s: the pad of a row is never stored (e + i < gf below)
  const int chunk = (parts + 3) / 4;
  const int q0 = grp * chunk, q1 = min(parts, q0 + chunk);
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  int q = q0;
  for (; q + 8 <= q1; q += 8) {
    float4 a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = on ? *reinterpret_cast<const float4*>(src + (size_t)(q + u) * ps + e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s[0] += (double)a[u].x; s[1] += (double)a[u].y; s[2] += (double)a[u].z; s[3] += (double)a[u].w; }
  }
  for (; q < q1; ++q) {
    const float4 a = on ? *reinterpret_cast<const float4*>(src + (size_t)q * ps + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    s[0] += (double)a.x; s[1] += (double)a.y; s[2] += (double)a.z; s[3] += (double)a.w;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) sh[grp][lane][i] = s[i];
  __syncthreads();
  if (grp == 0) {
    // ... and this block's partial of the squared gradient norm (the same 256-element blocks as grad_sumsq_kernel)
    double sq = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (e + i < gf) {
        const float g = (float)(((sh[0][lane][i] + sh[1][lane][i]) + sh[2][lane][i]) + sh[3][lane][i]);
        f.grad[n][e + i] = g;
        sq += (double)g * (double)g;
      }
    }
    sq = wave_sum(sq);
    if (f.gnorm && lane == 0) f.gnorm[n * ADAM_MAX_BLOCKS + ((int)blockIdx.x - (n ? blocks_a : 0))] = sq;
  }
  if (blockIdx.x != 0) return;
  // the six scalars: every workgroup's partial, added in workgroup order (one fp64 chain per scalar).  The partials are
  // fetched by all 256 threads at once and the chains run out of LDS: as plain global loads inside the chains they were
  // `total` dependent memory round trips on ONE wave, the longest path of the whole launch.
  __shared__ double stage[256 * NSTAT];
  const int total = f.net[0].parts + f.net[1].parts;
  double st = 0.0;
  for (int b0 = 0; b0 < total; b0 += 256) {
    const int nb = min(256, total - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * NSTAT; i += 256) stage[i] = f.stat_partials[(size_t)b0 * NSTAT + i];
    __syncthreads();
    if (threadIdx.x < NSTAT)
      for (int b = 0; b < nb; ++b) st += stage[b * NSTAT + threadIdx.x];
  }
  if (threadIdx.x < NSTAT) {
    const int qq = threadIdx.x;
    const double invB = 1.0 / (double)f.B;
    // scal_out: actor, entropy_penalty, critic, approx_kl, mirror, clip_fraction
    if (qq == 0) f.scal_out[0] = -st * invB;
    if (qq == 1) f.scal_out[3] = st * invB;
    if (qq == 2) f.scal_out[5] = st * invB;
    if (qq == 3) f.scal_out[4] = f.mirror ? st / ((double)f.B * f.act_dim) : 0.0;
    if (qq == 4) f.scal_out[2] = (double)f.vf_coeff * st * invB;
    if (qq == 5) {
      // entropy of a fixed-std Gaussian: the same f32 row value for every row (Normal.entropy: 0.5 + 0.5 log(2 pi) + log(std))
      float ent = 0.f;
      for (int a = 0; a < f.act_dim; ++a) ent += (0.5f + LOG_SQRT_2PI) + f.log_sd[a];
      f.scal_out[1] = -(st * (double)ent) / ((double)f.B * f.act_dim);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The optimiser half of the update (rl/algos/ppo.py:396-410): torch.nn.utils.clip_grad_norm_ + Adam.step for both
// networks on flat parameter / gradient / moment buffers, the stepped value written straight to its places in the
// packed streams: ONE launch instead of torch's ~60 (the foreach Adam and the norm clip are ~30 small kernels per
// network; at every minibatch size the host could not issue them as fast as the GPU ran them).
struct AdamNet {
  float *param, *exp_avg, *exp_avg_sq;
  const float* grad;
  int n, blocks;          // elements; 256-element blocks of the norm pass
  float* packed;          // the network's oly_mlp_pack stream (or NULL)
  const float *mean, *std;
  int in_dim, out_dim;
};
struct AdamArgs {
  AdamNet net[2];
  double* sumsq;          // [2][OLY_ADAM_MAX_BLOCKS] block partials of sum g^2
  float w1, beta2, w2, eps, neg_step, bc2_sqrt, max_norm;
};

// block partials of sum g^2 over 256-element blocks (the finishing launch leaves the same ones): lane l squares elements
// 4 l .. 4 l + 3 of the block in order, fp64, then the 64-lane tree; one block per wave
__global__ __launch_bounds__(256) void grad_sumsq_kernel(AdamArgs a, int wgs_a) {
  const int n = (int)blockIdx.x >= wgs_a;
  const AdamNet nt = a.net[n];
  const int blk = 4 * ((int)blockIdx.x - (n ? wgs_a : 0)) + (threadIdx.x >> 6);
  const long e = 256L * blk + 4 * (threadIdx.x & 63);
  double sq = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (e + i < nt.n) {
      const double g = (double)nt.grad[e + i];
      sq += g * g;
    }
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0 && blk < nt.blocks) a.sumsq[n * ADAM_MAX_BLOCKS + blk] = sq;
}

// Where parameter i of the flat order W1 [256, in] | b1 | W2 [256, 256] | b2 | W3 [out, 256] | b3 lives in the packed
// stream (the inverse of k11_mlp.hip's pack_stream): every weight has one place per operand role, i.e. the 32-column
// stream (K11), the 16-column stream (K13 / K14 forward) and, for W2 / W3, the transposed-role stream (K14 backward).
__device__ __forceinline__ void scatter_packed(const PackLayout& L, float* __restrict__ out, long i, float p) {
  const long ob1 = (long)HID * L.in_dim, oW2 = ob1 + HID, ob2 = oW2 + (long)HID * HID, oW3 = ob2 + HID, ob3 = oW3 + (long)L.out_dim * HID;
  // P32[tile][g][lane][q] = W[n = 32 tile + (lane & 31)][k = 2 (4 g + q) + (lane >> 5)]
  auto e32 = [](size_t base, int groups, int n, int k) {
    const int kk = k >> 1;
    return base + (((size_t)((n >> 5) * groups + (kk >> 2)) * 64 + ((n & 31) + 32 * (k & 1))) * 4 + (kk & 3));
  };
  // P16[tile][g][lane][q] = W[n = 16 tile + (lane & 15)][k = 16 g + 4 q + (lane >> 4)]
  auto e16 = [](size_t base, int groups, int n, int k) {
    return base + (((size_t)((n >> 4) * groups + (k >> 4)) * 64 + ((n & 15) + 16 * (k & 3))) * 4 + ((k >> 2) & 3));
  };
  // PT[tile][g][lane][q] = W[n = 16 g + 4 q + (lane >> 4)][k = 16 tile + (lane & 15)]
  auto eT = [](size_t base, int groups, int n, int k) {
    return base + (((size_t)((k >> 4) * groups + (n >> 4)) * 64 + ((k & 15) + 16 * (n & 3))) * 4 + ((n >> 2) & 3));
  };
  if (i < ob1) {
    const int n = (int)(i / L.in_dim), k = (int)(i - (long)n * L.in_dim);
    out[e32(L.w1, L.g1, n, k)] = p;
    out[e16(L.w1n, G1N, n, k)] = p;
  } else if (i < oW2) {
    out[L.b1 + (i - ob1)] = p;
  } else if (i < ob2) {
    const int r = (int)(i - oW2), n = r >> 8, k = r & 255;
    out[e32(L.w2, 32, n, k)] = p;
    out[e16(L.w2n, HID / 16, n, k)] = p;
    out[eT(L.w2t, HID / 16, n, k)] = p;
  } else if (i < oW3) {
    out[L.b2 + (i - ob2)] = p;
  } else if (i < ob3) {
    const int r = (int)(i - oW3), n = r >> 8, k = r & 255;
    out[e32(L.w3, 32, n, k)] = p;            // one 32-column tile: n < 32
    out[e16(L.w3n, HID / 16, n, k)] = p;
    out[eT(L.w3t, T3N, n, k)] = p;
  } else {
    out[L.b3 + (i - ob3)] = p;
  }
}

// clip_grad_norm_: g *= min(max_norm / (||g|| + 1e-6), 1) per network; Adam (torch.optim.Adam, amsgrad off, no decay):
//   m += (g - m)(1 - b1); v = v b2 + ((1 - b2) g) g; p += -(lr / (1 - b1^t)) (m / (sqrt(v) / sqrt(1 - b2^t) + eps))
__global__ __launch_bounds__(256) void adam_step_kernel(AdamArgs a, int blocks_a) {
  const int n = (int)blockIdx.x >= blocks_a;          // a block (hence a wave) belongs to ONE network
  const AdamNet nt = a.net[n];
  const long i = 256L * ((int)blockIdx.x - (n ? blocks_a : 0)) + threadIdx.x;
  // the squared norm from the block partials: lane l adds partials l, l + 64, ... in order, then the 64-lane tree
  double ss = 0.0;
  for (int b = threadIdx.x & 63; b < nt.blocks; b += 64) ss += a.sumsq[n * ADAM_MAX_BLOCKS + b];
  ss = __shfl(wave_sum(ss), 0, 64);
  PackLayout L;
  if (nt.packed) {
    L = pack_layout(nt.in_dim, nt.out_dim);
    if (i < MAX_IN) {                                   // the stream's input-normalisation tables
      nt.packed[L.mean + i] = (nt.mean && i < nt.in_dim) ? nt.mean[i] : 0.f;
      nt.packed[L.std + i] = (nt.std && i < nt.in_dim) ? nt.std[i] : 1.f;
    }
  }
  if (i >= nt.n) return;
  const float norm = (float)sqrt(ss);
  const float coef = fminf(a.max_norm / (norm + 1e-6f), 1.0f);
  const float g = nt.grad[i] * coef;
  float m = nt.exp_avg[i], v = nt.exp_avg_sq[i];
  m = m + (g - m) * a.w1;
  v = v * a.beta2 + (a.w2 * g) * g;
  nt.exp_avg[i] = m;
  nt.exp_avg_sq[i] = v;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  const float p = nt.param[i] + a.neg_step * (m / denom);
  nt.param[i] = p;
  if (nt.packed) scatter_packed(L, nt.packed, i, p);
}

// input images 4 XI, hidden images 3 HI, partial tiles, mirror rows, dZ3 images, constants, statistics; + dW1: 16 KT1 KB
constexpr size_t UPD_LDS_BASE = sizeof(float) * (4 * XI + 3 * HI + 8 * UR * PP + 2 * UR * 16 + 512 + 128 + 640 + 512) + sizeof(double) * NSTAT * 64;
constexpr size_t upd_lds(int kt1) { return UPD_LDS_BASE + (size_t)16 * kt1 * 64 * sizeof(float4); }

inline int grad_floats(int in_dim, int out_dim) { return HID * in_dim + HID + HID * HID + HID + out_dim * HID + out_dim; }
inline int pad4(int n) { return (n + 3) & ~3; }

// 16-row tiles of a network's work: with the mirror loss an actor tile holds eight rows of the minibatch and their mirrors
inline int tiles_actor(int B, int mirror) { return mirror ? (B + 7) / 8 : (B + UR - 1) / UR; }
inline int tiles_critic(int B) { return (B + UR - 1) / UR; }

void choose_parts(const oly_ctx* ctx, int B, int mirror, int* pa, int* pc) {
  const int ta = tiles_actor(B, mirror), tc = tiles_critic(B);
  const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
  // a network's share of the workgroups follows its work: tiles x the measured cost of a tile (tools/time_k14.py, K cycles
  // per 16-row tile: critic 41.4, actor 45.4: its loss phase is longer)
  const double wa = 45.4 * ta, wc = 41.4 * tc;
  int a = (int)(cus * wa / (wa + wc) + 0.5);
  a = a < 1 ? 1 : (a > cus - 1 ? cus - 1 : a);
  int cpart = cus - a;
  *pa = ta < a ? ta : a;
  *pc = tc < cpart ? tc : cpart;
}

}  // namespace

extern "C" int oly_ppo_update_grad_floats(int in_dim, int hidden, int out_dim) {
  if (hidden != HID || in_dim <= 0 || in_dim > MAX_IN || out_dim <= 0 || out_dim > 16) return -1;
  return grad_floats(in_dim, out_dim);
}

extern "C" int64_t oly_ppo_update_ws_floats(oly_ctx* ctx, int B, int in_dim, int act_dim, int mirror, int32_t* parts_actor,
                                            int32_t* parts_critic) {
  if (!ctx || B <= 0 || in_dim <= 0 || in_dim > MAX_IN || act_dim <= 0 || act_dim > 16) return -1;
  int pa, pc;
  choose_parts(ctx, B, mirror, &pa, &pc);
  if (parts_actor) *parts_actor = pa;
  if (parts_critic) *parts_critic = pc;
  // partial gradients of both networks, then the statistics (NSTAT doubles per part)
  return (int64_t)pa * pad4(grad_floats(in_dim, act_dim)) + (int64_t)pc * pad4(grad_floats(in_dim, 1)) + 2 * (int64_t)NSTAT * (pa + pc) + 4;
}

extern "C" int oly_ppo_update_grads(oly_ctx* ctx, const oly_ppo_update* u, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!u) OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_update_grads: NULL argument block");
  if (u->B <= 0 || u->in_dim <= 0 || u->in_dim > MAX_IN || u->act_dim <= 0 || u->act_dim > 16)
    OLY_FAIL(ctx, OLY_ERANGE, "oly_ppo_update_grads: supported shape is in <= %d -> 256 -> 256 -> act <= 16, B > 0 (got B=%d in=%d act=%d)",
             MAX_IN, u->B, u->in_dim, u->act_dim);
  if (!u->obs || !u->action || !u->adv || !u->ret || !u->old_mu || !u->packed_actor || !u->packed_critic || !u->sd ||
      !u->log_sd || !u->old_sd || !u->old_log_sd || !u->grad_actor || !u->grad_critic || !u->scal_out || !u->ws)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_update_grads: NULL pointer");
  const int mirror = u->mir_obs != nullptr;
  if (mirror && (!u->act_src || !u->act_sign)) OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_update_grads: mir_obs without the action mirror table");
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!al16(u->packed_actor) || !al16(u->packed_critic) || !al16(u->ws))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_update_grads: packed streams and workspace must be 16-byte aligned");
  int pa = u->parts_actor, pc = u->parts_critic;
  const int nta = tiles_actor(u->B, mirror), ntc = tiles_critic(u->B);
  if (pa <= 0 || pc <= 0) choose_parts(ctx, u->B, mirror, &pa, &pc);
  if (pa > nta || pc > ntc) OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_update_grads: more parts (%d, %d) than 16-row tiles (%d, %d)", pa, pc, nta, ntc);
  const int gfa = grad_floats(u->in_dim, u->act_dim), gfc = grad_floats(u->in_dim, 1);
  const int psa = pad4(gfa), psc = pad4(gfc);
  const int64_t need = (int64_t)pa * psa + (int64_t)pc * psc + 2 * (int64_t)NSTAT * (pa + pc) + 4;
  if (u->ws_floats < need) OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_update_grads: workspace of %lld floats, %lld needed", (long long)u->ws_floats, (long long)need);
  UpdArgs a;
  a.B = u->B; a.in_dim = u->in_dim; a.act_dim = u->act_dim; a.pad_ = 0;
  a.obs = u->obs; a.mir_obs = u->mir_obs; a.action = u->action; a.adv = u->adv; a.ret = u->ret; a.old_mu = u->old_mu;
  a.idx = u->idx;
  float* ws = u->ws;
  a.net[0] = UpdNet{u->packed_actor, ws, u->act_dim, u->normalize_actor, pa, gfa, psa, nta};
  a.net[1] = UpdNet{u->packed_critic, ws + (size_t)pa * psa, 1, u->normalize_critic, pc, gfc, psc, ntc};
  size_t off = (size_t)pa * psa + (size_t)pc * psc;
  off = (off + 3) & ~(size_t)3;                    // doubles: 8-byte aligned (ws is 16-byte aligned)
  a.stat_partials = reinterpret_cast<double*>(ws + off);
  a.sd = u->sd; a.log_sd = u->log_sd; a.old_sd = u->old_sd; a.old_log_sd = u->old_log_sd;
  a.clip = u->clip; a.vf_coeff = u->vf_coeff; a.mirror_coeff = u->mirror_coeff;
  a.mirror_gscale = (float)(2.0 / ((double)u->B * u->act_dim));
  a.inv_b = 1.0f / (float)u->B;
  a.act_src = u->act_src; a.act_sign = u->act_sign;
#ifdef OLY_DIAG
  {   // tools/time_k14.py passes a workspace with 2 * 8 * 64 * 16 extra 8-byte slots at its end
    const int64_t slots = 2L * 8 * 64 * 16 * 2;
    a.stamps = (u->ws_floats >= need + slots) ? reinterpret_cast<unsigned long long*>(u->ws + ((u->ws_floats - slots) & ~(int64_t)1)) : nullptr;
  }
#endif
  const int kt1 = (u->in_dim + 15) / 16;
  const dim3 grid(pa + pc), block(UT);
#define OLY_UPD_LAUNCH(K)                                                                                              \
  do {                                                                                                                 \
    if (!(ctx->upd_attr_done & (1u << K))) {                                                                           \
      OLY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_update_kernel<K>),                            \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)upd_lds(K)));                     \
      ctx->upd_attr_done |= 1u << K;                                                                                   \
    }                                                                                                                  \
    hipLaunchKernelGGL(ppo_update_kernel<K>, grid, block, upd_lds(K), oly_s(stream), a);                                  \
  } while (0)
  switch (kt1) {
    case 1: OLY_UPD_LAUNCH(1); break;
    case 2: OLY_UPD_LAUNCH(2); break;
    case 3: OLY_UPD_LAUNCH(3); break;
    default: OLY_UPD_LAUNCH(4); break;
  }
#undef OLY_UPD_LAUNCH
  OLY_LAUNCH_CHECK(ctx, "ppo_update_kernel");
  FinArgs f;
  f.net[0] = a.net[0]; f.net[1] = a.net[1];
  f.grad[0] = u->grad_actor; f.grad[1] = u->grad_critic;
  f.stat_partials = a.stat_partials;
  f.scal_out = u->scal_out;
  f.gnorm = u->gnorm_ws;
  f.log_sd = u->log_sd;
  f.B = u->B; f.act_dim = u->act_dim; f.mirror = mirror;
  f.vf_coeff = u->vf_coeff;
  const int fba = (gfa + 255) / 256, fbc = (gfc + 255) / 256;
  hipLaunchKernelGGL(ppo_update_finish_kernel, dim3(fba + fbc), dim3(256), 0, oly_s(stream), f, fba);
  OLY_LAUNCH_CHECK(ctx, "ppo_update_finish_kernel");
  return OLY_OK;
}

extern "C" int oly_ppo_adam_step(oly_ctx* ctx, const oly_ppo_adam* a, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!a) OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_adam_step: NULL argument block");
  if (a->in_dim <= 0 || a->in_dim > MAX_IN || a->step <= 0 || !a->ws)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_adam_step: bad in_dim / step / workspace");
  if (!(a->beta1 >= 0.f && a->beta1 < 1.f && a->beta2 >= 0.f && a->beta2 < 1.f) || !(a->lr >= 0.f) || !(a->eps >= 0.f))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_adam_step: bad hyper-parameters");
  AdamArgs k;
  for (int n = 0; n < 2; ++n) {
    const oly_adam_net& s = a->net[n];
    if (!s.param || !s.grad || !s.exp_avg || !s.exp_avg_sq || s.out_dim <= 0 || s.out_dim > 16)
      OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_adam_step: network %d: NULL buffer or bad out_dim", n);
    const int gf = grad_floats(a->in_dim, s.out_dim);
    k.net[n] = AdamNet{s.param, s.exp_avg, s.exp_avg_sq, s.grad, gf, (gf + 255) / 256, s.packed, s.in_mean, s.in_std, a->in_dim, s.out_dim};
    if (k.net[n].blocks > ADAM_MAX_BLOCKS) OLY_FAIL(ctx, OLY_ERANGE, "oly_ppo_adam_step: network too large");
  }
  k.sumsq = a->ws;
  // the step-dependent scalars in fp64 on the host (torch's non-capturable Adam forms them as python floats)
  const double bc1 = 1.0 - pow((double)a->beta1, (double)a->step), bc2 = 1.0 - pow((double)a->beta2, (double)a->step);
  k.w1 = (float)(1.0 - (double)a->beta1);
  k.beta2 = a->beta2;
  k.w2 = (float)(1.0 - (double)a->beta2);
  k.eps = a->eps;
  k.neg_step = (float)(-((double)a->lr / bc1));
  k.bc2_sqrt = (float)sqrt(bc2);
  k.max_norm = a->max_grad_norm;
  if (!a->norm_ready) {       // the finishing launch of oly_ppo_update_grads leaves the same partials (gnorm_ws)
    const int wa = (k.net[0].blocks + 3) / 4, wc = (k.net[1].blocks + 3) / 4;
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(wa + wc), dim3(256), 0, oly_s(stream), k, wa);
  }
  const int ba = (k.net[0].n + 255) / 256, bc = (k.net[1].n + 255) / 256;
  hipLaunchKernelGGL(adam_step_kernel, dim3(ba + bc), dim3(256), 0, oly_s(stream), k, ba);
  OLY_LAUNCH_CHECK(ctx, "adam step kernels");
  return OLY_OK;
}

// One epoch's minibatch loop (ppo.py:357-410) as ONE call: BatchSampler cuts the permutation into consecutive minibatches
// of u->B rows; each gets update_policy's gradients and one optimiser step.  3 launches per minibatch, queued from C: at the
// reference's minibatch size (64) the Python side of one call per launch group cost more than the kernels ran.
extern "C" int oly_ppo_update_epoch(oly_ctx* ctx, const oly_ppo_update* u, const oly_ppo_adam* a, const int32_t* perm,
                                    int n_batches, double* scal_out, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!u || !a || !perm || !scal_out || n_batches < 0)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_update_epoch: NULL argument or negative n_batches");
  if (!u->gnorm_ws || u->gnorm_ws != a->ws)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_update_epoch: u->gnorm_ws must be a->ws (the gradient norm travels through it)");
  for (int n = 0; n < 2; ++n)
    if (a->net[n].grad != (n ? u->grad_critic : u->grad_actor) || a->net[n].packed != (n ? u->packed_critic : u->packed_actor))
      OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_update_epoch: network %d: the optimiser's grad / packed buffers are not the update's", n);
  if (a->step <= 0 || (long)a->step + n_batches > 0x7fffffffL) OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_update_epoch: bad step");
  oly_ppo_update uu = *u;
  oly_ppo_adam aa = *a;
  aa.norm_ready = 1;
  for (int b = 0; b < n_batches; ++b) {
    uu.idx = perm + (size_t)b * (size_t)uu.B;
    uu.scal_out = scal_out + 6 * (size_t)b;
    int rc = oly_ppo_update_grads(ctx, &uu, stream);
    if (rc != OLY_OK) return rc;
    aa.step = a->step + b;
    rc = oly_ppo_adam_step(ctx, &aa, stream);
    if (rc != OLY_OK) return rc;
  }
  return OLY_OK;
}
