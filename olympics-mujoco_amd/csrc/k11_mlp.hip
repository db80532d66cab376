// K11: forward pass of the rollout's two ReLU MLPs (actor mean 41 -> 256 -> 256 -> 12 and critic
// 41 -> 256 -> 256 -> 1: rl/policies/actor.py:142-195, critic.py:37-74) as ONE launch on the f32
// matrix cores.
//
// Why: in the per-vec-step regime (N = 4096 rows per forward) the library path is six GEMM launches
// of ~10 us each plus four ReLU launches (profiles/r02: 99 us per actor + critic forward, replayed
// from a graph); the arithmetic is 1.3 GFLOP = 8 us at the f32 MFMA rate.  Here a workgroup (8 waves) owns
// 32 rows of ONE network and carries them through all three layers: activations never leave LDS,
// weights stream from L2 in the MFMA B-operand layout (packed once per policy update by
// mlp_pack_kernel), 128 row tiles x 2 networks = 256 workgroups = one per CU.
//
// Numerics: v_mfma_f32_32x32x2_f32 is exact f32 (one rounding per product, k ascending), so
//   y = b + sum_k x_k * w_k  is the f32 fma chain  fma(x_k, w_k, acc)  in k order, starting from 0,
// with the bias added after the chain; layer 3 splits k over the eight waves (k in [32w, 32w+32))
// and adds the eight partial chains in wave order, then the bias.  Deterministic, and restated
// bit for bit by the oracle (oly_mlp_forward_cpu); against torch's own fp32 Linear the difference is
// summation order only (<= 1e-5 relative on these layers).
#include <cstdlib>

#include "oly_common.h"
#include "mlp_tiles.h"

using namespace oly_mlp;
namespace {
// B operand of v_mfma_f32_32x32x2_f32: lane l holds B[k = l >> 5][n = l & 31].  Packed so that one
// 16-byte load per lane feeds four consecutive k-steps of one 32-column tile:
//   P[tile][group g][lane][q]  =  W[n = 32 tile + (lane & 31)][k = 2 (4 g + q) + (lane >> 5)]
struct PackSrc {
  PackLayout L;
  const float *W1, *B1, *W2, *B2, *W3, *B3, *mean, *std;
  float* out;
};

__device__ __forceinline__ void pack_stream(const PackSrc& p, size_t e0, size_t stride) {
  const PackLayout& L = p.L;
  const float *W1 = p.W1, *B1 = p.B1, *W2 = p.W2, *B2 = p.B2, *W3 = p.W3, *B3 = p.B3, *mean = p.mean, *std = p.std;
  float* out = p.out;
  for (size_t e = e0; e < L.total; e += stride) {
    float v = 0.f;
    if (e < L.b1) {
      const size_t r = e - L.w1;
      const int q = r & 3, lane = (r >> 2) & 63;
      const int g = (int)((r >> 8) % L.g1), tile = (int)((r >> 8) / L.g1);
      const int k = 2 * (4 * g + q) + (lane >> 5), n = 32 * tile + (lane & 31);
      v = k < L.in_dim ? W1[(size_t)n * L.in_dim + k] : 0.f;
    } else if (e < L.w2) {
      v = B1[e - L.b1];
    } else if (e < L.b2) {
      const size_t r = e - L.w2;
      const int q = r & 3, lane = (r >> 2) & 63;
      const int g = (int)((r >> 8) & 31), tile = (int)(r >> 13);
      const int k = 2 * (4 * g + q) + (lane >> 5), n = 32 * tile + (lane & 31);
      v = W2[(size_t)n * HID + k];
    } else if (e < L.w3) {
      v = B2[e - L.b2];
    } else if (e < L.b3) {
      const size_t r = e - L.w3;
      const int q = r & 3, lane = (r >> 2) & 63;
      const int g = (int)(r >> 8);
      const int k = 2 * (4 * g + q) + (lane >> 5), n = lane & 31;
      v = n < L.out_dim ? W3[(size_t)n * HID + k] : 0.f;
    } else if (e < L.mean) {
      const int n = (int)(e - L.b3);
      v = n < L.out_dim ? B3[n] : 0.f;
    } else if (e < L.std) {
      const int k = (int)(e - L.mean);
      v = (mean && k < L.in_dim) ? mean[k] : 0.f;
    } else if (e < L.w1n) {
      const int k = (int)(e - L.std);
      v = (std && k < L.in_dim) ? std[k] : 1.f;
    } else if (e >= L.w2t) {
      // the transposed-role streams of K14: PT[tile][group][lane][q] = W[n = 16 group + 4 q + (lane >> 4)][k = 16 tile + (lane & 15)]
      const bool l2 = e < L.w3t;
      const size_t r = e - (l2 ? L.w2t : L.w3t);
      const int groups = l2 ? HID / 16 : T3N;
      const int q = r & 3, lane = (r >> 2) & 63;
      const int g = (int)((r >> 8) % groups), tile = (int)((r >> 8) / groups);
      const int n = 16 * g + 4 * q + (lane >> 4), k = 16 * tile + (lane & 15);
      if (l2) v = W2[(size_t)n * HID + k];
      else v = n < L.out_dim ? W3[(size_t)n * HID + k] : 0.f;
    } else {
      // the 16-column-tile streams: P16[tile][group][lane][q] = W[16 tile + (lane & 15)][16 group + 4 q + (lane >> 4)]
      const bool l1 = e < L.w2n, l2 = !l1 && e < L.w3n;
      const size_t r = e - (l1 ? L.w1n : l2 ? L.w2n : L.w3n);
      const int groups = l1 ? G1N : HID / 16;
      const int q = r & 3, lane = (r >> 2) & 63;
      const int g = (int)((r >> 8) % groups), tile = (int)((r >> 8) / groups);
      const int k = 16 * g + 4 * q + (lane >> 4), n = 16 * tile + (lane & 15);
      if (l1) v = k < L.in_dim ? W1[(size_t)n * L.in_dim + k] : 0.f;
      else if (l2) v = W2[(size_t)n * HID + k];
      else v = n < L.out_dim ? W3[(size_t)n * HID + k] : 0.f;
    }
    out[e] = v;
  }
}

__global__ void mlp_pack_kernel(PackLayout L, const float* __restrict__ W1, const float* __restrict__ B1,
                                const float* __restrict__ W2, const float* __restrict__ B2,
                                const float* __restrict__ W3, const float* __restrict__ B3,
                                const float* __restrict__ mean, const float* __restrict__ std,
                                float* __restrict__ out) {
  const PackSrc p{L, W1, B1, W2, B2, W3, B3, mean, std, out};
  pack_stream(p, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}


struct MlpNet {
  const float* packed;
  float* y;
  int out_dim, normalize;
};
struct MlpArgs {
  int N, in_dim;
  const float* x;
  MlpNet net[2];
};

// 8 waves: wave w owns output columns [32 w, 32 w + 32) of the hidden layers (two waves per SIMD keep the
// matrix pipe fed across each other's LDS / L2 waits) and k in [32 w, 32 w + 32) of the output layer.
__global__ __launch_bounds__(THREADS) void mlp_forward_kernel(MlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xT = lds;                    // [MAX_IN][LDP]   layer-1 input, k-major
  float* hA = xT + MAX_IN * LDP;      // [HID][LDP]      layer-1 output; later the layer-3 partials
  float* hB = hA + HID * LDP;         // [HID][LDP]      layer-2 output
  const MlpNet net = p.net[blockIdx.y];
  const PackLayout L = pack_layout(p.in_dim, net.out_dim);
  const float* __restrict__ P = net.packed;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * RT;
  const int rows = min(RT, p.N - row0);
  const int in_dim = p.in_dim;

  // ---- stage the input rows, normalised, transposed, zero-padded
  for (int e = tid; e < MAX_IN * RT; e += THREADS) {
    const int m = e / MAX_IN, k = e - m * MAX_IN;      // consecutive threads: consecutive k of one row
    float v = 0.f;
    if (m < rows && k < in_dim) {
      v = p.x[(size_t)(row0 + m) * in_dim + k];
      if (net.normalize) v = (v - P[L.mean + k]) / P[L.std + k];
    }
    xT[k * LDP + m] = v;
  }
  __syncthreads();

  const float4* P4 = reinterpret_cast<const float4*>(P);
  {  // ---- layer 1: [32, in <= 64] x [64, 256] (k zero-padded to 64)
    f32x16 acc = {0};
    layer_tile<G1>(xT, P4 + (L.w1 >> 2) + (size_t)wave * G1 * 64, lane, acc);
    store_relu(acc, P + L.b1, 32 * wave, lane, hA);
  }
  __syncthreads();
  {  // ---- layer 2: [32, 256] x [256, 256]
    f32x16 acc = {0};
    layer_tile<32>(hA, P4 + (L.w2 >> 2) + (size_t)wave * 32 * 64, lane, acc);
    store_relu(acc, P + L.b2, 32 * wave, lane, hB);
  }
  __syncthreads();
  {  // ---- layer 3: [32, 256] x [256, out <= 32]; wave w owns k in [32 w, 32 w + 32)
    f32x16 acc = {0};
    layer_tile<4>(hB + (size_t)(32 * wave) * LDP, P4 + (L.w3 >> 2) + (size_t)(4 * wave) * 64, lane, acc);
    const int r = lane & 31, h = lane >> 5;
    float* part = hA + (size_t)wave * RT * LDP;        // [row][col] partial of this wave
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      part[row * LDP + r] = acc[i];
    }
  }
  __syncthreads();
  const int out_dim = net.out_dim;
  for (int e = tid; e < rows * out_dim; e += THREADS) {
    const int m = e / out_dim, c = e - m * out_dim;
    float s = hA[m * LDP + c];
#pragma unroll
    for (int w = 1; w < KSPLIT; ++w) s += hA[(w * RT + m) * LDP + c];
    s += P[L.b3 + c];
    net.y[(size_t)(row0 + m) * out_dim + c] = s;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The same forward for SMALL batches, 16 rows per workgroup on v_mfma_f32_16x16x4_f32 (the tiles K13 runs inside
// its step loop, mlp_tiles.h).  4096 rows x 2 networks are 256 of the 32-row tiles: one 8-wave workgroup per CU, a
// chain of 176 dependent 64-cycle MFMAs per wave with nothing beside it to hide the input load, the barriers and
// the epilogue.  As 512 four-wave workgroups of 16 rows, two per CU, a tile's chain is half as long (320 MFMAs of
// 32 cycles on four independent accumulators) and one workgroup's staging overlaps the other's layers.  Same
// arithmetic value for value: k-ascending fma chains, the output layer as the same eight partial chains over
// k in [32 j, 32 j + 32) added in order j, bias last (wave w runs chains 2 w and 2 w + 1).
constexpr int RT16 = 16, THREADS16 = 256, PP16 = MAX_OUT + 1;
static_assert(KSPLIT * RT16 * PP16 <= (MAX_IN + HID) * RT16, "the output layer's partial tiles alias the input and layer-1 images");

template <int G1>      // groups of 16 inputs in layer 1
__global__ __launch_bounds__(THREADS16, 2) void mlp_forward16_kernel(MlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xT = lds;                     // [MAX_IN x 16]  input image (act16 layout); with hA later the partial tiles
  float* hA = xT + MAX_IN * RT16;      // [HID x 16]     layer-1 image
  float* hB = hA + HID * RT16;         // [HID x 16]     layer-2 image
  const MlpNet net = p.net[blockIdx.y];
  const PackLayout L = pack_layout(p.in_dim, net.out_dim);
  const float* __restrict__ P = net.packed;
  const float4* P4 = reinterpret_cast<const float4*>(P);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * RT16;
  const int rows = min(RT16, p.N - row0);
  const int in_dim = p.in_dim;

  // ---- stage the input rows, normalised, zero-padded, as the A-operand image
  for (int e = tid; e < MAX_IN * RT16; e += THREADS16) {
    const int m = e / MAX_IN, k = e - m * MAX_IN;      // consecutive threads: consecutive k of one row
    float v = 0.f;
    if (m < rows && k < in_dim) {
      v = p.x[(size_t)(row0 + m) * in_dim + k];
      if (net.normalize) v = (v - P[L.mean + k]) / P[L.std + k];
    }
    xT[act16_index(k, m)] = v;
  }
  // this wave's hidden-layer biases, requested before the first barrier
  float bias1[4], bias2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    bias1[t] = P[L.b1 + 16 * (4 * wave + t) + (lane & 15)];
    bias2[t] = P[L.b2 + 16 * (4 * wave + t) + (lane & 15)];
  }
  __syncthreads();
  {  // ---- layer 1: [16, in <= 64] x [in, 256]; wave w owns column tiles 4 w .. 4 w + 3
    f32x4 acc[4] = {{0}, {0}, {0}, {0}};
    const float4* base = P4 + (L.w1n >> 2) + (size_t)(4 * wave) * G1N * 64;
    const float4* const w[4] = {base, base + G1N * 64, base + 2 * G1N * 64, base + 3 * G1N * 64};
    layer_tiles16<G1, 4>(reinterpret_cast<const float4*>(xT), w, lane, acc);
#pragma unroll
    for (int t = 0; t < 4; ++t) store_relu16v(acc[t], bias1[t], 4 * wave + t, lane, hA);
  }
  __syncthreads();
  {  // ---- layer 2: [16, 256] x [256, 256]
    f32x4 acc[4] = {{0}, {0}, {0}, {0}};
    const float4* base = P4 + (L.w2n >> 2) + (size_t)(4 * wave) * (HID / 16) * 64;
    const float4* const w[4] = {base, base + (HID / 16) * 64, base + 2 * (HID / 16) * 64, base + 3 * (HID / 16) * 64};
    layer_tiles16<HID / 16, 4>(reinterpret_cast<const float4*>(hA), w, lane, acc);
#pragma unroll
    for (int t = 0; t < 4; ++t) store_relu16v(acc[t], bias2[t], 4 * wave + t, lane, hB);
  }
  __syncthreads();
  {  // ---- output layer: [16, 256] x [256, out <= 32] as eight partial chains; wave w runs chains 2 w, 2 w + 1
    const int c = lane & 15, h2 = lane >> 4;
    const bool two = net.out_dim > 16;           // a second 16-column tile
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * wave + jj;
      const float4* a4 = reinterpret_cast<const float4*>(hB) + (size_t)(2 * j) * 64;
      float* part = xT + (size_t)j * RT16 * PP16;                 // [row][col] partial of chain j
      if (two) {
        f32x4 acc[2] = {{0}, {0}};
        const float4* const w[2] = {P4 + (L.w3n >> 2) + (size_t)(2 * j) * 64,
                                    P4 + (L.w3n >> 2) + (size_t)(HID / 16 + 2 * j) * 64};
        layer_tiles16<2, 2>(a4, w, lane, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          part[(4 * h2 + i) * PP16 + c] = acc[0][i];
          part[(4 * h2 + i) * PP16 + 16 + c] = acc[1][i];
        }
      } else {
        f32x4 acc[1] = {{0}};
        const float4* const w[1] = {P4 + (L.w3n >> 2) + (size_t)(2 * j) * 64};
        layer_tiles16<2, 1>(a4, w, lane, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) part[(4 * h2 + i) * PP16 + c] = acc[0][i];
      }
    }
  }
  __syncthreads();
  const int out_dim = net.out_dim;
  for (int e = tid; e < rows * out_dim; e += THREADS16) {
    const int m = e / out_dim, c = e - m * out_dim;
    float s = xT[m * PP16 + c];
#pragma unroll
    for (int w = 1; w < KSPLIT; ++w) s += xT[(w * RT16 + m) * PP16 + c];
    s += P[L.b3 + c];
    net.y[(size_t)(row0 + m) * out_dim + c] = s;
  }
}

constexpr size_t MLP16_LDS = sizeof(float) * (MAX_IN + 2 * HID) * RT16;
constexpr size_t MLP_LDS = sizeof(float) * (MAX_IN + 2 * HID) * LDP;
}  // namespace

extern "C" int64_t oly_mlp_packed_floats(int in_dim, int hidden, int out_dim) {
  if (hidden != HID || in_dim <= 0 || in_dim > MAX_IN || out_dim <= 0 || out_dim > MAX_OUT) return -1;
  return (int64_t)pack_layout(in_dim, out_dim).total;
}

extern "C" int oly_mlp_pack(oly_ctx* ctx, int in_dim, int hidden, int out_dim, const float* w1, const float* b1,
                            const float* w2, const float* b2, const float* w3, const float* b3,
                            const float* in_mean, const float* in_std, float* packed, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (hidden != HID || in_dim <= 0 || in_dim > MAX_IN || out_dim <= 0 || out_dim > MAX_OUT)
    OLY_FAIL(ctx, OLY_ERANGE, "oly_mlp_pack: supported shape is in <= %d -> %d -> %d -> out <= %d (got %d, %d, %d)",
             MAX_IN, HID, HID, MAX_OUT, in_dim, hidden, out_dim);
  if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !packed) OLY_FAIL(ctx, OLY_EINVAL, "oly_mlp_pack: NULL pointer");
  if ((reinterpret_cast<uintptr_t>(packed) & 15) != 0) OLY_FAIL(ctx, OLY_EINVAL, "oly_mlp_pack: packed must be 16-byte aligned");
  const PackLayout L = pack_layout(in_dim, out_dim);
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(128), dim3(256), 0, oly_s(stream), L, w1, b1, w2, b2, w3, b3, in_mean,
                     in_std, packed);
  OLY_LAUNCH_CHECK(ctx, "mlp_pack_kernel");
  return OLY_OK;
}

extern "C" int oly_mlp_forward2(oly_ctx* ctx, int N, int in_dim, const float* x, const float* packed_a, int out_a,
                                int normalize_a, float* y_a, const float* packed_b, int out_b, int normalize_b,
                                float* y_b, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (N < 0 || in_dim <= 0 || in_dim > MAX_IN) OLY_FAIL(ctx, OLY_EINVAL, "oly_mlp_forward2: bad N / in_dim");
  if (N == 0) return OLY_OK;
  if (!x || !packed_a || !y_a || out_a <= 0 || out_a > MAX_OUT || (packed_b && (!y_b || out_b <= 0 || out_b > MAX_OUT)))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_mlp_forward2: bad argument");
  if (!ctx->mlp_attr_done) {
    OLY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_forward_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)MLP_LDS));
    ctx->mlp_attr_done = true;
  }
  MlpArgs a;
  a.N = N;
  a.in_dim = in_dim;
  a.x = x;
  a.net[0] = MlpNet{packed_a, y_a, out_a, normalize_a};
  a.net[1] = MlpNet{packed_b, y_b, out_b, normalize_b};
  // 16-row tiles while the 32-row tiles would leave CUs without a second workgroup (OLY_K11_ROWS = 16 / 32 forces one)
  static const int force_rows = [] { const char* e = getenv("OLY_K11_ROWS"); return e ? atoi(e) : 0; }();
  const int nets = packed_b ? 2 : 1;
  const long slots = 2L * (ctx->num_cu > 0 ? ctx->num_cu : 256);
  const bool rows16 = force_rows == 16 || (force_rows != 32 && (long)((N + RT - 1) / RT) * nets < slots);
  if (rows16) {
    dim3 grid16((N + RT16 - 1) / RT16, nets);
    switch ((in_dim + 15) / 16) {
      case 1: hipLaunchKernelGGL(mlp_forward16_kernel<1>, grid16, dim3(THREADS16), MLP16_LDS, oly_s(stream), a); break;
      case 2: hipLaunchKernelGGL(mlp_forward16_kernel<2>, grid16, dim3(THREADS16), MLP16_LDS, oly_s(stream), a); break;
      case 3: hipLaunchKernelGGL(mlp_forward16_kernel<3>, grid16, dim3(THREADS16), MLP16_LDS, oly_s(stream), a); break;
      default: hipLaunchKernelGGL(mlp_forward16_kernel<4>, grid16, dim3(THREADS16), MLP16_LDS, oly_s(stream), a); break;
    }
    OLY_LAUNCH_CHECK(ctx, "mlp_forward16_kernel");
    return OLY_OK;
  }
  dim3 grid((N + RT - 1) / RT, nets);
  hipLaunchKernelGGL(mlp_forward_kernel, grid, dim3(THREADS), MLP_LDS, oly_s(stream), a);
  OLY_LAUNCH_CHECK(ctx, "mlp_forward_kernel");
  return OLY_OK;
}
