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
#include "oly_common.h"
#include "mlp_tiles.h"

using namespace oly_mlp;
namespace {
// B operand of v_mfma_f32_32x32x2_f32: lane l holds B[k = l >> 5][n = l & 31].  Packed so that one
// 16-byte load per lane feeds four consecutive k-steps of one 32-column tile:
//   P[tile][group g][lane][q]  =  W[n = 32 tile + (lane & 31)][k = 2 (4 g + q) + (lane >> 5)]
__global__ void mlp_pack_kernel(PackLayout L, const float* __restrict__ W1, const float* __restrict__ B1,
                                const float* __restrict__ W2, const float* __restrict__ B2,
                                const float* __restrict__ W3, const float* __restrict__ B3,
                                const float* __restrict__ mean, const float* __restrict__ std,
                                float* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < L.total; e += stride) {
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
  dim3 grid((N + RT - 1) / RT, packed_b ? 2 : 1);
  hipLaunchKernelGGL(mlp_forward_kernel, grid, dim3(THREADS), MLP_LDS, oly_s(stream), a);
  OLY_LAUNCH_CHECK(ctx, "mlp_forward_kernel");
  return OLY_OK;
}
