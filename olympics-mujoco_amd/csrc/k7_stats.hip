// K7: statistics + normalisation.
//   oly_adv_stats / oly_adv_normalize   rl/algos/ppo.py:335-336, gail_TRPO.py:128
//   oly_col_stats                       Standardizer.update_mean_std networks.py:76-81,
//                                       get_normalization_params / RunningMeanStd
//                                       rl/envs/normalize.py:35-48,182-208
// Sums are accumulated in fp64 and combined in a FIXED order (per-thread -> wave shuffle ->
// LDS -> per-block partial -> one finishing block): deterministic run to run, no float
// atomics.  stats live in device memory so that the multi-GPU path can all-gather them over
// RCCL without a host round trip.  Bound: HBM, 4 B/element (stats), 8 B/element (normalise).
#include "oly_common.h"

namespace {

constexpr int THREADS = 256;

__global__ __launch_bounds__(THREADS) void stats_partial_kernel(long n, const float* __restrict__ x,
                                                                double* __restrict__ ws) {
  __shared__ double sh[2 * (THREADS / 64)];
  double s = 0.0, ss = 0.0;
  const long n4 = n >> 2;
  const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  const long stride = (long)gridDim.x * THREADS;
  if (vec) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long i = (long)blockIdx.x * THREADS + threadIdx.x; i < n4; i += stride) {
      const float4 v = x4[i];
      const double a = v.x, b = v.y, c = v.z, d = v.w;
      s += a; ss += a * a;
      s += b; ss += b * b;
      s += c; ss += c * c;
      s += d; ss += d * d;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride) {
      const double a = x[i];
      s += a; ss += a * a;
    }
  } else {
    for (long i = (long)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride) {
      const double a = x[i];
      s += a; ss += a * a;
    }
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[2 * w] = s; sh[2 * w + 1] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tss = 0.0;
    for (int i = 0; i < THREADS / 64; ++i) { ts += sh[2 * i]; tss += sh[2 * i + 1]; }
    ws[2 * blockIdx.x] = ts;
    ws[2 * blockIdx.x + 1] = tss;
  }
}

__global__ __launch_bounds__(64) void stats_finish_kernel(int nblocks, long n,
                                                          const double* __restrict__ ws,
                                                          double* __restrict__ out) {
  // lane i sums partials i, i+64, ... in order; lanes are then combined by the fixed tree
  double s = 0.0, ss = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) { s += ws[2 * i]; ss += ws[2 * i + 1]; }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if (threadIdx.x == 0) { out[0] = (double)n; out[1] = s; out[2] = ss; }
}

// st: [parts][3] (count, sum, sumsq) triples, one per rank; combined by a balanced pairwise tree
// in rank order ((r0 + r1) + (r2 + r3)) + ... : the same on every rank, and equal to what one rank
// holding all shards as aligned sub-trees would form.
// Lane i of a wave holds part i (zero beyond `parts`): `v += shfl_down(v, h)` for h = 1, 2, 4, ... leaves exactly that
// tree in lane 0 (the zero padding adds exact zeros); every wave computes it for itself.  (An indexed local array of
// OLY_MAX_STAT_PARTS doubles went to scratch memory: 528 B per lane for a kernel of 28 registers.)
static_assert(OLY_MAX_STAT_PARTS <= 64, "one wave holds all parts");
__device__ __forceinline__ void combine_parts(const double* __restrict__ st, int parts, double out[3]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double v = lane < parts ? st[3 * lane + k] : 0.0;
#pragma unroll
    for (int h = 1; h < 64; h <<= 1) {
      const double o = __shfl_down(v, h, 64);
      if ((lane & (2 * h - 1)) == 0) v += o;
    }
    out[k] = __shfl(v, 0, 64);
  }
}

__global__ __launch_bounds__(THREADS) void normalize_kernel(long n, float* __restrict__ x,
                                                            const double* __restrict__ st_, int parts,
                                                            int ddof, double eps) {
  double st[3];
  if (parts == 1) { st[0] = st_[0]; st[1] = st_[1]; st[2] = st_[2]; }
  else combine_parts(st_, parts, st);
  const double cnt = st[0], mean = st[1] / cnt;
  double var = (st[2] - cnt * mean * mean) / (cnt - (double)ddof);
  if (var < 0.0) var = 0.0;
  const double denom = sqrt(var) + eps;
  const long stride = (long)gridDim.x * THREADS;
  const long n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? (n >> 2) : 0;
  float4* x4 = reinterpret_cast<float4*>(x);
  for (long i = (long)blockIdx.x * THREADS + threadIdx.x; i < n4; i += stride) {
    float4 v = x4[i];
    v.x = (float)(((double)v.x - mean) / denom);
    v.y = (float)(((double)v.y - mean) / denom);
    v.z = (float)(((double)v.z - mean) / denom);
    v.w = (float)(((double)v.w - mean) / denom);
    x4[i] = v;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride)
    x[i] = (float)(((double)x[i] - mean) / denom);
}

// Column sums of x [B,D]: TPB = (256 / D) * D threads; thread t owns column t % D and rows
// t / D, t / D + rpb, ...; a block owns a contiguous slab of rows.
__global__ __launch_bounds__(THREADS) void col_partial_kernel(int B, int D, int rows_per_block,
                                                              const float* __restrict__ x,
                                                              double* __restrict__ ws) {
  extern __shared__ double shc[];  // [lanes_per_col][D][2]
  const int lanes = blockDim.x / D;  // rows processed concurrently
  const int c = threadIdx.x % D, rl = threadIdx.x / D;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(B, r0 + rows_per_block);
  double s = 0.0, ss = 0.0;
  for (int r = r0 + rl; r < r1; r += lanes) {
    const double a = x[(size_t)r * D + c];
    s += a; ss += a * a;
  }
  shc[(rl * D + c) * 2] = s;
  shc[(rl * D + c) * 2 + 1] = ss;
  __syncthreads();
  if (rl == 0) {
    for (int l = 1; l < lanes; ++l) { s += shc[(l * D + c) * 2]; ss += shc[(l * D + c) * 2 + 1]; }
    ws[((size_t)blockIdx.x * D + c) * 2] = s;
    ws[((size_t)blockIdx.x * D + c) * 2 + 1] = ss;
  }
}

// D % 4 == 0 and 16-B aligned rows: a thread owns FOUR adjacent columns (one 16-B load per
// row) and rows t / (D/4) + k * lanes; two rows in flight per iteration.
__global__ __launch_bounds__(THREADS) void col_partial4_kernel(int B, int D, int rows_per_block,
                                                               const float* __restrict__ x,
                                                               double* __restrict__ ws) {
  extern __shared__ double shc[];  // [lanes][D][2]
  const int D4 = D >> 2;
  const int lanes = blockDim.x / D4;
  const int c4 = threadIdx.x % D4, rl = threadIdx.x / D4;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(B, r0 + rows_per_block);
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  const float4* x4 = reinterpret_cast<const float4*>(x);
  auto acc = [&](const float4& v) {
    const double a = v.x, b = v.y, c = v.z, d = v.w;
    s[0] += a; ss[0] += a * a;
    s[1] += b; ss[1] += b * b;
    s[2] += c; ss[2] += c * c;
    s[3] += d; ss[3] += d * d;
  };
  int r = r0 + rl;
  for (; r + lanes < r1; r += 2 * lanes) {
    const float4 v0 = x4[(size_t)r * D4 + c4];
    const float4 v1 = x4[(size_t)(r + lanes) * D4 + c4];
    acc(v0);
    acc(v1);
  }
  if (r < r1) acc(x4[(size_t)r * D4 + c4]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    shc[(rl * D + 4 * c4 + k) * 2] = s[k];
    shc[(rl * D + 4 * c4 + k) * 2 + 1] = ss[k];
  }
  __syncthreads();
  if (threadIdx.x < D) {
    const int c = threadIdx.x;
    double ts = 0.0, tss = 0.0;
    for (int l = 0; l < lanes; ++l) { ts += shc[(l * D + c) * 2]; tss += shc[(l * D + c) * 2 + 1]; }
    ws[((size_t)blockIdx.x * D + c) * 2] = ts;
    ws[((size_t)blockIdx.x * D + c) * 2 + 1] = tss;
  }
}

// one wave per column: lane i sums partials i, i+64, ... in order, then the fixed shuffle tree
__global__ __launch_bounds__(64) void col_finish_kernel(int nblocks, int B, int D,
                                                        const double* __restrict__ ws,
                                                        double* __restrict__ colstats, int accumulate) {
  const int c = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) {
    s += ws[((size_t)b * D + c) * 2];
    ss += ws[((size_t)b * D + c) * 2 + 1];
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if (threadIdx.x != 0) return;
  if (accumulate) {
    colstats[c] += (double)B; colstats[D + c] += s; colstats[2 * D + c] += ss;
  } else {
    colstats[c] = (double)B; colstats[D + c] = s; colstats[2 * D + c] = ss;
  }
}

}  // namespace

extern "C" int oly_adv_stats(oly_ctx* ctx, int64_t n, const float* x, double* stats3_out,
                             oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (n < 0 || !stats3_out || (n > 0 && !x)) OLY_FAIL(ctx, OLY_EINVAL, "oly_adv_stats: bad argument");
  long want = (n / 4 + THREADS - 1) / THREADS;
  int nb = (int)(want < 1 ? 1 : (want > OLY_STATS_MAX_BLOCKS ? OLY_STATS_MAX_BLOCKS : want));
  hipLaunchKernelGGL(stats_partial_kernel, dim3(nb), dim3(THREADS), 0, oly_s(stream), (long)n, x,
                     ctx->stats_ws);
  hipLaunchKernelGGL(stats_finish_kernel, dim3(1), dim3(64), 0, oly_s(stream), nb, (long)n,
                     ctx->stats_ws, stats3_out);
  OLY_LAUNCH_CHECK(ctx, "stats kernels");
  return OLY_OK;
}

int oly_stats_finish(oly_ctx* ctx, int nblocks, int64_t n, double* stats3_out, oly_stream stream) {
  hipLaunchKernelGGL(stats_finish_kernel, dim3(1), dim3(64), 0, oly_s(stream), nblocks, (long)n,
                     ctx->stats_ws, stats3_out);
  OLY_LAUNCH_CHECK(ctx, "stats_finish_kernel");
  return OLY_OK;
}

extern "C" int oly_adv_normalize_parts(oly_ctx* ctx, int64_t n, float* x, const double* stats3_parts,
                                       int parts, int ddof, double eps, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (n < 0 || !stats3_parts || (n > 0 && !x) || ddof < 0 || parts < 1 || parts > OLY_MAX_STAT_PARTS)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_adv_normalize: bad argument (parts=%d)", parts);
  if (n == 0) return OLY_OK;
  long want = (n / 4 + THREADS - 1) / THREADS;
  int nb = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
  hipLaunchKernelGGL(normalize_kernel, dim3(nb), dim3(THREADS), 0, oly_s(stream), (long)n, x, stats3_parts,
                     parts, ddof, eps);
  OLY_LAUNCH_CHECK(ctx, "normalize_kernel");
  return OLY_OK;
}

extern "C" int oly_adv_normalize(oly_ctx* ctx, int64_t n, float* x, const double* stats3, int ddof,
                                 double eps, oly_stream stream) {
  return oly_adv_normalize_parts(ctx, n, x, stats3, 1, ddof, eps, stream);
}

extern "C" int oly_col_stats(oly_ctx* ctx, int B, int D, const float* x, double* colstats,
                             int accumulate, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (B < 0 || D <= 0 || D > OLY_MAX_OBS || !colstats || (B > 0 && !x))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_col_stats: bad argument (B=%d D=%d)", B, D);
  if ((D & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && D >= 4) {
    const int D4 = D >> 2, lanes4 = THREADS / D4, tpb4 = lanes4 * D4;
    int nb4 = (B + 8 * lanes4 - 1) / (8 * lanes4);  // >= 8 rows per thread
    if (nb4 < 1) nb4 = 1;
    if (nb4 > OLY_STATS_MAX_BLOCKS) nb4 = OLY_STATS_MAX_BLOCKS;
    const int rpb4 = (B + nb4 - 1) / nb4;
    hipLaunchKernelGGL(col_partial4_kernel, dim3(nb4), dim3(tpb4), sizeof(double) * 2 * lanes4 * D,
                       oly_s(stream), B, D, rpb4, x, ctx->stats_ws);
    hipLaunchKernelGGL(col_finish_kernel, dim3(D), dim3(64), 0, oly_s(stream), nb4, B, D,
                       ctx->stats_ws, colstats, accumulate);
    OLY_LAUNCH_CHECK(ctx, "col stats kernels");
    return OLY_OK;
  }
  const int lanes = THREADS / D;
  const int tpb = lanes * D;
  int nb = (B + 64 * lanes - 1) / (64 * lanes);  // >= 64 rows per lane-row keeps blocks busy
  if (nb < 1) nb = 1;
  if (nb > OLY_STATS_MAX_BLOCKS) nb = OLY_STATS_MAX_BLOCKS;
  const int rpb = (B + nb - 1) / nb;
  hipLaunchKernelGGL(col_partial_kernel, dim3(nb), dim3(tpb), sizeof(double) * 2 * tpb, oly_s(stream),
                     B, D, rpb, x, ctx->stats_ws);
  hipLaunchKernelGGL(col_finish_kernel, dim3(D), dim3(64), 0, oly_s(stream), nb, B, D,
                     ctx->stats_ws, colstats, accumulate);
  OLY_LAUNCH_CHECK(ctx, "col stats kernels");
  return OLY_OK;
}
