// K8: discriminator-reward pre-amble and epilogue (the MLP GEMMs stay in PyTorch-ROCm).
//   oly_disc_standardize  prepare_discrim_inputs gail_TRPO.py:297-313 + Standardizer.forward
//                         imitation_lib/utils/networks.py:68-74
//   oly_obs_filter        Normalize._obfilt rl/envs/normalize.py:139-147
//   oly_disc_reparam      reparameterize networks.py:21-24
//   oly_disc_reward       GAIL.make_discrim_reward gail_TRPO.py:320-327
// Elementwise, HBM-bound: 8 B/element (standardise), 16 B (reparam), 8 B (reward).
#include "oly_common.h"

namespace {
constexpr int THREADS = 256;

__global__ __launch_bounds__(THREADS) void standardize_kernel(int B, int Dx, int D,
                                                              const float* __restrict__ x,
                                                              const int* __restrict__ mask,
                                                              const double* __restrict__ mean,
                                                              const double* __restrict__ sd,
                                                              float* __restrict__ out) {
  const long total = (long)B * D;
  const long stride = (long)gridDim.x * THREADS;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += stride) {
    const int b = (int)(e / D), j = (int)(e - (long)b * D);
    const int c = mask ? mask[j] : j;
    // float32 batch minus float64 mean, over float64 std, narrowed by .float()
    out[e] = (float)(((double)x[(size_t)b * Dx + c] - mean[j]) / sd[j]);
  }
}

// Row-major [B,D] with D % 4 == 0 and 16-B aligned rows: one float4 per lane, the column of a
// chunk is a 32-bit modulo of the chunk index (no 64-bit division per element).
__global__ __launch_bounds__(THREADS) void standardize4_kernel(long total4, int D4,
                                                               const float4* __restrict__ x,
                                                               const double* __restrict__ mean,
                                                               const double* __restrict__ sd,
                                                               float4* __restrict__ out) {
  const long stride = (long)gridDim.x * THREADS;
  long e = (long)blockIdx.x * THREADS + threadIdx.x;
  if (e >= total4) return;
  int c4 = (int)(e % D4);
  const int step = (int)(stride % D4);
  for (; e < total4; e += stride) {
    const float4 v = x[e];
    const int j = 4 * c4;
    float4 o;
    o.x = (float)(((double)v.x - mean[j]) / sd[j]);
    o.y = (float)(((double)v.y - mean[j + 1]) / sd[j + 1]);
    o.z = (float)(((double)v.z - mean[j + 2]) / sd[j + 2]);
    o.w = (float)(((double)v.w - mean[j + 3]) / sd[j + 3]);
    out[e] = o;
    c4 += step;
    if (c4 >= D4) c4 -= D4;
  }
}

__global__ __launch_bounds__(THREADS) void obs_filter4_kernel(long total4, int D4, const float4* __restrict__ x,
                                                              const double* __restrict__ mean,
                                                              const double* __restrict__ var, double eps,
                                                              double clip, float4* __restrict__ out) {
  __shared__ double s_den[OLY_MAX_OBS], s_mean[OLY_MAX_OBS];   // sqrt(var + eps) once per column, not per element
  for (int j = threadIdx.x; j < 4 * D4; j += THREADS) {
    s_den[j] = sqrt(var[j] + eps);
    s_mean[j] = mean[j];
  }
  __syncthreads();
  const long stride = (long)gridDim.x * THREADS;
  long e = (long)blockIdx.x * THREADS + threadIdx.x;
  if (e >= total4) return;
  int c4 = (int)(e % D4);
  const int step = (int)(stride % D4);
  for (; e < total4; e += stride) {
    const float4 v = x[e];
    const int j = 4 * c4;
    const float in[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double t = ((double)in[k] - s_mean[j + k]) / s_den[j + k];
      if (clip > 0.0) t = fmin(fmax(t, -clip), clip);
      o[k] = (float)t;
    }
    out[e] = make_float4(o[0], o[1], o[2], o[3]);
    c4 += step;
    if (c4 >= D4) c4 -= D4;
  }
}

// Normalize._obfilt (rl/envs/normalize.py:139-147): clip((obs - mean) / sqrt(var + eps), -c, c)
__global__ __launch_bounds__(THREADS) void obs_filter_kernel(long total, int D,
                                                             const float* __restrict__ x,
                                                             const double* __restrict__ mean,
                                                             const double* __restrict__ var, double eps,
                                                             double clip, float* __restrict__ out) {
  const long stride = (long)gridDim.x * THREADS;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += stride) {
    const int j = (int)(e % D);
    double v = ((double)x[e] - mean[j]) / sqrt(var[j] + eps);
    if (clip > 0.0) v = fmin(fmax(v, -clip), clip);
    out[e] = (float)v;
  }
}

__global__ __launch_bounds__(THREADS) void reparam_kernel(long n, const float* __restrict__ mu,
                                                          const float* __restrict__ logvar,
                                                          const float* __restrict__ eps,
                                                          float* __restrict__ z) {
  const long stride = (long)gridDim.x * THREADS;
  for (long i = (long)blockIdx.x * THREADS + threadIdx.x; i < n; i += stride)
    z[i] = mu[i] + expf(logvar[i] / 2.0f) * eps[i];
}

__device__ __forceinline__ float reward_of(float d) {
  // numpy evaluates every step in float32 (gail_TRPO.py:320-327 on the network's float32 output), and so
  // does this: expf / logf are the <= 1 ulp device functions, the same class of error as numpy's own
  // float32 exp / log; the 1 - p cancellation amplifies either to the tolerance the tests state.
  // (Round 1 took exp / log in fp64: 27 % of the HBM peak, fp64-transcendental-bound.)
  const float e = expf(-d);
  const float p = 1.0f / (1.0f + e);
  const float q = 1.0f - p + 1e-8f;
  return -logf(q);
}

// vec4: both pointers 16-byte aligned; the n % 4 tail goes through the scalar lanes of the last pass
__global__ __launch_bounds__(THREADS) void reward_kernel(long n, int vec4, const float* __restrict__ d,
                                                         float* __restrict__ r) {
  const long stride = (long)gridDim.x * THREADS;
  const long tid = (long)blockIdx.x * THREADS + threadIdx.x;
  long done = 0;
  if (vec4) {
    const long n4 = n >> 2;
    for (long i = tid; i < n4; i += stride) {
      const float4 x = reinterpret_cast<const float4*>(d)[i];
      reinterpret_cast<float4*>(r)[i] = make_float4(reward_of(x.x), reward_of(x.y), reward_of(x.z), reward_of(x.w));
    }
    done = n4 << 2;
  }
  for (long i = done + tid; i < n; i += stride) r[i] = reward_of(d[i]);
}

inline int blocks_for(long n) {
  long b = (n + THREADS - 1) / THREADS;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
}  // namespace

extern "C" int oly_disc_standardize(oly_ctx* ctx, int B, int Dx, int D, const float* x,
                                    const int32_t* mask, const double* mean, const double* sd,
                                    float* out, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (B < 0 || Dx <= 0 || D <= 0 || (!mask && D != Dx) || !mean || !sd || (B > 0 && (!x || !out)))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_disc_standardize: bad argument");
  if (B == 0) return OLY_OK;
  if (!mask && (D & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    const long total4 = (long)B * (D / 4);
    hipLaunchKernelGGL(standardize4_kernel, dim3(blocks_for(total4)), dim3(THREADS), 0, oly_s(stream), total4, D / 4,
                       reinterpret_cast<const float4*>(x), mean, sd, reinterpret_cast<float4*>(out));
    OLY_LAUNCH_CHECK(ctx, "standardize4_kernel");
    return OLY_OK;
  }
  hipLaunchKernelGGL(standardize_kernel, dim3(blocks_for((long)B * D)), dim3(THREADS), 0, oly_s(stream),
                     B, Dx, D, x, mask, mean, sd, out);
  OLY_LAUNCH_CHECK(ctx, "standardize_kernel");
  return OLY_OK;
}

extern "C" int oly_obs_filter(oly_ctx* ctx, int B, int D, const float* x, const double* mean,
                              const double* var, double eps, double clip, float* out,
                              oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (B < 0 || D <= 0 || !mean || !var || (B > 0 && (!x || !out)))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_obs_filter: bad argument");
  if (B == 0) return OLY_OK;
  if ((D & 3) == 0 && D <= OLY_MAX_OBS && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    const long total4 = (long)B * (D / 4);
    hipLaunchKernelGGL(obs_filter4_kernel, dim3(blocks_for(total4)), dim3(THREADS), 0, oly_s(stream), total4, D / 4,
                       reinterpret_cast<const float4*>(x), mean, var, eps, clip, reinterpret_cast<float4*>(out));
    OLY_LAUNCH_CHECK(ctx, "obs_filter4_kernel");
    return OLY_OK;
  }
  hipLaunchKernelGGL(obs_filter_kernel, dim3(blocks_for((long)B * D)), dim3(THREADS), 0, oly_s(stream),
                     (long)B * D, D, x, mean, var, eps, clip, out);
  OLY_LAUNCH_CHECK(ctx, "obs_filter_kernel");
  return OLY_OK;
}

extern "C" int oly_disc_reparam(oly_ctx* ctx, int64_t n, const float* mu, const float* logvar,
                                const float* eps, float* z, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (n < 0 || (n > 0 && (!mu || !logvar || !eps || !z)))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_disc_reparam: bad argument");
  if (n == 0) return OLY_OK;
  hipLaunchKernelGGL(reparam_kernel, dim3(blocks_for(n)), dim3(THREADS), 0, oly_s(stream), (long)n, mu,
                     logvar, eps, z);
  OLY_LAUNCH_CHECK(ctx, "reparam_kernel");
  return OLY_OK;
}

extern "C" int oly_disc_reward(oly_ctx* ctx, int64_t B, const float* logits, float* reward,
                               oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (B < 0 || (B > 0 && (!logits || !reward))) OLY_FAIL(ctx, OLY_EINVAL, "oly_disc_reward: bad argument");
  if (B == 0) return OLY_OK;
  const int vec4 = ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(reward)) & 15) == 0;
  hipLaunchKernelGGL(reward_kernel, dim3(blocks_for(vec4 ? (B + 3) / 4 : B)), dim3(THREADS), 0, oly_s(stream), (long)B,
                     vec4, logits, reward);
  OLY_LAUNCH_CHECK(ctx, "reward_kernel");
  return OLY_OK;
}
