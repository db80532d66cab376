// K9: the loss terms of PPO.update_policy (rl/algos/ppo.py:232-282), forward value and
// gradients in ONE pass over the minibatch, plus the mirror-symmetry pieces:
//   oly_signed_perm   mirror_observation / mirror_action   rl/envs/wrappers.py:51-57,75-82
//   oly_mirror_loss   (det - mirror_action(policy(mirror_obs)))^2 mean   ppo.py:261-268
//   oly_ppo_loss      clip surrogate, value loss, entropy, approx KL, clip fraction
//                     ppo.py:236-259,270-273
// The reference evaluates ~30 elementwise torch ops + 6 reductions per minibatch and their
// autograd twins; here every row is read once.  Bound: HBM at full-batch sizes
// (16*A + 12 B in, 4*A + 4 B out per row), launch latency at the reference's minibatch sizes.
// Elementwise arithmetic is fp32 in torch's operation order; sums are fp64 in a fixed order.
#include "oly_common.h"

namespace {

constexpr int THREADS = 256;
constexpr float LOG_SQRT_2PI = 0.9189385332046727f;           // math.log(math.sqrt(2*math.pi))
constexpr float ENTROPY_CONST = 0.5f + 0.9189385332046727f;   // 0.5 + 0.5*math.log(2*math.pi)

__global__ __launch_bounds__(THREADS) void signed_perm_kernel(long total, int D,
                                                              const float* __restrict__ x,
                                                              const int* __restrict__ src,
                                                              const float* __restrict__ sign,
                                                              float* __restrict__ out) {
  const long stride = (long)gridDim.x * THREADS;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += stride) {
    const long b = e / D;
    const int j = (int)(e - b * D);
    out[e] = sign[j] * x[b * D + src[j]];
  }
}

// block partial of one fp64 value per thread -> ws[blockIdx.x * nq + q]
template <int NQ>
__device__ __forceinline__ void block_partials(double (&v)[NQ], double* __restrict__ ws) {
  __shared__ double sh[NQ * (THREADS / 64)];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double s = wave_sum(v[q]);
    if ((threadIdx.x & 63) == 0) sh[q * (THREADS / 64) + w] = s;
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    double t = 0.0;
    for (int i = 0; i < THREADS / 64; ++i) t += sh[threadIdx.x * (THREADS / 64) + i];
    ws[(size_t)blockIdx.x * NQ + threadIdx.x] = t;
  }
}

// lane i sums partials i, i+64, ... in order, then the fixed shuffle tree; out[q] = sum * scale[q]
template <int NQ>
__global__ __launch_bounds__(64) void finish_kernel(int nblocks, const double* __restrict__ ws,
                                                    double* __restrict__ out, double s0, double s1,
                                                    double s2, double s3, double s4) {
  const double scale[5] = {s0, s1, s2, s3, s4};
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += ws[(size_t)b * NQ + q];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[q] = s * scale[q];
  }
}

__global__ __launch_bounds__(THREADS) void mirror_loss_kernel(long total, int A, float gscale,
                                                              const float* __restrict__ det,
                                                              const float* __restrict__ mir,
                                                              const int* __restrict__ src,
                                                              const float* __restrict__ sign,
                                                              float* __restrict__ grad_det,
                                                              float* __restrict__ grad_mir,
                                                              double* __restrict__ ws) {
  double v[1] = {0.0};
  const long stride = (long)gridDim.x * THREADS;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += stride) {
    const long b = e / A;
    const int j = (int)(e - b * A);
    const int i = src[j];
    const float sg = sign[j];
    const float d = det[e] - sg * mir[b * A + i];
    v[0] += (double)(d * d);
    const float g = gscale * d;  // 2 / (B*A) * diff
    if (grad_det) grad_det[e] = g;
    if (grad_mir) grad_mir[b * A + i] = -sg * g;  // src is a permutation: no two j share i
  }
  block_partials<1>(v, ws);
}

struct PpoArgs {
  int B, A;
  const float *mu, *sd, *old_mu, *old_sd, *action, *adv, *ret, *value;
  int sd_mode, old_sd_mode;
  float clip, vf_coeff;
  float *grad_mu, *grad_sd, *grad_value;
  double* ws;
};

__device__ __forceinline__ float sd_at(const float* sd, int mode, long row, int A, int j) {
  return mode == OLY_STD_SCALAR ? sd[0] : mode == OLY_STD_PER_DIM ? sd[j] : sd[row * A + j];
}

// One workgroup owns RB consecutive rows: [RB, A] tiles of mu / old_mu / action are staged
// through LDS with dense loads (row stride padded to an odd word count), thread r then walks
// row r; d actor / d mu goes back through the same LDS tile to a dense store.
__global__ __launch_bounds__(THREADS) void ppo_loss_kernel(PpoArgs p, int RB) {
  extern __shared__ float lds[];
  const int A = p.A, AS = A | 1;
  float* s_mu = lds;
  float* s_old = s_mu + RB * AS;
  float* s_act = s_old + RB * AS;
  const float lo = 1.0f - p.clip, hi = 1.0f + p.clip;
  const float invB = 1.0f / (float)p.B;
  double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const long ntiles = ((long)p.B + RB - 1) / RB;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long row0 = tile * RB;
    const int rows = (int)min((long)RB, (long)p.B - row0);
    const int n = rows * A;
    const long base = row0 * A;
    for (int e = threadIdx.x; e < n; e += THREADS) {
      const int r = e / A, j = e - r * A;
      s_mu[r * AS + j] = p.mu[base + e];
      s_old[r * AS + j] = p.old_mu[base + e];
      s_act[r * AS + j] = p.action[base + e];
    }
    __syncthreads();
    const int r = threadIdx.x;
    if (r < rows) {
      const long row = row0 + r;
      float lp = 0.0f, olp = 0.0f, ent = 0.0f;
      // a scalar std is the same value in every element: its log and 2*var are taken once (same
      // numbers torch computes elementwise on the broadcast tensor)
      const bool sc = p.sd_mode == OLY_STD_SCALAR, osc = p.old_sd_mode == OLY_STD_SCALAR;
      const float sd0 = sc ? p.sd[0] : 1.0f, osd0 = osc ? p.old_sd[0] : 1.0f;
      const float lsd0 = logf(sd0), losd0 = logf(osd0), v20 = 2.0f * (sd0 * sd0), ov20 = 2.0f * (osd0 * osd0);
      for (int j = 0; j < A; ++j) {
        const float a = s_act[r * AS + j];
        const float sd = sc ? sd0 : sd_at(p.sd, p.sd_mode, row, A, j);
        const float osd = osc ? osd0 : sd_at(p.old_sd, p.old_sd_mode, row, A, j);
        const float lsd = sc ? lsd0 : logf(sd), losd = osc ? losd0 : logf(osd);
        const float v2 = sc ? v20 : 2.0f * (sd * sd), ov2 = osc ? ov20 : 2.0f * (osd * osd);
        const float t = a - s_mu[r * AS + j], ot = a - s_old[r * AS + j];
        // Normal.log_prob: -((v - loc)**2) / (2*var) - log(scale) - log(sqrt(2 pi))
        lp += -(t * t) / v2 - lsd - LOG_SQRT_2PI;
        olp += -(ot * ot) / ov2 - losd - LOG_SQRT_2PI;
        ent += ENTROPY_CONST + lsd;
      }
      const float log_ratio = lp - olp;
      const float ratio = expf(log_ratio);
      const float adv = p.adv[row];
      const float cpi = ratio * adv;
      const float rc = fminf(fmaxf(ratio, lo), hi);
      const float cl = rc * adv;
      v[0] += (double)fminf(cpi, cl);
      v[1] += (double)ent;
      const float dv = p.ret[row] - p.value[row];
      v[2] += (double)(dv * dv);
      v[3] += (double)((ratio - 1.0f) - log_ratio);
      v[4] += (fabsf(ratio - 1.0f) > p.clip) ? 1.0 : 0.0;
      if (p.grad_value) p.grad_value[row] = p.vf_coeff * 2.0f * (p.value[row] - p.ret[row]) * invB;
      // torch.min splits the gradient on ties; clamp passes it inside [lo, hi] (inclusive)
      const float inr = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
      const float w = cpi < cl ? 1.0f : (cpi == cl ? 0.5f + 0.5f * inr : inr);
      const float g_lp = -invB * adv * w * ratio;  // d actor / d lp
      if (p.grad_mu || p.grad_sd) {
        for (int j = 0; j < A; ++j) {
          const float sd = sd_at(p.sd, p.sd_mode, row, A, j);
          const float t = s_act[r * AS + j] - s_mu[r * AS + j];
          s_mu[r * AS + j] = g_lp * t / (sd * sd);
          if (p.grad_sd) s_old[r * AS + j] = g_lp * (t * t / (sd * sd * sd) - 1.0f / sd);
        }
      }
    }
    __syncthreads();
    if (p.grad_mu || p.grad_sd) {
      for (int e = threadIdx.x; e < n; e += THREADS) {
        const int rr = e / A, j = e - rr * A;
        if (p.grad_mu) p.grad_mu[base + e] = s_mu[rr * AS + j];
        if (p.grad_sd) p.grad_sd[base + e] = s_old[rr * AS + j];
      }
    }
    __syncthreads();
  }
  block_partials<5>(v, p.ws);
}

// A % 4 == 0, A <= 16, 16-B aligned rows: one lane owns one row in REGISTERS (A4 float4 loads per
// array, float4 gradient stores), no LDS staging, no barriers in the row loop.  Same arithmetic in
// the same order as ppo_loss_kernel.
template <int A4>
__global__ __launch_bounds__(THREADS) void ppo_loss_rows_kernel(PpoArgs p) {
  constexpr int A = 4 * A4;
  const float lo = 1.0f - p.clip, hi = 1.0f + p.clip;
  const float invB = 1.0f / (float)p.B;
  double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const bool sc = p.sd_mode == OLY_STD_SCALAR, osc = p.old_sd_mode == OLY_STD_SCALAR;
  const float sd0 = sc ? p.sd[0] : 1.0f, osd0 = osc ? p.old_sd[0] : 1.0f;
  const float lsd0 = logf(sd0), losd0 = logf(osd0), v20 = 2.0f * (sd0 * sd0), ov20 = 2.0f * (osd0 * osd0);
  const long stride = (long)gridDim.x * THREADS;
  for (long row = (long)blockIdx.x * THREADS + threadIdx.x; row < p.B; row += stride) {
    float m[A], om[A], ac[A];
    const float4* m4 = reinterpret_cast<const float4*>(p.mu) + row * A4;
    const float4* o4 = reinterpret_cast<const float4*>(p.old_mu) + row * A4;
    const float4* a4 = reinterpret_cast<const float4*>(p.action) + row * A4;
#pragma unroll
    for (int q = 0; q < A4; ++q) {
      const float4 x = m4[q], y = o4[q], z = a4[q];
      m[4 * q] = x.x; m[4 * q + 1] = x.y; m[4 * q + 2] = x.z; m[4 * q + 3] = x.w;
      om[4 * q] = y.x; om[4 * q + 1] = y.y; om[4 * q + 2] = y.z; om[4 * q + 3] = y.w;
      ac[4 * q] = z.x; ac[4 * q + 1] = z.y; ac[4 * q + 2] = z.z; ac[4 * q + 3] = z.w;
    }
    float lp = 0.0f, olp = 0.0f, ent = 0.0f;
#pragma unroll
    for (int j = 0; j < A; ++j) {
      const float sd = sc ? sd0 : sd_at(p.sd, p.sd_mode, row, A, j);
      const float osd = osc ? osd0 : sd_at(p.old_sd, p.old_sd_mode, row, A, j);
      const float lsd = sc ? lsd0 : logf(sd), losd = osc ? losd0 : logf(osd);
      const float v2 = sc ? v20 : 2.0f * (sd * sd), ov2 = osc ? ov20 : 2.0f * (osd * osd);
      const float t = ac[j] - m[j], ot = ac[j] - om[j];
      lp += -(t * t) / v2 - lsd - LOG_SQRT_2PI;
      olp += -(ot * ot) / ov2 - losd - LOG_SQRT_2PI;
      ent += ENTROPY_CONST + lsd;
    }
    const float log_ratio = lp - olp;
    const float ratio = expf(log_ratio);
    const float adv = p.adv[row];
    const float cpi = ratio * adv;
    const float rc = fminf(fmaxf(ratio, lo), hi);
    const float cl = rc * adv;
    v[0] += (double)fminf(cpi, cl);
    v[1] += (double)ent;
    const float rt = p.ret[row], vl = p.value[row];
    const float dv = rt - vl;
    v[2] += (double)(dv * dv);
    v[3] += (double)((ratio - 1.0f) - log_ratio);
    v[4] += (fabsf(ratio - 1.0f) > p.clip) ? 1.0 : 0.0;
    if (p.grad_value) p.grad_value[row] = p.vf_coeff * 2.0f * (vl - rt) * invB;
    const float inr = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
    const float w = cpi < cl ? 1.0f : (cpi == cl ? 0.5f + 0.5f * inr : inr);
    const float g_lp = -invB * adv * w * ratio;
    if (p.grad_mu || p.grad_sd) {
      float gm[A], gs[A];
#pragma unroll
      for (int j = 0; j < A; ++j) {
        const float sd = sc ? sd0 : sd_at(p.sd, p.sd_mode, row, A, j);
        const float t = ac[j] - m[j];
        gm[j] = g_lp * t / (sd * sd);
        gs[j] = g_lp * (t * t / (sd * sd * sd) - 1.0f / sd);
      }
#pragma unroll
      for (int q = 0; q < A4; ++q) {
        if (p.grad_mu)
          reinterpret_cast<float4*>(p.grad_mu)[row * A4 + q] = make_float4(gm[4 * q], gm[4 * q + 1], gm[4 * q + 2], gm[4 * q + 3]);
        if (p.grad_sd)
          reinterpret_cast<float4*>(p.grad_sd)[row * A4 + q] = make_float4(gs[4 * q], gs[4 * q + 1], gs[4 * q + 2], gs[4 * q + 3]);
      }
    }
  }
  block_partials<5>(v, p.ws);
}

inline int blocks_for(long n, int cap) {
  long b = (n + THREADS - 1) / THREADS;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int oly_signed_perm(oly_ctx* ctx, int B, int D, const float* x, const int32_t* src,
                               const float* sign, float* out, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (B < 0 || D <= 0 || !src || !sign || (B > 0 && (!x || !out || x == out)))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_signed_perm: bad argument (B=%d D=%d)", B, D);
  if (B == 0) return OLY_OK;
  const long total = (long)B * D;
  hipLaunchKernelGGL(signed_perm_kernel, dim3(blocks_for(total, 4096)), dim3(THREADS), 0, oly_s(stream),
                     total, D, x, src, sign, out);
  OLY_LAUNCH_CHECK(ctx, "signed_perm_kernel");
  return OLY_OK;
}

extern "C" int oly_mirror_loss(oly_ctx* ctx, int B, int A, const float* det, const float* mir,
                               const int32_t* src, const float* sign, double* loss_out,
                               float* grad_det, float* grad_mir, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (B <= 0 || A <= 0 || !det || !mir || !src || !sign || !loss_out)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_mirror_loss: bad argument (B=%d A=%d)", B, A);
  const long total = (long)B * A;
  const int nb = blocks_for(total, OLY_STATS_MAX_BLOCKS);
  const double inv = 1.0 / (double)total;
  hipLaunchKernelGGL(mirror_loss_kernel, dim3(nb), dim3(THREADS), 0, oly_s(stream), total, A,
                     (float)(2.0 * inv), det, mir, src, sign, grad_det, grad_mir, ctx->stats_ws);
  hipLaunchKernelGGL(finish_kernel<1>, dim3(1), dim3(64), 0, oly_s(stream), nb, ctx->stats_ws, loss_out,
                     inv, 0.0, 0.0, 0.0, 0.0);
  OLY_LAUNCH_CHECK(ctx, "mirror loss kernels");
  return OLY_OK;
}

extern "C" int oly_ppo_loss(oly_ctx* ctx, int B, int A, const float* mu, const float* sd, int sd_mode,
                            const float* old_mu, const float* old_sd, int old_sd_mode,
                            const float* action, const float* adv, const float* ret,
                            const float* value, float clip, float vf_coeff, double* scal_out,
                            float* grad_mu, float* grad_sd, float* grad_value, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (B <= 0 || A <= 0 || A > OLY_MAX_ACT || !mu || !sd || !old_mu || !old_sd || !action || !adv ||
      !ret || !value || !scal_out || sd_mode < 0 || sd_mode > 2 || old_sd_mode < 0 || old_sd_mode > 2)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_ppo_loss: bad argument (B=%d A=%d)", B, A);
  const int RB = THREADS;
  const size_t lds = sizeof(float) * 3 * RB * (A | 1);  // <= 3*256*65*4 = 195 KB at A = 64: cap rows
  int rb = RB;
  size_t need = lds;
  while (need > 60 * 1024) { rb >>= 1; need = sizeof(float) * 3 * rb * (A | 1); }
  long ntiles = ((long)B + rb - 1) / rb;
  const int nb = (int)(ntiles > OLY_STATS_MAX_BLOCKS ? OLY_STATS_MAX_BLOCKS : ntiles);
  PpoArgs p{B, A, mu, sd, old_mu, old_sd, action, adv, ret, value, sd_mode, old_sd_mode, clip, vf_coeff,
            grad_mu, grad_sd, grad_value, ctx->stats_ws};
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if ((A & 3) == 0 && A <= 16 && al16(mu) && al16(old_mu) && al16(action) && al16(grad_mu) && al16(grad_sd)) {
    const int nbr = blocks_for(B, OLY_STATS_MAX_BLOCKS);
    switch (A / 4) {
      case 1: hipLaunchKernelGGL(ppo_loss_rows_kernel<1>, dim3(nbr), dim3(THREADS), 0, oly_s(stream), p); break;
      case 2: hipLaunchKernelGGL(ppo_loss_rows_kernel<2>, dim3(nbr), dim3(THREADS), 0, oly_s(stream), p); break;
      case 3: hipLaunchKernelGGL(ppo_loss_rows_kernel<3>, dim3(nbr), dim3(THREADS), 0, oly_s(stream), p); break;
      default: hipLaunchKernelGGL(ppo_loss_rows_kernel<4>, dim3(nbr), dim3(THREADS), 0, oly_s(stream), p); break;
    }
    const double invBr = 1.0 / (double)B, invBAr = 1.0 / ((double)B * A);
    hipLaunchKernelGGL(finish_kernel<5>, dim3(1), dim3(64), 0, oly_s(stream), nbr, ctx->stats_ws, scal_out,
                       -invBr, -invBAr, (double)vf_coeff * invBr, invBr, invBr);
    OLY_LAUNCH_CHECK(ctx, "ppo loss kernels");
    return OLY_OK;
  }
  hipLaunchKernelGGL(ppo_loss_kernel, dim3(nb), dim3(THREADS), need, oly_s(stream), p, rb);
  const double invB = 1.0 / (double)B, invBA = 1.0 / ((double)B * A);
  hipLaunchKernelGGL(finish_kernel<5>, dim3(1), dim3(64), 0, oly_s(stream), nb, ctx->stats_ws, scal_out,
                     -invB, -invBA, (double)vf_coeff * invB, invB, invB);
  OLY_LAUNCH_CHECK(ctx, "ppo loss kernels");
  return OLY_OK;
}
