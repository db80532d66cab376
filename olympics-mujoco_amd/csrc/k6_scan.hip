// K6: reverse return / advantage scan over a [T,N] rollout block.
//   OLY_SCAN_RETURN  PPOBuffer.finish_path            rl/algos/ppo.py:68-84 (+ :335 adv)
//   OLY_SCAN_GAE     mushroom_rl compute_gae          call site gail_TRPO.py:126-127
//
// The recurrence is sequential in t and must round exactly like the reference (no
// re-association), so one lane owns one environment and walks t = T-1 .. 0.  For every t the
// 64 lanes of a wave touch 64 consecutive elements of each [T,N] array (coalesced); loads of
// a CHUNK of steps are issued before the dependent fp chain so that the wave keeps CHUNK x 4
// loads in flight.  One wave per workgroup: with N = 4096 that is 64 workgroups on 64
// different CUs, each with its own memory pipeline.  Bound: HBM/latency, 17-21 B per element.
#include "oly_common.h"

namespace {

constexpr int CHUNK = 16;

struct Chunk {
  float r[CHUNK], v[CHUNK], nv[CHUNK];
  uint8_t f[CHUNK];
};

template <int MODE>
__device__ __forceinline__ void load_chunk(Chunk& c, int t_hi, int T, int N, int n,
                                           const float* __restrict__ rew, const float* __restrict__ val,
                                           const float* __restrict__ next_val,
                                           const uint8_t* __restrict__ flags) {
#pragma unroll
  for (int k = 0; k < CHUNK; ++k) {
    const int t = t_hi - k;
    if (t >= 0) {
      const size_t e = (size_t)t * N + n;
      c.r[k] = rew[e];
      c.v[k] = val[e];
      uint8_t f = flags[e];
      if (t == T - 1) f |= OLY_FLAG_LAST;  // the block end always cuts the segment
      c.f[k] = f;
      if (MODE == OLY_SCAN_GAE)
        c.nv[k] = next_val[e];
      else
        c.nv[k] = ((f & OLY_FLAG_LAST) && !(f & OLY_FLAG_ABSORBING)) ? next_val[e] : 0.f;
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(64) void scan_kernel(int T, int N, double gamma, double lam,
                                                  const float* __restrict__ rew,
                                                  const float* __restrict__ val,
                                                  const float* __restrict__ next_val,
                                                  const uint8_t* __restrict__ flags,
                                                  float* __restrict__ ret, float* __restrict__ adv) {
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  const float g32 = (float)gamma;
  const float gl32 = (float)(gamma * lam);
  double R = 0.0;      // RETURN mode carry (float64, as numpy promotes it)
  float a_next = 0.f;  // GAE mode carry (float32 arrays in the reference)
  // two register chunks: the loads of chunk i+1 are in flight while chunk i runs its
  // dependent fp chain (software pipelining; the recurrence itself cannot be re-associated)
  Chunk cur, nxt;
  load_chunk<MODE>(cur, T - 1, T, N, n, rew, val, next_val, flags);
  for (int t_hi = T - 1; t_hi >= 0; t_hi -= CHUNK) {
    if (t_hi - CHUNK >= 0) load_chunk<MODE>(nxt, t_hi - CHUNK, T, N, n, rew, val, next_val, flags);
#pragma unroll
    for (int k = 0; k < CHUNK; ++k) {
      const int t = t_hi - k;
      if (t >= 0) {
        const size_t e = (size_t)t * N + n;
        if (MODE == OLY_SCAN_RETURN) {
          if (cur.f[k] & OLY_FLAG_LAST) {
            const float p = g32 * cur.nv[k];  // python float * float32 array: float32 product
            R = (double)p + (double)cur.r[k];
          } else {
            R = gamma * R + (double)cur.r[k];
          }
          const float rt = (float)R;
          ret[e] = rt;
          adv[e] = rt - cur.v[k];
        } else {
          float a;
          if (cur.f[k] & OLY_FLAG_LAST) {
            a = cur.r[k] - cur.v[k];
            if (!(cur.f[k] & OLY_FLAG_ABSORBING)) a += g32 * cur.nv[k];
          } else {
            a = cur.r[k] + g32 * cur.nv[k] - cur.v[k] + gl32 * a_next;
          }
          adv[e] = a;
          ret[e] = a + cur.v[k];
          a_next = a;
        }
      }
    }
    cur = nxt;
  }
}

}  // namespace

extern "C" int oly_return_scan(oly_ctx* ctx, int mode, int T, int N, double gamma, double lam,
                               const float* rew, const float* val, const float* next_val,
                               const uint8_t* flags, float* ret, float* adv, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (T < 0 || N < 0) OLY_FAIL(ctx, OLY_EINVAL, "oly_return_scan: negative T or N");
  if (T == 0 || N == 0) return OLY_OK;
  if (!rew || !val || !next_val || !flags || !ret || !adv)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_return_scan: NULL pointer");
  dim3 grid((N + 63) / 64), block(64);
  if (mode == OLY_SCAN_RETURN)
    hipLaunchKernelGGL(scan_kernel<OLY_SCAN_RETURN>, grid, block, 0, oly_s(stream), T, N, gamma, lam,
                       rew, val, next_val, flags, ret, adv);
  else if (mode == OLY_SCAN_GAE)
    hipLaunchKernelGGL(scan_kernel<OLY_SCAN_GAE>, grid, block, 0, oly_s(stream), T, N, gamma, lam, rew,
                       val, next_val, flags, ret, adv);
  else
    OLY_FAIL(ctx, OLY_EINVAL, "oly_return_scan: unknown mode %d", mode);
  OLY_LAUNCH_CHECK(ctx, "scan_kernel");
  return OLY_OK;
}
