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
#include <cstdlib>
#include <type_traits>

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
__global__ __launch_bounds__(64) void scan_kernel(int T, int N, int epw, double gamma, double lam,
                                                  const float* __restrict__ rew,
                                                  const float* __restrict__ val,
                                                  const float* __restrict__ next_val,
                                                  const uint8_t* __restrict__ flags,
                                                  float* __restrict__ ret, float* __restrict__ adv) {
  // epw = environments per wave: a lone wave issues ~1 instruction per 4 cycles whatever its
  // lane count, so with few environments the chains are spread over MORE (narrower) waves
  const int n = blockIdx.x * epw + threadIdx.x;
  if (threadIdx.x >= epw || n >= N) return;
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

// ---------------------------------------------------------------------------------------
// Tiled variant (default): a workgroup of 4 waves owns 64 environments.  ALL waves prefetch the
// next TT-step tile of rew/val/next_val/flags (4 x 32 loads in flight per workgroup instead of
// one wave's chunk) while wave 0 runs the sequential recurrence over the current tile out of
// LDS.  Same arithmetic, same order: results are bit-identical to scan_kernel.
// ---------------------------------------------------------------------------------------
constexpr int TT = 64;          // time steps per tile
constexpr int SCAN_THREADS = 256;

template <int MODE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_kernel(int T, int N, double gamma, double lam,
                                                                 const float* __restrict__ rew,
                                                                 const float* __restrict__ val,
                                                                 const float* __restrict__ next_val,
                                                                 const uint8_t* __restrict__ flags,
                                                                 float* __restrict__ ret,
                                                                 float* __restrict__ adv) {
  __shared__ float s_r[TT][64], s_v[TT][64], s_nv[TT][64];
  __shared__ uint8_t s_f[TT][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const bool env_ok = n < N;
  constexpr int RPW = TT / 4;  // rows per wave per tile
  float pr[RPW], pv[RPW], pnv[RPW];
  uint8_t pf[RPW];
  const int ntiles = (T + TT - 1) / TT;

  auto prefetch = [&](int k) {
    const int t_top = T - 1 - k * TT;  // highest step of the tile; row tt <-> t = t_top - tt
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int t = t_top - (w + 4 * j);
      if (t >= 0 && env_ok) {
        const size_t e = (size_t)t * N + n;
        pr[j] = rew[e];
        pv[j] = val[e];
        uint8_t f = flags[e];
        if (t == T - 1) f |= OLY_FLAG_LAST;
        pf[j] = f;
        if (MODE == OLY_SCAN_GAE)
          pnv[j] = next_val[e];
        else
          pnv[j] = ((f & OLY_FLAG_LAST) && !(f & OLY_FLAG_ABSORBING)) ? next_val[e] : 0.f;
      }
    }
  };

  const float g32 = (float)gamma;
  const float gl32 = (float)(gamma * lam);
  double R = 0.0;
  float a_next = 0.f;
  prefetch(0);
  for (int k = 0; k < ntiles; ++k) {
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int tt = w + 4 * j;
      s_r[tt][lane] = pr[j];
      s_v[tt][lane] = pv[j];
      s_nv[tt][lane] = pnv[j];
      s_f[tt][lane] = pf[j];
    }
    __syncthreads();
    if (k + 1 < ntiles) prefetch(k + 1);
    if (w == 0 && env_ok) {
      const int t_top = T - 1 - k * TT;
#pragma unroll 8
      for (int tt = 0; tt < TT; ++tt) {
        const int t = t_top - tt;
        if (t < 0) break;
        const size_t e = (size_t)t * N + n;
        const float r = s_r[tt][lane], v = s_v[tt][lane], nv = s_nv[tt][lane];
        const uint8_t f = s_f[tt][lane];
        if (MODE == OLY_SCAN_RETURN) {
          if (f & OLY_FLAG_LAST) {
            const float p = g32 * nv;
            R = (double)p + (double)r;
          } else {
            R = gamma * R + (double)r;
          }
          const float rt = (float)R;
          ret[e] = rt;
          adv[e] = rt - v;
        } else {
          float a;
          if (f & OLY_FLAG_LAST) {
            a = r - v;
            if (!(f & OLY_FLAG_ABSORBING)) a += g32 * nv;
          } else {
            a = r + g32 * nv - v + gl32 * a_next;
          }
          adv[e] = a;
          ret[e] = a + v;
          a_next = a;
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Wide variant (default when N % 4 == 0 and the arrays are 16-B aligned): same tiling, but
// every global access is 16 B per lane (4 environments of one step) and the outputs go
// through LDS so that wave 0 only does the recurrence: per tile a lane issues 8 loads and 4
// stores instead of 32 and (on wave 0) 64.
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_wide_kernel(int T, int N, double gamma, double lam,
                                                                 const float* __restrict__ rew,
                                                                 const float* __restrict__ val,
                                                                 const float* __restrict__ next_val,
                                                                 const uint8_t* __restrict__ flags,
                                                                 float* __restrict__ ret,
                                                                 float* __restrict__ adv) {
  __shared__ __attribute__((aligned(16))) float s_r[TT][64], s_v[TT][64], s_nv[TT][64];
  __shared__ __attribute__((aligned(16))) float s_ret[TT][64], s_adv[TT][64];
  __shared__ __attribute__((aligned(16))) uint8_t s_f[TT][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n0 = blockIdx.x * 64;
  const int c4 = tid & 15;        // which group of 4 environments inside the 64
  const int rr = tid >> 4;        // row within a 16-row slab
  const bool col_ok = n0 + 4 * c4 < N;   // N % 4 == 0: a float4 is all-in or all-out
  const bool env_ok = n0 + lane < N;
  constexpr int SL = TT / 16;     // slabs per tile
  float4 pr[SL], pv[SL], pnv[SL];
  uchar4 pf[SL];
  const int ntiles = (T + TT - 1) / TT;

  auto prefetch = [&](int k) {
    const int t_top = T - 1 - k * TT;
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const int t = t_top - (rr + 16 * j);
      if (t >= 0 && col_ok) {
        const size_t e = (size_t)t * N + n0 + 4 * c4;
        pr[j] = *reinterpret_cast<const float4*>(rew + e);
        pv[j] = *reinterpret_cast<const float4*>(val + e);
        pnv[j] = *reinterpret_cast<const float4*>(next_val + e);
        pf[j] = *reinterpret_cast<const uchar4*>(flags + e);
      }
    }
  };

  const float g32 = (float)gamma;
  const float gl32 = (float)(gamma * lam);
  double R = 0.0;
  float a_next = 0.f;
  prefetch(0);
  for (int k = 0; k < ntiles; ++k) {
    const int t_top = T - 1 - k * TT;
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const int tt = rr + 16 * j;
      *reinterpret_cast<float4*>(&s_r[tt][4 * c4]) = pr[j];
      *reinterpret_cast<float4*>(&s_v[tt][4 * c4]) = pv[j];
      *reinterpret_cast<float4*>(&s_nv[tt][4 * c4]) = pnv[j];
      *reinterpret_cast<uchar4*>(&s_f[tt][4 * c4]) = pf[j];
    }
    __syncthreads();
    if (k + 1 < ntiles) prefetch(k + 1);
    if (w == 0 && env_ok) {
      // batches of SB steps: all LDS reads of a batch first (one round trip), then the
      // dependent fp chain in registers, then the LDS writes.  Rows with t < 0 (last tile
      // only, and last in time order) compute on garbage that is never stored or carried.
      constexpr int SB = 8;
      for (int tb = 0; tb < TT; tb += SB) {
        float r8[SB], v8[SB], nv8[SB], o_ret[SB], o_adv[SB];
        uint8_t f8[SB];
#pragma unroll
        for (int q = 0; q < SB; ++q) {
          r8[q] = s_r[tb + q][lane];
          v8[q] = s_v[tb + q][lane];
          nv8[q] = s_nv[tb + q][lane];
          f8[q] = s_f[tb + q][lane];
        }
#pragma unroll
        for (int q = 0; q < SB; ++q) {
          uint8_t f = f8[q];
          if (t_top - (tb + q) == T - 1) f |= OLY_FLAG_LAST;
          if (MODE == OLY_SCAN_RETURN) {
            if (f & OLY_FLAG_LAST) {
              const float nv = (f & OLY_FLAG_ABSORBING) ? 0.f : nv8[q];
              const float p = g32 * nv;
              R = (double)p + (double)r8[q];
            } else {
              R = gamma * R + (double)r8[q];
            }
            const float rt = (float)R;
            o_ret[q] = rt;
            o_adv[q] = rt - v8[q];
          } else {
            float a;
            if (f & OLY_FLAG_LAST) {
              a = r8[q] - v8[q];
              if (!(f & OLY_FLAG_ABSORBING)) a += g32 * nv8[q];
            } else {
              a = r8[q] + g32 * nv8[q] - v8[q] + gl32 * a_next;
            }
            o_adv[q] = a;
            o_ret[q] = a + v8[q];
            a_next = a;
          }
        }
#pragma unroll
        for (int q = 0; q < SB; ++q) {
          s_ret[tb + q][lane] = o_ret[q];
          s_adv[tb + q][lane] = o_adv[q];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const int tt = rr + 16 * j;
      const int t = t_top - tt;
      if (t >= 0 && col_ok) {
        const size_t e = (size_t)t * N + n0 + 4 * c4;
        *reinterpret_cast<float4*>(ret + e) = *reinterpret_cast<const float4*>(&s_ret[tt][4 * c4]);
        *reinterpret_cast<float4*>(adv + e) = *reinterpret_cast<const float4*>(&s_adv[tt][4 * c4]);
      }
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// Lean variant (default when the wide layout applies): the serial wave runs ONLY the
// recurrence.  The lanes that load a tile turn it, while it is still in their registers, into
// the per-step constant of the recurrence,
//   RETURN:  R_t = sel ? b : gamma * R_{t+1} + b      b = f64(r)            (sel = segment end:
//                                                      b = f64(g32 * nv_eff) + f64(r))
//   GAE:     a_t = sel ? c : c + gl32 * a_{t+1}       c = (r + g32*nv) - v  (sel: c = (r - v) [+ g32*nv])
// which are exactly the reference's operations in the reference's order; wave 0 reads (b|c, sel),
// does one multiply-add pair per step and writes the carry; all lanes then derive ret / adv
// (RETURN: ret = f32(R), adv = ret - v; GAE: adv = a, ret = a + v) during the store sweep.
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_lean_kernel(int T, int N, double gamma, double lam,
                                                                 const float* __restrict__ rew,
                                                                 const float* __restrict__ val,
                                                                 const float* __restrict__ next_val,
                                                                 const uint8_t* __restrict__ flags,
                                                                 float* __restrict__ ret,
                                                                 float* __restrict__ adv) {
  using carry_t = typename std::conditional<MODE == OLY_SCAN_RETURN, double, float>::type;
  __shared__ __attribute__((aligned(16))) carry_t s_b[TT][64];   // recurrence constant, then the carry
  __shared__ __attribute__((aligned(16))) float s_v[TT][64];
  __shared__ __attribute__((aligned(16))) uint8_t s_sel[TT][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n0 = blockIdx.x * 64;
  const int c4 = tid & 15, rr = tid >> 4;
  const bool col_ok = n0 + 4 * c4 < N;
  const bool env_ok = n0 + lane < N;
  constexpr int SL = TT / 16;
  float4 pr[SL], pv[SL], pnv[SL];
  uchar4 pf[SL];
  const int ntiles = (T + TT - 1) / TT;
  const float g32 = (float)gamma;
  const float gl32 = (float)(gamma * lam);

  auto prefetch = [&](int k) {
    const int t_top = T - 1 - k * TT;
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const int t = t_top - (rr + 16 * j);
      if (t >= 0 && col_ok) {
        const size_t e = (size_t)t * N + n0 + 4 * c4;
        pr[j] = *reinterpret_cast<const float4*>(rew + e);
        pv[j] = *reinterpret_cast<const float4*>(val + e);
        pnv[j] = *reinterpret_cast<const float4*>(next_val + e);
        pf[j] = *reinterpret_cast<const uchar4*>(flags + e);
      }
    }
  };
  auto konst = [&](float r, float v, float nv, uint8_t f, bool top) -> carry_t {
    const bool last = (f & OLY_FLAG_LAST) || top, ab = f & OLY_FLAG_ABSORBING;
    if (MODE == OLY_SCAN_RETURN) {
      if (last) {
        const float p = g32 * (ab ? 0.f : nv);
        return (carry_t)((double)p + (double)r);
      }
      return (carry_t)(double)r;
    }
    if (last) {
      float a = r - v;
      if (!ab) a += g32 * nv;
      return (carry_t)a;
    }
    return (carry_t)(r + g32 * nv - v);
  };

  carry_t carry = 0;
  prefetch(0);
  for (int k = 0; k < ntiles; ++k) {
    const int t_top = T - 1 - k * TT;
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const int tt = rr + 16 * j;
      const bool top = (t_top - tt) == T - 1;
      const float rv[4] = {pr[j].x, pr[j].y, pr[j].z, pr[j].w};
      const float vv[4] = {pv[j].x, pv[j].y, pv[j].z, pv[j].w};
      const float nn[4] = {pnv[j].x, pnv[j].y, pnv[j].z, pnv[j].w};
      const uint8_t ff[4] = {pf[j].x, pf[j].y, pf[j].z, pf[j].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s_b[tt][4 * c4 + q] = konst(rv[q], vv[q], nn[q], ff[q], top);
        s_sel[tt][4 * c4 + q] = (uint8_t)(((ff[q] & OLY_FLAG_LAST) || top) ? 1 : 0);
      }
      *reinterpret_cast<float4*>(&s_v[tt][4 * c4]) = pv[j];
    }
    __syncthreads();
    if (k + 1 < ntiles) prefetch(k + 1);
    if (w == 0 && env_ok) {
      constexpr int SB = 16;
      for (int tb = 0; tb < TT; tb += SB) {
        carry_t b8[SB];
        uint8_t s8[SB];
#pragma unroll
        for (int q = 0; q < SB; ++q) {
          b8[q] = s_b[tb + q][lane];
          s8[q] = s_sel[tb + q][lane];
        }
#pragma unroll
        for (int q = 0; q < SB; ++q) {
          carry_t nxt;
          if (MODE == OLY_SCAN_RETURN)
            nxt = (carry_t)(gamma * (double)carry + (double)b8[q]);
          else
            nxt = (carry_t)((float)b8[q] + gl32 * (float)carry);
          carry = s8[q] ? b8[q] : nxt;
          b8[q] = carry;
        }
#pragma unroll
        for (int q = 0; q < SB; ++q) s_b[tb + q][lane] = b8[q];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const int tt = rr + 16 * j;
      const int t = t_top - tt;
      if (t >= 0 && col_ok) {
        const size_t e = (size_t)t * N + n0 + 4 * c4;
        const float4 v4 = *reinterpret_cast<const float4*>(&s_v[tt][4 * c4]);
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
        float ro[4], ao[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const carry_t c = s_b[tt][4 * c4 + q];
          if (MODE == OLY_SCAN_RETURN) {
            ro[q] = (float)c;
            ao[q] = ro[q] - vv[q];
          } else {
            ao[q] = (float)c;
            ro[q] = ao[q] + vv[q];
          }
        }
        *reinterpret_cast<float4*>(ret + e) = make_float4(ro[0], ro[1], ro[2], ro[3]);
        *reinterpret_cast<float4*>(adv + e) = make_float4(ao[0], ao[1], ao[2], ao[3]);
      }
    }
  }
}

static bool wide_ok(const void* a, const void* b, const void* c, const void* d, const void* e, const void* f) {
  auto al = [](const void* p, uintptr_t m) { return (reinterpret_cast<uintptr_t>(p) & m) == 0; };
  return al(a, 15) && al(b, 15) && al(c, 15) && al(d, 3) && al(e, 15) && al(f, 15);
}

extern "C" int oly_return_scan(oly_ctx* ctx, int mode, int T, int N, double gamma, double lam,
                               const float* rew, const float* val, const float* next_val,
                               const uint8_t* flags, float* ret, float* adv, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (T < 0 || N < 0) OLY_FAIL(ctx, OLY_EINVAL, "oly_return_scan: negative T or N");
  if (T == 0 || N == 0) return OLY_OK;
  if (!rew || !val || !next_val || !flags || !ret || !adv)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_return_scan: NULL pointer");
  static const int variant = [] { const char* e = getenv("OLY_K6_VARIANT"); return e ? atoi(e) : 1; }();  // 0 chunk, 1 auto (lean if possible, else tile), 2 wide, 3 tile
  dim3 grid((N + 63) / 64);
  if (mode != OLY_SCAN_RETURN && mode != OLY_SCAN_GAE)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_return_scan: unknown mode %d", mode);
  if (variant == 0) {  // single-wave register-chunk kernel
    static const int epw_env = [] { const char* e = getenv("OLY_K6_EPW"); return e ? atoi(e) : 0; }();
    int epw = 64;
    while (epw > 8 && (long)((N + epw - 1) / epw) < 4L * ctx->num_cu) epw >>= 1;  // >= 4 waves per CU if possible
    if (epw_env > 0) epw = epw_env;
    dim3 g0((N + epw - 1) / epw);
    if (mode == OLY_SCAN_RETURN)
      hipLaunchKernelGGL(scan_kernel<OLY_SCAN_RETURN>, g0, dim3(64), 0, oly_s(stream), T, N, epw, gamma, lam,
                         rew, val, next_val, flags, ret, adv);
    else
      hipLaunchKernelGGL(scan_kernel<OLY_SCAN_GAE>, g0, dim3(64), 0, oly_s(stream), T, N, epw, gamma, lam, rew,
                         val, next_val, flags, ret, adv);
  } else if (variant == 1 && N % 4 == 0 && wide_ok(rew, val, next_val, flags, ret, adv)) {
    if (mode == OLY_SCAN_RETURN)
      hipLaunchKernelGGL(scan_lean_kernel<OLY_SCAN_RETURN>, grid, dim3(SCAN_THREADS), 0, oly_s(stream), T, N,
                         gamma, lam, rew, val, next_val, flags, ret, adv);
    else
      hipLaunchKernelGGL(scan_lean_kernel<OLY_SCAN_GAE>, grid, dim3(SCAN_THREADS), 0, oly_s(stream), T, N,
                         gamma, lam, rew, val, next_val, flags, ret, adv);
  } else if (variant == 2 && N % 4 == 0 && wide_ok(rew, val, next_val, flags, ret, adv)) {
    if (mode == OLY_SCAN_RETURN)
      hipLaunchKernelGGL(scan_wide_kernel<OLY_SCAN_RETURN>, grid, dim3(SCAN_THREADS), 0, oly_s(stream), T, N,
                         gamma, lam, rew, val, next_val, flags, ret, adv);
    else
      hipLaunchKernelGGL(scan_wide_kernel<OLY_SCAN_GAE>, grid, dim3(SCAN_THREADS), 0, oly_s(stream), T, N,
                         gamma, lam, rew, val, next_val, flags, ret, adv);
  } else {
    if (mode == OLY_SCAN_RETURN)
      hipLaunchKernelGGL(scan_tile_kernel<OLY_SCAN_RETURN>, grid, dim3(SCAN_THREADS), 0, oly_s(stream), T, N,
                         gamma, lam, rew, val, next_val, flags, ret, adv);
    else
      hipLaunchKernelGGL(scan_tile_kernel<OLY_SCAN_GAE>, grid, dim3(SCAN_THREADS), 0, oly_s(stream), T, N,
                         gamma, lam, rew, val, next_val, flags, ret, adv);
  }
  OLY_LAUNCH_CHECK(ctx, "scan_kernel");
  return OLY_OK;
}
