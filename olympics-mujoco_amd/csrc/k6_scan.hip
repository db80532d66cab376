// K6: reverse return / advantage scan over a [T,N] rollout block.
//   OLY_SCAN_RETURN  PPOBuffer.finish_path            rl/algos/ppo.py:68-84 (+ :335 adv)
//   OLY_SCAN_GAE     mushroom_rl compute_gae          call site gail_TRPO.py:126-127
//
// The recurrence is sequential in t and must round exactly like the reference (no
// re-association), so one lane owns one environment and walks t = T-1 .. 0; the parallelism is
// across environments only.  Two kernels, same arithmetic in the same order (bit-identical):
//   scan_pipe_kernel  (default) chain wave + mover waves, software-pipelined through LDS
//   scan_tile_kernel  scalar-load fallback for unaligned buffers / N % 4 != 0
// Bound: the serial fp chain (T dependent mul+add+select per env) at small N, HBM at large N;
// 17-21 B per element.
#include <cstdlib>
#include <type_traits>

#include "oly_common.h"

namespace {

// ---------------------------------------------------------------------------------------
// Fallback: a workgroup of 4 waves owns 64 environments.  ALL waves prefetch the next TT-step
// tile of rew/val/next_val/flags with scalar loads (any alignment, any N) while wave 0 runs the
// sequential recurrence over the current tile out of LDS.
// ---------------------------------------------------------------------------------------
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
typedef double nt_f64x2 __attribute__((ext_vector_type(2)));
// streaming (non-temporal) 16-byte accesses of the pipelined kernel's movers: every byte is touched once
__device__ __forceinline__ float4 nt_load4(const float* p) {
  const nt_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ double2 nt_load2(const double* p) {
  const nt_f64x2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f64x2*>(p));
  return make_double2(v.x, v.y);
}
__device__ __forceinline__ void nt_store4(float* p, float a, float b, float c, float d) {
  nt_f32x4 v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<nt_f32x4*>(p));
}

constexpr int TT = 64;          // time steps per tile
constexpr int SCAN_THREADS = 256;

template <int MODE, bool REW64>
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_kernel(int T, int N, double gamma, double lam,
                                                                 const void* __restrict__ rew_,
                                                                 const float* __restrict__ val,
                                                                 const float* __restrict__ next_val,
                                                                 const uint8_t* __restrict__ flags,
                                                                 float* __restrict__ ret,
                                                                 float* __restrict__ adv) {
  using rew_t = typename std::conditional<REW64, double, float>::type;
  const rew_t* __restrict__ rew = static_cast<const rew_t*>(rew_);
  __shared__ rew_t s_r[TT][64];
  __shared__ float s_v[TT][64], s_nv[TT][64];
  __shared__ uint8_t s_f[TT][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const bool env_ok = n < N;
  constexpr int RPW = TT / 4;  // rows per wave per tile
  rew_t pr[RPW];
  float pv[RPW], pnv[RPW];
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
        const rew_t r = s_r[tt][lane];
        const float v = s_v[tt][lane], nv = s_nv[tt][lane];
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
            a = (float)r - v;
            if (!(f & OLY_FLAG_ABSORBING)) a += g32 * nv;
          } else {
            a = (float)r + g32 * nv - v + gl32 * a_next;
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

// ---------------------------------------------------------------------------------------------
// Pipelined scan: the serial chain (wave 0) never waits for memory.  Waves 1-3 ("movers") load
// tiles two ahead into registers, turn tile k+1 into the recurrence constants in LDS, and derive
// and store ret/adv of tile k-1, all while wave 0 runs the chain of tile k.  Three LDS buffers,
// one barrier per tile.  Same arithmetic and order as scan_lean_kernel (bit-exact).
// ---------------------------------------------------------------------------------------------
// REW64: rewards arrive as float64 (what env.step returns in the reference; RETURN mode only).
// stats_ws != NULL: the movers also accumulate sum / sum of squares of the advantages they store
// (fp64, fixed order) and the workgroup leaves one partial pair at stats_ws[2 * env_block].
template <int MODE, bool REW64, int NTHREADS, int DEPTH, int EPW, int PT>   // EPW envs per workgroup, PT steps per tile
__global__ __launch_bounds__(NTHREADS) void scan_pipe_kernel(int T, int N, double gamma, double lam,
                                                                 const void* __restrict__ rew_,
                                                                 const float* __restrict__ val,
                                                                 const float* __restrict__ next_val,
                                                                 const uint8_t* __restrict__ flags,
                                                                 float* __restrict__ ret,
                                                                 float* __restrict__ adv,
                                                                 double* __restrict__ stats_ws) {
  using carry_t = typename std::conditional<MODE == OLY_SCAN_RETURN, double, float>::type;
  using rew_t = typename std::conditional<REW64, double, float>::type;
  const rew_t* __restrict__ rew = static_cast<const rew_t*>(rew_);
  constexpr int GPR = EPW / 4;               // float4 groups per tile row
  constexpr int PIPE_GROUPS = PT * GPR;
  constexpr int PIPE_MOVERS = NTHREADS - 64;
  constexpr int PIPE_SL = (PIPE_GROUPS + PIPE_MOVERS - 1) / PIPE_MOVERS;
  extern __shared__ __attribute__((aligned(16))) unsigned char pipe_lds[];
  typedef carry_t (*b_t)[PT][EPW];
  typedef float (*v_t)[PT][EPW];
  typedef uint8_t (*s_t)[PT][EPW];
  b_t s_b = reinterpret_cast<b_t>(pipe_lds);
  v_t s_v = reinterpret_cast<v_t>(pipe_lds + 3 * sizeof(carry_t) * PT * EPW);
  s_t s_sel = reinterpret_cast<s_t>(pipe_lds + 3 * (sizeof(carry_t) + sizeof(float)) * PT * EPW);

  const int tid = threadIdx.x, lane = tid & 63;
  const bool chain_wave = tid < 64;
  const int m = tid - 64;                    // mover index
  // XCD-aware placement: workgroups b and b + 8 share an XCD (and its L2), so give each XCD a
  // contiguous range of environment blocks: the 64-B flag sectors that 16-env neighbours share are
  // then fetched once per XCD instead of once per workgroup.  Speed only, any placement is correct.
  const int nblk = gridDim.x;
  const int eb = (nblk % 8 == 0) ? (int)(blockIdx.x % 8) * (nblk / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int n0 = eb * EPW;
  const bool env_ok = lane < EPW && n0 + lane < N;
  const int ntiles = (T + PT - 1) / PT;
  const float g32 = (float)gamma;
  const float gl32 = (float)(gamma * lam);

  struct Regs { rew_t r[PIPE_SL][4]; float4 v[PIPE_SL], nv[PIPE_SL]; uchar4 f[PIPE_SL]; };
  Regs R[DEPTH];   // tile k travels in R[k % DEPTH]; statically indexed everywhere

  auto load = [&](int k, Regs& R) {
    const int t_top = T - 1 - k * PT;
#pragma unroll
    for (int j = 0; j < PIPE_SL; ++j) {
      const int g = m + PIPE_MOVERS * j;
      const int tt = g / GPR, c4 = g % GPR;
      const int t = t_top - tt;
      if (g < PIPE_GROUPS && t >= 0 && n0 + 4 * c4 < N) {
        const size_t e = (size_t)t * N + n0 + 4 * c4;
        if (REW64) {
          const double2 lo = nt_load2(reinterpret_cast<const double*>(rew + e));
          const double2 hi = nt_load2(reinterpret_cast<const double*>(rew + e + 2));
          R.r[j][0] = (rew_t)lo.x; R.r[j][1] = (rew_t)lo.y; R.r[j][2] = (rew_t)hi.x; R.r[j][3] = (rew_t)hi.y;
        } else {
          const float4 r4 = nt_load4(reinterpret_cast<const float*>(rew + e));
          R.r[j][0] = (rew_t)r4.x; R.r[j][1] = (rew_t)r4.y; R.r[j][2] = (rew_t)r4.z; R.r[j][3] = (rew_t)r4.w;
        }
        R.v[j] = nt_load4(val + e);
        R.nv[j] = nt_load4(next_val + e);
        R.f[j] = *reinterpret_cast<const uchar4*>(flags + e);
      }
    }
  };
  auto konst = [&](rew_t r, float v, float nv, uint8_t f, bool top) -> carry_t {
    const bool last = (f & OLY_FLAG_LAST) || top, ab = f & OLY_FLAG_ABSORBING;
    if (MODE == OLY_SCAN_RETURN) {
      if (last) {
        const float p = g32 * (ab ? 0.f : nv);
        return (carry_t)((double)p + (double)r);
      }
      return (carry_t)(double)r;
    }
    if (last) {
      float a = (float)r - v;
      if (!ab) a += g32 * nv;
      return (carry_t)a;
    }
    return (carry_t)((float)r + g32 * nv - v);
  };
  auto precompute = [&](int k, const Regs& R) {
    const int buf = k % 3;
    const int t_top = T - 1 - k * PT;
#pragma unroll
    for (int j = 0; j < PIPE_SL; ++j) {
      const int g = m + PIPE_MOVERS * j;
      const int tt = g / GPR, c4 = g % GPR;
      if (g >= PIPE_GROUPS) continue;
      const bool top = (t_top - tt) == T - 1;
      const rew_t rv[4] = {R.r[j][0], R.r[j][1], R.r[j][2], R.r[j][3]};
      const float vv[4] = {R.v[j].x, R.v[j].y, R.v[j].z, R.v[j].w};
      const float nn[4] = {R.nv[j].x, R.nv[j].y, R.nv[j].z, R.nv[j].w};
      const uint8_t ff[4] = {R.f[j].x, R.f[j].y, R.f[j].z, R.f[j].w};
      uchar4 sel;
      carry_t bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = konst(rv[q], vv[q], nn[q], ff[q], top);
      sel.x = ((ff[0] & OLY_FLAG_LAST) || top) ? 1 : 0;
      sel.y = ((ff[1] & OLY_FLAG_LAST) || top) ? 1 : 0;
      sel.z = ((ff[2] & OLY_FLAG_LAST) || top) ? 1 : 0;
      sel.w = ((ff[3] & OLY_FLAG_LAST) || top) ? 1 : 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) s_b[buf][tt][4 * c4 + q] = bq[q];
      *reinterpret_cast<uchar4*>(&s_sel[buf][tt][4 * c4]) = sel;
      *reinterpret_cast<float4*>(&s_v[buf][tt][4 * c4]) = R.v[j];
    }
  };
  double acc_s = 0.0, acc_ss = 0.0;
  auto store = [&](int k) {
    const int buf = k % 3;
    const int t_top = T - 1 - k * PT;
#pragma unroll
    for (int j = 0; j < PIPE_SL; ++j) {
      const int g = m + PIPE_MOVERS * j;
      const int tt = g / GPR, c4 = g % GPR;
      const int t = t_top - tt;
      if (g < PIPE_GROUPS && t >= 0 && n0 + 4 * c4 < N) {
        const size_t e = (size_t)t * N + n0 + 4 * c4;
        const float4 v4 = *reinterpret_cast<const float4*>(&s_v[buf][tt][4 * c4]);
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
        float ro[4], ao[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const carry_t c = s_b[buf][tt][4 * c4 + q];
          if (MODE == OLY_SCAN_RETURN) {
            ro[q] = (float)c;
            ao[q] = ro[q] - vv[q];
          } else {
            ao[q] = (float)c;
            ro[q] = ao[q] + vv[q];
          }
        }
        nt_store4(ret + e, ro[0], ro[1], ro[2], ro[3]);
        nt_store4(adv + e, ao[0], ao[1], ao[2], ao[3]);
        if (stats_ws) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double a = (double)ao[q];
            acc_s += a;
            acc_ss += a * a;
          }
        }
      }
    }
  };

  carry_t carry = 0;
  // (issuing the LDS reads of batch n+1 ahead of batch n's chain was measured: slower)
  auto chain = [&](int k) {
    const int buf = k % 3;
    const int steps = min(PT, T - k * PT);
    constexpr int SB = 16;
    for (int tb = 0; tb < steps; tb += SB) {
      carry_t b8[SB];
      uint8_t s8[SB];
#pragma unroll
      for (int q = 0; q < SB; ++q) {
        b8[q] = s_b[buf][tb + q][lane];
        s8[q] = s_sel[buf][tb + q][lane];
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
      for (int q = 0; q < SB; ++q) s_b[buf][tb + q][lane] = b8[q];
    }
  };
  // iteration i: wave 0 chains tile i | movers store tile i-1, stage tile i+1 (registers `cur`
  // hold it), then refill `cur` with tile i+1+DEPTH
  auto iter = [&](int i, Regs& cur) {
    if (chain_wave) {
      if (env_ok) chain(i);
    } else {
      if (i >= 1) store(i - 1);
      if (i + 1 < ntiles) precompute(i + 1, cur);
      if (i + 1 + DEPTH < ntiles) load(i + 1 + DEPTH, cur);
    }
    __syncthreads();
  };

  if (!chain_wave) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (d < ntiles) load(d, R[d]);
    precompute(0, R[0]);
    if (DEPTH < ntiles) load(DEPTH, R[0]);
  }
  __syncthreads();
  for (int i = 0; i < ntiles; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (i + d < ntiles) iter(i + d, R[(d + 1) % DEPTH]);
  }
  if (!chain_wave) store(ntiles - 1);
  if (stats_ws) {
    // fixed order: lane tree inside each wave, then waves 0..3 in order (the chain wave adds zeros)
    double* sh = reinterpret_cast<double*>(pipe_lds);
    __syncthreads();   // LDS tiles are dead from here on
    const double ws_ = wave_sum(acc_s), wss_ = wave_sum(acc_ss);
    if (lane == 0) { sh[2 * (tid >> 6)] = ws_; sh[2 * (tid >> 6) + 1] = wss_; }
    __syncthreads();
    if (tid == 0) {
      double ts = 0.0, tss = 0.0;
      for (int i = 0; i < NTHREADS / 64; ++i) { ts += sh[2 * i]; tss += sh[2 * i + 1]; }
      stats_ws[2 * eb] = ts;
      stats_ws[2 * eb + 1] = tss;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Wide shapes (N >= 128 environments per CU): one LANE per environment, no LDS, no barriers.
// With thousands of waves the occupancy hides the latency that the pipelined kernel needs mover
// waves for, and nothing waits for a workgroup's chain wave.  Per lane a software pipeline over
// chunks of LU steps: flags two chunks ahead, rewards / values one chunk ahead, and next_val
// ONLY where the flags say it is read (RETURN mode: cut steps that are not terminal): every lane
// issues the load, lanes that do not need it present an out-of-range buffer offset, which costs no
// memory traffic.  All loads and stores are branch-free buffer operations (range-checked by the
// hardware), so the loop is straight-line code and the waits are counted (vmcnt(n)), not drained.
// Same arithmetic, same order as the other two kernels: bit-identical results.
// ---------------------------------------------------------------------------------------------
constexpr int LANE_THREADS = 256;
constexpr int LU = 8;
// cache policy of the lane kernel's buffer operations: nt (2).  Every byte is touched once; measured against
// the default policy at [400, 262144]: 0.344 -> 0.319 ms (f32 rewards), 0.442 -> 0.397 ms (f64 + statistics)
constexpr int LANE_AUX = 2;

template <int MODE, bool REW64>
__global__ __launch_bounds__(LANE_THREADS) void scan_lane_kernel(int T, int N, double gamma, double lam,
                                                                 const void* __restrict__ rew_,
                                                                 const float* __restrict__ val,
                                                                 const float* __restrict__ next_val,
                                                                 const uint8_t* __restrict__ flags,
                                                                 float* __restrict__ ret, float* __restrict__ adv,
                                                                 double* __restrict__ stats_ws) {
  using rew_t = typename std::conditional<REW64, double, float>::type;
  constexpr unsigned OOB = 0x80000000u;     // beyond every num_records (the host keeps the buffers below 2 GiB)
  const size_t total = (size_t)T * N;
  const __amdgpu_buffer_rsrc_t rs_rew = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(rew_), 0, (int)(total * sizeof(rew_t)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_val = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(val), 0, (int)(total * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_nv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(next_val), 0, (int)(total * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_fl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(flags), 0, (int)total, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ret = __builtin_amdgcn_make_buffer_rsrc(ret, 0, (int)(total * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_adv = __builtin_amdgcn_make_buffer_rsrc(adv, 0, (int)(total * 4), 0x00020000);
  const int n = blockIdx.x * LANE_THREADS + threadIdx.x;
  const bool env_ok = n < N;
  const int nchunks = (T + LU - 1) / LU;
  const float g32 = (float)gamma;
  const float gl32 = (float)(gamma * lam);

  // element index of step u of chunk c (t = T-1 - (c*LU + u)), OOB for steps before 0 / envs beyond N
  auto elem = [&](int c, int u, bool& ok) -> unsigned {
    const int t = T - 1 - (c * LU + u);
    ok = env_ok && t >= 0;
    return (unsigned)t * (unsigned)N + (unsigned)n;
  };
  struct Main { rew_t r[LU]; float v[LU], nv[LU]; unsigned fl[LU]; };   // fl: the chunk's flags, so the flag registers are free again
  auto load_flags = [&](int c, unsigned (&f)[LU]) {
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      bool ok;
      const unsigned e = elem(c, u, ok);
      unsigned x = __builtin_amdgcn_raw_buffer_load_b8(rs_fl, ok ? e : OOB, 0, LANE_AUX);
      if (T - 1 - (c * LU + u) == T - 1) x |= OLY_FLAG_LAST;          // the block end cuts every environment
      f[u] = x;
    }
  };
  auto load_main = [&](int c, const unsigned (&f)[LU], Main& m) {
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      bool ok;
      const unsigned e = elem(c, u, ok);
      if (REW64) {
        m.r[u] = (rew_t)__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs_rew, ok ? e * 8u : OOB, 0, LANE_AUX));
      } else {
        m.r[u] = (rew_t)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_rew, ok ? e * 4u : OOB, 0, LANE_AUX));
      }
      m.v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_val, ok ? e * 4u : OOB, 0, LANE_AUX));
      m.fl[u] = f[u];
      const bool need_nv = MODE == OLY_SCAN_GAE ? true : ((f[u] & OLY_FLAG_LAST) && !(f[u] & OLY_FLAG_ABSORBING));
      m.nv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_nv, (ok && need_nv) ? e * 4u : OOB, 0, LANE_AUX));
    }
  };
  double R = 0.0;
  float a_next = 0.f;
  double acc_s = 0.0, acc_ss = 0.0;
  auto compute = [&](int c, const Main& m) {
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      bool ok;
      const unsigned e = elem(c, u, ok);
      const bool last = (m.fl[u] & OLY_FLAG_LAST) != 0, ab = (m.fl[u] & OLY_FLAG_ABSORBING) != 0;
      float rt, at;
      if (MODE == OLY_SCAN_RETURN) {
        const float p = g32 * ((last && !ab) ? m.nv[u] : 0.f);
        const double cut = (double)p + (double)m.r[u];
        const double run = gamma * R + (double)m.r[u];
        const double Rn = last ? cut : run;
        R = ok ? Rn : R;
        rt = (float)Rn;
        at = rt - m.v[u];
      } else {
        float a_cut = (float)m.r[u] - m.v[u];
        if (!ab) a_cut += g32 * m.nv[u];
        const float a_run = (float)m.r[u] + g32 * m.nv[u] - m.v[u] + gl32 * a_next;
        at = last ? a_cut : a_run;
        a_next = ok ? at : a_next;
        rt = at + m.v[u];
      }
      const unsigned off = ok ? e * 4u : OOB;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rt), rs_ret, off, 0, LANE_AUX);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, at), rs_adv, off, 0, LANE_AUX);
      if (stats_ws && ok) {
        const double a = (double)at;
        acc_s += a;
        acc_ss += a * a;
      }
    }
  };

  unsigned fA[LU], fB[LU];
  Main mA, mB;
  // Issue order per chunk c: main(c+1) [needs flags(c+1), issued one chunk ago], flags(c+2), then the
  // arithmetic and the stores of chunk c [needs main(c), issued one chunk ago].  In-order completion makes
  // every wait a counted one: at most 56 younger operations stay in flight (vmcnt holds 63).
  load_flags(0, fA);
  load_flags(1, fB);
  load_main(0, fA, mA);
  for (int c = 0; c < nchunks; c += 2) {
    load_main(c + 1, fB, mB);       // flags(c+1) -> fB consumed (copied into mB.fl)
    load_flags(c + 2, fA);          // fA held flags(c): already copied into mA.fl
    compute(c, mA);
    load_main(c + 2, fA, mA);
    load_flags(c + 3, fB);
    compute(c + 1, mB);
  }
  if (stats_ws) {
    __shared__ double sh[2 * LANE_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double ws_ = wave_sum(acc_s), wss_ = wave_sum(acc_ss);
    if (lane == 0) { sh[2 * w] = ws_; sh[2 * w + 1] = wss_; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double ts = 0.0, tss = 0.0;
      for (int i = 0; i < LANE_THREADS / 64; ++i) { ts += sh[2 * i]; tss += sh[2 * i + 1]; }
      stats_ws[2 * blockIdx.x] = ts;
      stats_ws[2 * blockIdx.x + 1] = tss;
    }
  }
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void rollout_cuts_kernel(int N, int max_traj_len, int last_step,
                                                           const uint8_t* __restrict__ done,
                                                           int* __restrict__ traj_len,
                                                           uint8_t* __restrict__ flags,
                                                           int* __restrict__ n_cut) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  bool cut = false;
  if (n < N) {
    const int len = traj_len[n] + 1;
    const bool d = done[n] != 0;
    cut = d || len >= max_traj_len || last_step != 0;
    flags[n] = (uint8_t)((cut ? OLY_FLAG_LAST : 0) | (d ? OLY_FLAG_ABSORBING : 0));
    traj_len[n] = cut ? 0 : len;
  }
  const int c = __popcll(__ballot(cut));
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(n_cut, c);   // integer count: order-independent
}
}  // namespace

extern "C" int oly_rollout_cuts(oly_ctx* ctx, int N, int max_traj_len, int last_step, const uint8_t* done,
                                int32_t* traj_len, uint8_t* flags, int32_t* n_cut, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (N < 0 || !n_cut || (N > 0 && (!done || !traj_len || !flags)))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_rollout_cuts: bad argument");
  OLY_HIP(ctx, hipMemsetAsync(n_cut, 0, sizeof(int32_t), oly_s(stream)));
  if (N == 0) return OLY_OK;
  hipLaunchKernelGGL(rollout_cuts_kernel, dim3((N + 255) / 256), dim3(256), 0, oly_s(stream), N, max_traj_len,
                     last_step, done, traj_len, flags, n_cut);
  OLY_LAUNCH_CHECK(ctx, "rollout_cuts_kernel");
  return OLY_OK;
}

static bool wide_ok(const void* a, const void* b, const void* c, const void* d, const void* e, const void* f) {
  auto al = [](const void* p, uintptr_t m) { return (reinterpret_cast<uintptr_t>(p) & m) == 0; };
  return al(a, 15) && al(b, 15) && al(c, 15) && al(d, 3) && al(e, 15) && al(f, 15);
}

extern "C" int oly_return_scan_stats(oly_ctx* ctx, int mode_flags, int T, int N, double gamma, double lam,
                                     const void* rew, const float* val, const float* next_val,
                                     const uint8_t* flags, float* ret, float* adv, double* stats3_out,
                                     oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (T < 0 || N < 0) OLY_FAIL(ctx, OLY_EINVAL, "oly_return_scan: negative T or N");
  const int mode = mode_flags & 0xff;
  const bool rew64 = (mode_flags & OLY_SCAN_REW_F64) != 0;
  if (mode != OLY_SCAN_RETURN && mode != OLY_SCAN_GAE)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_return_scan: unknown mode %d", mode);
  if (rew64 && mode != OLY_SCAN_RETURN)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_return_scan: OLY_SCAN_REW_F64 is defined for OLY_SCAN_RETURN only");
  if (T == 0 || N == 0) {
    if (stats3_out) return oly_adv_stats(ctx, 0, nullptr, stats3_out, stream);
    return OLY_OK;
  }
  if (!rew || !val || !next_val || !flags || !ret || !adv)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_return_scan: NULL pointer");
  static const int variant = [] { const char* e = getenv("OLY_K6_VARIANT"); return e ? atoi(e) : 1; }();  // 1 auto, 3 force the fallback
  static const int pipe_cfg = [] { const char* e = getenv("OLY_K6_PIPE"); return e ? atoi(e) : 0; }();   // 0 auto; 7 forces the lane kernel
  // the lane-per-environment kernel addresses its buffers through 32-bit buffer offsets: each below 2 GiB
  const bool fits32 = (size_t)T * (size_t)N * (rew64 ? 8 : 4) < ((size_t)1 << 31);
  // measured on MI355X (tools/time_k6_wide.py): RETURN mode gains 17-32 % from N = 256 environments per CU up
  // (next_val is fetched at the cut steps only); GAE reads next_val everywhere and stays on the pipelined kernel
  if (variant == 1 && fits32 &&
      (pipe_cfg == 7 || (pipe_cfg == 0 && mode == OLY_SCAN_RETURN && N >= 256 * ctx->num_cu))) {
    const int nblocks = (N + LANE_THREADS - 1) / LANE_THREADS;
    double* ws = (stats3_out && (size_t)nblocks * 2 * sizeof(double) <= ctx->stats_ws_bytes) ? ctx->stats_ws : nullptr;
    dim3 g(nblocks), b(LANE_THREADS);
    if (mode == OLY_SCAN_RETURN && rew64)
      hipLaunchKernelGGL((scan_lane_kernel<OLY_SCAN_RETURN, true>), g, b, 0, oly_s(stream), T, N, gamma, lam, rew, val,
                         next_val, flags, ret, adv, ws);
    else if (mode == OLY_SCAN_RETURN)
      hipLaunchKernelGGL((scan_lane_kernel<OLY_SCAN_RETURN, false>), g, b, 0, oly_s(stream), T, N, gamma, lam, rew, val,
                         next_val, flags, ret, adv, ws);
    else
      hipLaunchKernelGGL((scan_lane_kernel<OLY_SCAN_GAE, false>), g, b, 0, oly_s(stream), T, N, gamma, lam, rew, val,
                         next_val, flags, ret, adv, ws);
    OLY_LAUNCH_CHECK(ctx, "scan_lane_kernel");
    if (stats3_out) {
      if (ws) return oly_stats_finish(ctx, nblocks, (int64_t)T * N, stats3_out, stream);
      return oly_adv_stats(ctx, (int64_t)T * N, adv, stats3_out, stream);
    }
    return OLY_OK;
  }
  if (variant == 1 && N % 4 == 0 && wide_ok(rew, val, next_val, flags, ret, adv)) {
    // three LDS buffers of [PT][EPW] (constant fp64/fp32 + value f32 + select u8)
    int nblocks = 0;
    double* ws = nullptr;
#define OLY_PIPE_LAUNCH(ID, NT, DP, EPW, PTT)                                                            \
  do {                                                                                                   \
    auto kr = scan_pipe_kernel<OLY_SCAN_RETURN, false, NT, DP, EPW, PTT>;                                \
    auto kd = scan_pipe_kernel<OLY_SCAN_RETURN, true, NT, DP, EPW, PTT>;                                 \
    auto kg = scan_pipe_kernel<OLY_SCAN_GAE, false, NT, DP, EPW, PTT>;                                   \
    const size_t lds_r = 3 * (sizeof(double) + sizeof(float) + 1) * PTT * EPW;                           \
    const size_t lds_g = 3 * (sizeof(float) + sizeof(float) + 1) * PTT * EPW;                            \
    if (!(ctx->scan_attr_done & (1u << ID))) {                                                           \
      OLY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kr),                                \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));         \
      OLY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kd),                                \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));         \
      OLY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kg),                                \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g));         \
      ctx->scan_attr_done |= (1u << ID);                                                                 \
    }                                                                                                    \
    dim3 g((N + EPW - 1) / EPW);                                                                         \
    nblocks = (int)g.x;                                                                                  \
    if (stats3_out && (size_t)nblocks * 2 * sizeof(double) <= ctx->stats_ws_bytes) ws = ctx->stats_ws;   \
    if (mode == OLY_SCAN_RETURN && rew64)                                                                \
      hipLaunchKernelGGL(kd, g, dim3(NT), lds_r, oly_s(stream), T, N, gamma, lam, rew, val, next_val,    \
                         flags, ret, adv, ws);                                                           \
    else if (mode == OLY_SCAN_RETURN)                                                                    \
      hipLaunchKernelGGL(kr, g, dim3(NT), lds_r, oly_s(stream), T, N, gamma, lam, rew, val, next_val,    \
                         flags, ret, adv, ws);                                                           \
    else                                                                                                 \
      hipLaunchKernelGGL(kg, g, dim3(NT), lds_g, oly_s(stream), T, N, gamma, lam, rew, val, next_val,    \
                         flags, ret, adv, ws);                                                           \
  } while (0)
    // few environments: small workgroups so that every CU runs a chain; many: wide rows
    int cfg = pipe_cfg;
    if (cfg == 0) cfg = (N >= 128 * ctx->num_cu) ? 1 : (N >= 64 * ctx->num_cu ? 2 : 3);
    switch (cfg) {
      case 1: OLY_PIPE_LAUNCH(1, 256, 2, 64, 32); break;   // [400,32768]: 50 us (4.4 TB/s)
      case 2: OLY_PIPE_LAUNCH(2, 256, 2, 32, 64); break;
      default: OLY_PIPE_LAUNCH(3, 256, 2, 16, 32); break;  // [400,4096]: 16 / 14 us
    }
#undef OLY_PIPE_LAUNCH
    OLY_LAUNCH_CHECK(ctx, "scan_pipe_kernel");
    if (stats3_out) {
      if (ws) return oly_stats_finish(ctx, nblocks, (int64_t)T * N, stats3_out, stream);
      return oly_adv_stats(ctx, (int64_t)T * N, adv, stats3_out, stream);
    }
    return OLY_OK;
  }
  dim3 grid((N + 63) / 64);
  if (mode == OLY_SCAN_RETURN && rew64)
    hipLaunchKernelGGL((scan_tile_kernel<OLY_SCAN_RETURN, true>), grid, dim3(SCAN_THREADS), 0, oly_s(stream), T,
                       N, gamma, lam, rew, val, next_val, flags, ret, adv);
  else if (mode == OLY_SCAN_RETURN)
    hipLaunchKernelGGL((scan_tile_kernel<OLY_SCAN_RETURN, false>), grid, dim3(SCAN_THREADS), 0, oly_s(stream), T,
                       N, gamma, lam, rew, val, next_val, flags, ret, adv);
  else
    hipLaunchKernelGGL((scan_tile_kernel<OLY_SCAN_GAE, false>), grid, dim3(SCAN_THREADS), 0, oly_s(stream), T, N,
                       gamma, lam, rew, val, next_val, flags, ret, adv);
  OLY_LAUNCH_CHECK(ctx, "scan_tile_kernel");
  if (stats3_out) return oly_adv_stats(ctx, (int64_t)T * N, adv, stats3_out, stream);
  return OLY_OK;
}

extern "C" int oly_return_scan(oly_ctx* ctx, int mode, int T, int N, double gamma, double lam,
                               const float* rew, const float* val, const float* next_val,
                               const uint8_t* flags, float* ret, float* adv, oly_stream stream) {
  return oly_return_scan_stats(ctx, mode, T, N, gamma, lam, rew, val, next_val, flags, ret, adv, nullptr,
                               stream);
}
