// K1 + K5: fused observation build / has-fallen / previous-obs reward / action scale+clamp
// for imitation-learning robots (UnitreeH1), over T*N (step, env) rows per launch.
//
// Replaces, per row (file:line under the reference tree):
//   mushroom ObservationHelper._build_obs driven by UnitreeH1.py:303-355
//   LocoEnvBase._create_observation            loco_env_base.py:737-767
//   BaseHumanoidRobot.is_absorbing             base_humanoid_robot.py:246-260
//   UnitreeH1._has_fallen                      UnitreeH1.py:162-203
//   LocoEnvBase.reward / TargetVelocityReward  loco_env_base.py:776-781, utils/reward.py:66-74
//   LocoEnvBase._preprocess_action             loco_env_base.py:1050-1069 (+ ctrlrange clamp)
//
// Bound: HBM.  Algorithmic bytes per row (H1, fp32 outputs): 2*17*8 + 11*4 in, 32*4 + 4 + 1 +
// 11*4 out (+ 8 B/env/launch carried state) = 493 B/row + fall_code 1 B when requested.
//
// Data movement: the physics host hands over AoS rows (qpos [R,nq] f64 ...).  A workgroup owns
// ROWS consecutive rows = one contiguous byte range of every input and output array, so
// every global access is a dense 16-B-per-lane stream: inputs are staged into LDS in their
// global order, the per-row gather / permutation happens on LDS reads, and outputs leave
// as float4 / double2 stores of the row-major output tile.
#include <cstdlib>

#include "oly_common.h"

namespace {

constexpr int THREADS = 256;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // one 16-B lane access

template <int NQ, int NV, int NGRF, int NACT, int NU, int NOBS>
struct StaticDims {
  static constexpr int nq = NQ, nv = NV, n_grf = NGRF, n_act = NACT, nu = NU, n_obs = NOBS;
  static constexpr bool is_static = true;
  static constexpr int n_obs_ct = NOBS;
  template <int ROWS, int TH>
  static constexpr int max_chunks() {
    return (ROWS * (NQ + NV + NGRF) / 2 + ROWS * NACT / 4 + TH - 1) / TH;
  }
  __device__ explicit StaticDims(const IlDev*) {}
};
struct DynDims {
  int nq, nv, n_grf, n_act, nu, n_obs;
  static constexpr bool is_static = false;
  static constexpr int n_obs_ct = 1;
  template <int ROWS, int TH>
  static constexpr int max_chunks() { return 1; }
  __device__ explicit DynDims(const IlDev* m)
      : nq(m->nq), nv(m->nv), n_grf(m->n_grf), n_act(m->n_act), nu(m->nu), n_obs(m->n_obs) {}
};

struct IlArgs {
  const IlDev* md;
  long R;  // T*N rows
  int N;
  const double* qpos;
  const double* qvel;
  const float* action;
  const double* grf;
  const double* prev_in;
  double* prev_out;
  void* obs;
  float* reward;
  uint8_t* absorbing;
  uint8_t* fall_code;
  void* ctrl;
  int fast;  // all base pointers 16-B aligned: dense 16-B staging / stores allowed
  long tile0;    // first tile of this launch (generic kernel) / number of full tiles (tile kernel)
};

// LDS carve (all offsets in bytes, every region 16-B aligned):
//   [ staged doubles: q | v | g ][ staged action floats ][ tables ]
template <int ROWS, class D>
struct Carve {
  int q, v, g, a, tab_off, tab_str, tab_csrc, tab_act, total;
  __host__ __device__ Carve(int nq, int nv, int n_grf, int n_act, int nu, int n_obs) {
    auto al = [](int x) { return (x + 15) & ~15; };
    q = 0;
    v = q + ROWS * nq * 8;
    g = v + ROWS * nv * 8;
    a = al(g + ROWS * n_grf * 8);
    tab_off = al(a + ROWS * n_act * 4);
    tab_str = al(tab_off + n_obs * 4);
    tab_csrc = al(tab_str + n_obs * 4);
    tab_act = al(tab_csrc + nu * 4);
    total = al(tab_act + 4 * nu * 8);
  }
};

#ifndef OLY_K1_NT
#define OLY_K1_NT 1  // bit 0: non-temporal tile loads (+17 % measured), bit 1: non-temporal stores (no gain)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 ld16(const u32x4* p) {
  return (OLY_K1_NT & 1) ? __builtin_nontemporal_load(p) : *p;
}
__device__ __forceinline__ void st16(f32x4* p, f32x4 v) {
  if (OLY_K1_NT & 2) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ void st16(f64x2* p, f64x2 v) {
  if (OLY_K1_NT & 2) __builtin_nontemporal_store(v, p); else *p = v;
}

#ifndef OLY_K1_ABLATE
#define OLY_K1_ABLATE 0  // diagnostic builds only: 1 = input stream only, 2 = no input loads
#endif

constexpr int FAST_FALL = 8;  // fall tests kept in registers (H1/Atlas/Talos have 4..7)

// One workgroup per tile of ROWS rows.  Everything that does not depend on the staged data
// (column tables, thresholds, per-lane source offsets) is fetched BEFORE the tile's loads are
// waited for, so that after the barrier the tile is turned into outputs with one LDS round
// trip per output vector; the long pole of a workgroup's life is then its input stream, and
// several resident workgroups per CU keep that stream saturated.
template <int ROWS, class D, bool OBS64, bool CTRL64>
__global__ __launch_bounds__(THREADS) void il_step_kernel(IlArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const IlDev* __restrict__ md = p.md;
  const D d(md);
  const Carve<ROWS, D> cv(d.nq, d.nv, d.n_grf, d.n_act, d.nu, d.n_obs);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const long row0 = (p.tile0 + blockIdx.x) * ROWS;
  const int rows = (p.R - row0 < ROWS) ? (int)(p.R - row0) : ROWS;
  const bool full = (rows == ROWS) && p.fast;

  double* sq = reinterpret_cast<double*>(lds + cv.q);
  double* sv = reinterpret_cast<double*>(lds + cv.v);
  double* sg = reinterpret_cast<double*>(lds + cv.g);
  float* sa = reinterpret_cast<float*>(lds + cv.a);
  int* t_off = reinterpret_cast<int*>(lds + cv.tab_off);  // column -> element offset of row 0
  int* t_str = reinterpret_cast<int*>(lds + cv.tab_str);  // column -> row stride (elements)
  int* t_cidx = reinterpret_cast<int*>(lds + cv.tab_csrc);      // actuator -> action slot | -1
  double* t_act = reinterpret_cast<double*>(lds + cv.tab_act);  // per ACTUATOR: mean|delta|lo|hi

  // staged-row addressing: q | v | g are contiguous in LDS, so element `sidx` of the staged
  // row [qpos | qvel | grf] of local row r lives at off(sidx) + r * stride(sidx)
  auto col_off = [&](int sidx) -> int {
    if (sidx < d.nq) return sidx;
    if (sidx < d.nq + d.nv) return ROWS * d.nq + (sidx - d.nq);
    return ROWS * (d.nq + d.nv) + (sidx - d.nq - d.nv);
  };
  auto col_str = [&](int sidx) -> int {
    if (sidx < d.nq) return d.nq;
    if (sidx < d.nq + d.nv) return d.nv;
    return d.n_grf;
  };
  const int first_grf = d.n_obs - d.n_grf;

  // ---- (1) issue the whole input tile: every 16-B load of the tile before any wait
  const int cq = ROWS * d.nq / 2, cvv = ROWS * d.nv / 2, cg = ROWS * d.n_grf / 2;
  const int ca = p.action ? ROWS * d.n_act / 4 : 0;
  const int cd = cq + cvv + cg;
  const int total = (OLY_K1_ABLATE & 2) ? 0 : cd + ca;
  u32x4* ld = reinterpret_cast<u32x4*>(sq);
  u32x4* la = reinterpret_cast<u32x4*>(sa);
  constexpr int MAXC = D::template max_chunks<ROWS, THREADS>();
  u32x4 regs[MAXC];
  const bool reg_staged = D::is_static && full;
  if (reg_staged) {
    const u32x4* gq = reinterpret_cast<const u32x4*>(p.qpos + row0 * d.nq);
    const u32x4* gv = reinterpret_cast<const u32x4*>(p.qvel + row0 * d.nv);
    const u32x4* gg = reinterpret_cast<const u32x4*>(p.grf ? p.grf + row0 * d.n_grf : nullptr);
    const u32x4* ga = reinterpret_cast<const u32x4*>(p.action ? p.action + row0 * d.n_act : nullptr);
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int g = tid + k * THREADS;
      if (g < total) {
        const u32x4* src = g < cq ? gq + g : (g < cq + cvv ? gv + (g - cq) : (g < cd ? gg + (g - cq - cvv) : ga + (g - cd)));
        regs[k] = *src;
      }
    }
  }

  // ---- (2) while those are in flight: tables and per-lane constants
  for (int i = tid; i < d.n_obs; i += THREADS) {
    const int sidx = md->src[i];
    t_off[i] = col_off(sidx);
    t_str[i] = col_str(sidx);
  }
  for (int j = tid; j < d.nu; j += THREADS) {
    const int k = md->ctrl_src[j];
    t_cidx[j] = k;
    const int kk = k < 0 ? 0 : k;
    t_act[j] = md->act_mean[kk];
    t_act[d.nu + j] = md->act_delta[kk];
    t_act[2 * d.nu + j] = md->ctrl_lo[kk];
    t_act[3 * d.nu + j] = md->ctrl_hi[kk];
  }
  // columns owned by this lane in the dense output sweep are loop-invariant when the sweep
  // stride is a multiple of the row width (H1: 256 lanes x 4 floats = 32 rows x 32 columns)
  constexpr int OW = OBS64 ? 2 : 4;  // observation values per lane store
  const bool fixed_cols = D::is_static && ((THREADS * OW) % d.n_obs == 0);
  int my_off[OW], my_str[OW];
  if (fixed_cols) {
    const int c0 = (tid * OW) % d.n_obs;
#pragma unroll
    for (int k = 0; k < OW; ++k) {
      const int sidx = md->src[c0 + k];
      my_off[k] = col_off(sidx);
      my_str[k] = col_str(sidx);
    }
  }
  // wave-uniform parameters of the per-row tests (scalar loads)
  const int nf = md->n_fall;
  int f_off[FAST_FALL], f_str[FAST_FALL];
  double f_lo[FAST_FALL], f_hi[FAST_FALL];
  bool f_grf[FAST_FALL];
#pragma unroll
  for (int k = 0; k < FAST_FALL; ++k) {
    const int kk = k < nf ? k : 0;
    const int sidx = md->fall_sidx[kk];
    f_off[k] = col_off(sidx);
    f_str[k] = col_str(sidx);
    f_lo[k] = md->fall_lo[kk];
    f_hi[k] = md->fall_hi[kk];
    f_grf[k] = d.n_grf > 0 && md->fall_idx[kk] >= first_grf;
  }
  const int rt = md->reward_type;
  const int rew_off = col_off(md->reward_sidx), rew_str = col_str(md->reward_sidx);
  const bool rew_grf = d.n_grf > 0 && md->reward_idx >= first_grf;
  const double tvel = md->target_velocity;
  const int use_abs = md->use_absorbing;
  const long gr = row0 + tid;  // global row of this lane in the per-row phase
  double prev0 = 0.0;
  if (tid < rows && rt != OLY_REWARD_NONE && gr < p.N) prev0 = p.prev_in[gr];

  // ---- (3) land the tile in LDS (global order == LDS order)
  if (reg_staged) {
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      const int g = tid + k * THREADS;
      if (g < total) {
        if (g < cd) ld[g] = regs[k];
        else la[g - cd] = regs[k];
      }
    }
  } else if (full) {
    const u32x4* gq = reinterpret_cast<const u32x4*>(p.qpos + row0 * d.nq);
    const u32x4* gv = reinterpret_cast<const u32x4*>(p.qvel + row0 * d.nv);
    const u32x4* gg = reinterpret_cast<const u32x4*>(p.grf ? p.grf + row0 * d.n_grf : nullptr);
    const u32x4* ga = reinterpret_cast<const u32x4*>(p.action ? p.action + row0 * d.n_act : nullptr);
#pragma unroll 8
    for (int g = tid; g < total; g += THREADS) {
      const u32x4* src = g < cq ? gq + g : (g < cq + cvv ? gv + (g - cq) : (g < cd ? gg + (g - cq - cvv) : ga + (g - cd)));
      const u32x4 x = *src;
      if (g < cd) ld[g] = x;
      else la[g - cd] = x;
    }
  } else {
    for (int i = tid; i < rows * d.nq; i += THREADS) sq[i] = p.qpos[row0 * d.nq + i];
    for (int i = tid; i < rows * d.nv; i += THREADS) sv[i] = p.qvel[row0 * d.nv + i];
    if (p.grf)
      for (int i = tid; i < rows * d.n_grf; i += THREADS) sg[i] = p.grf[row0 * d.n_grf + i];
    if (p.action)
      for (int i = tid; i < rows * d.n_act; i += THREADS) sa[i] = p.action[row0 * d.n_act + i];
  }
  __syncthreads();
  if (OLY_K1_ABLATE == 1) {  // diagnostic build: input stream only
    if (sq[tid] == 1.2345e300) p.reward[0] = 1.f;
    return;
  }

  // ---- (4) per-row scalars: fall tests, reward of the NEXT step, carried state, flags.
  // ROWS is a multiple of 64 and <= THREADS: a wave is either all-in or all-out.
  if (tid < ROWS && !(OLY_K1_ABLATE & 16)) {
    const int r = tid;
    unsigned code = 0, ab = 0;
    if (r < rows) {
      double fv[FAST_FALL];
#pragma unroll
      for (int k = 0; k < FAST_FALL; ++k)
        if (k < nf) fv[k] = sq[f_off[k] + r * f_str[k]];
#pragma unroll
      for (int k = 0; k < FAST_FALL; ++k)
        if (k < nf) {
          const double v = f_grf[k] ? fv[k] / 1000.0 : fv[k];
          if (code == 0 && (v < f_lo[k] || v > f_hi[k])) code = (unsigned)(k + 1);
        }
      for (int k = FAST_FALL; k < nf; ++k) {  // robots with more tests than the fast set
        const int sidx = md->fall_sidx[k];
        double v = sq[col_off(sidx) + r * col_str(sidx)];
        if (d.n_grf > 0 && md->fall_idx[k] >= first_grf) v = v / 1000.0;
        if (code == 0 && (v < md->fall_lo[k] || v > md->fall_hi[k])) code = (unsigned)(k + 1);
      }
      ab = (code != 0 && use_abs) ? 1u : 0u;
      if (rt == OLY_REWARD_NONE) {
        p.reward[gr] = 0.0f;
      } else {
        double x = sq[rew_off + r * rew_str];
        if (rew_grf) x = x / 1000.0;
        auto f = [&](double s) -> float {
          if (rt == OLY_REWARD_TARGET_VELOCITY) {
            const double dv = s - tvel;
            return (float)exp(-(dv * dv));
          }
          return (float)s;  // PosReward
        };
        // step-0 rows use the carried state, read BEFORE any last-step row could overwrite it
        // (prev_in may alias prev_out only when T == 1: then both are this very lane)
        if (gr < p.N) p.reward[gr] = f(prev0);
        if (gr + p.N < p.R)
          p.reward[gr + p.N] = f(x);  // reward(t+1) reads obs(t), utils/reward.py:73
        else
          p.prev_out[gr - (p.R - p.N)] = x;  // self._obs of the last step
      }
    }
    // flags: pack 4 rows per dword inside the wave (lane i < 16 gathers lanes 4i..4i+3)
    if (full) {
      unsigned a4 = 0, c4 = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int srcl = (lane & 15) * 4 + k;
        a4 |= (unsigned)__shfl((int)ab, srcl, 64) << (8 * k);
        c4 |= (unsigned)__shfl((int)code, srcl, 64) << (8 * k);
      }
      if (lane < 16) {
        const long q4 = (row0 + (r - lane)) / 4 + lane;
        reinterpret_cast<unsigned*>(p.absorbing)[q4] = a4;
        if (p.fall_code) reinterpret_cast<unsigned*>(p.fall_code)[q4] = c4;
      }
    } else if (r < rows) {
      p.absorbing[row0 + r] = (uint8_t)ab;
      if (p.fall_code) p.fall_code[row0 + r] = (uint8_t)code;
    }
  }

  // created-observation column c of local row r, in float64 (table-driven form)
  auto obs_val = [&](int r, int c) -> double {
    const double x = sq[t_off[c] + r * t_str[c]];
    if (d.n_grf > 0 && c >= first_grf) return x / 1000.0;  // mean_grf / 1000.0
    return x;
  };

  // ---- (5) observation tile
  if (full) {
    const int nvec = ROWS * d.n_obs / OW;
    if (fixed_cols) {
      const int c0 = (tid * OW) % d.n_obs;
      const int rstep = THREADS * OW / d.n_obs;
      constexpr int NIT = (ROWS * D::n_obs_ct / OW + THREADS - 1) / THREADS;
      double v[NIT > 0 ? NIT : 1][OW];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int r = tid * OW / d.n_obs + it * rstep;
#pragma unroll
        for (int k = 0; k < OW; ++k) v[it][k] = (OLY_K1_ABLATE & 4) ? (double)(r + k) : sq[my_off[k] + r * my_str[k]];
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * THREADS;
        if (i < nvec) {
#pragma unroll
          for (int k = 0; k < OW; ++k)
            if (d.n_grf > 0 && c0 + k >= first_grf) v[it][k] = v[it][k] / 1000.0;
          if (OBS64) {
            double2 o;
            o.x = v[it][0]; o.y = v[it][OW - 1];
            reinterpret_cast<double2*>(static_cast<double*>(p.obs) + row0 * d.n_obs)[i] = o;
          } else {
            float4 o;
            o.x = (float)v[it][0]; o.y = (float)v[it][1 % OW];
            o.z = (float)v[it][2 % OW]; o.w = (float)v[it][3 % OW];
            reinterpret_cast<float4*>(static_cast<float*>(p.obs) + row0 * d.n_obs)[i] = o;
          }
        }
      }
    } else if (OBS64) {
      double2* out = reinterpret_cast<double2*>(static_cast<double*>(p.obs) + row0 * d.n_obs);
      for (int i = tid; i < nvec; i += THREADS) {
        int e = 2 * i, r = e / d.n_obs, c = e - r * d.n_obs;
        double2 o;
        o.x = obs_val(r, c);
        if (++c == d.n_obs) { c = 0; ++r; }
        o.y = obs_val(r, c);
        out[i] = o;
      }
    } else {
      float4* out = reinterpret_cast<float4*>(static_cast<float*>(p.obs) + row0 * d.n_obs);
      for (int i = tid; i < nvec; i += THREADS) {
        int e = 4 * i, r = e / d.n_obs, c = e - r * d.n_obs;
        float4 o;
        o.x = (float)obs_val(r, c);
        if (++c == d.n_obs) { c = 0; ++r; }
        o.y = (float)obs_val(r, c);
        if (++c == d.n_obs) { c = 0; ++r; }
        o.z = (float)obs_val(r, c);
        if (++c == d.n_obs) { c = 0; ++r; }
        o.w = (float)obs_val(r, c);
        out[i] = o;
      }
    }
  } else {
    for (int e = tid; e < rows * d.n_obs; e += THREADS) {
      const int r = e / d.n_obs, c = e - r * d.n_obs;
      const double v = obs_val(r, c);
      if (OBS64)
        static_cast<double*>(p.obs)[row0 * d.n_obs + e] = v;
      else
        static_cast<float*>(p.obs)[row0 * d.n_obs + e] = (float)v;
    }
  }

  // ---- (6) control tile: un-normalise, clamp to ctrlrange, in actuator order
  if (p.ctrl) {
    auto ctrl_val = [&](int r, int j) -> double {
      if (OLY_K1_ABLATE & 8) return (double)(r - j);
      const int k = t_cidx[j];
      const double a = (double)sa[r * d.n_act + (k < 0 ? 0 : k)];
      double u = a * t_act[d.nu + j] + t_act[j];
      const double lo = t_act[2 * d.nu + j], hi = t_act[3 * d.nu + j];
      if (u < lo) u = lo;
      if (u > hi) u = hi;
      return k < 0 ? 0.0 : u;
    };
    if (full && !CTRL64) {
      float4* out = reinterpret_cast<float4*>(static_cast<float*>(p.ctrl) + row0 * d.nu);
      const int n4 = ROWS * d.nu / 4;
      for (int i = tid; i < n4; i += THREADS) {
        int e = 4 * i, r = e / d.nu, j = e - r * d.nu;
        float4 o;
        o.x = (float)ctrl_val(r, j);
        if (++j == d.nu) { j = 0; ++r; }
        o.y = (float)ctrl_val(r, j);
        if (++j == d.nu) { j = 0; ++r; }
        o.z = (float)ctrl_val(r, j);
        if (++j == d.nu) { j = 0; ++r; }
        o.w = (float)ctrl_val(r, j);
        out[i] = o;
      }
    } else {
      for (int e = tid; e < rows * d.nu; e += THREADS) {
        const int r = e / d.nu, j = e - r * d.nu;
        const double u = ctrl_val(r, j);
        if (CTRL64)
          static_cast<double*>(p.ctrl)[row0 * d.nu + e] = u;
        else
          static_cast<float*>(p.ctrl)[row0 * d.nu + e] = (float)u;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Fast path: compile-time robot dimensions, full 16-B-aligned tiles only, PERSISTENT
// workgroups.  Setup (tables, per-lane source offsets, thresholds) is paid once per
// workgroup; per tile the body is: land prefetched registers in LDS -> barrier -> issue the
// next tile's loads (in flight during the rest) -> per-row tests -> dense obs sweep -> dense
// ctrl sweep -> barrier.  No tail handling, no table lookups in the sweeps.
// ------------------------------------------------------------------------------------
template <int ROWS, class D, bool OBS64, bool CTRL64>
__global__ __launch_bounds__(THREADS) void il_tile_kernel(IlArgs p) {
  static_assert(D::is_static, "tile kernel needs compile-time dimensions");
  static_assert(ROWS % 64 == 0 && ROWS <= THREADS, "ROWS: whole waves, at most one row per lane");
  constexpr int NQ = D::nq, NV = D::nv, NG = D::n_grf, NA = D::n_act, NU = D::nu, NO = D::n_obs;
  constexpr int CQ = ROWS * NQ / 2, CV = ROWS * NV / 2, CG = ROWS * NG / 2, CA = ROWS * NA / 4;  // 16-B chunks
  constexpr int KQ = (CQ + THREADS - 1) / THREADS, KV = (CV + THREADS - 1) / THREADS,
                KG = (CG + THREADS - 1) / THREADS, KA = (CA + THREADS - 1) / THREADS;
  constexpr int ND = NQ + NV + NG;                  // staged doubles per row: q | v | mean_grf
  constexpr int OW = OBS64 ? 2 : 4;                 // obs values per lane store
  static_assert((ROWS * NO) % OW == 0, "observation tile must split into whole lane vectors");
  constexpr int NOV = ROWS * NO / OW;               // obs vectors per tile
  constexpr int NIT = (NOV + THREADS - 1) / THREADS; // obs stores per lane per tile
  constexpr int CW = CTRL64 ? 2 : 4;                // ctrl values per lane store
  constexpr int NCV = ROWS * NU / CW;               // ctrl vectors per tile
  constexpr int NCI = (NCV + THREADS - 1) / THREADS;

  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  double* sq = reinterpret_cast<double*>(lds);                                  // [ROWS][NQ]
  double* sv = sq + ROWS * NQ;                                                  // [ROWS][NV]
  double* sg = sv + ROWS * NV;                                                  // [ROWS][NG] (foot forces)
  float* sa = reinterpret_cast<float*>(lds + (size_t)ROWS * ND * 8);            // [ROWS][NA]
  double* t_act = reinterpret_cast<double*>(lds + (size_t)ROWS * ND * 8 + (size_t)ROWS * NA * 4);
  const IlDev* __restrict__ md = p.md;
  const int tid = threadIdx.x, lane = tid & 63;
  const bool with_ctrl = p.ctrl != nullptr;

  auto col_off = [&](int sidx) -> int {
    return sidx < NQ ? sidx : sidx < NQ + NV ? ROWS * NQ + (sidx - NQ) : ROWS * (NQ + NV) + (sidx - NQ - NV);
  };
  auto col_str = [&](int sidx) -> int { return sidx < NQ ? NQ : sidx < NQ + NV ? NV : NG; };
  auto is_grf = [&](int sidx) -> bool { return NG > 0 && sidx >= NQ + NV; };   // stored as mean_grf / 1000

  // ---- once per workgroup
  for (int j = tid; j < NU; j += THREADS) {   // per ACTUATOR: mean | delta | lo | hi
    const int k = md->ctrl_src[j];
    const int kk = k < 0 ? 0 : k;
    t_act[j] = md->act_mean[kk];
    t_act[NU + j] = md->act_delta[kk];
    t_act[2 * NU + j] = md->ctrl_lo[kk];
    t_act[3 * NU + j] = md->ctrl_hi[kk];
  }
  // The tile-local position of every value a lane stores is the same for every tile, so the
  // LDS element it reads (column permutation + row stride) is computed ONCE: e_off[it][k].
  int e_off[NIT][OW];
  unsigned e_grf = 0;                         // bit it*OW+k: that value is a foot-force column
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int k = 0; k < OW; ++k) {
      const int e = (tid + it * THREADS) * OW + k;
      int off = 0;
      if (e < ROWS * NO) {
        const int r = e / NO, c = e - r * NO;
        const int sidx = md->src[c];
        off = col_off(sidx) + r * col_str(sidx);
        if (is_grf(sidx)) e_grf |= 1u << (it * OW + k);
      }
      e_off[it][k] = off;
    }
  static_assert(NIT * OW <= 32, "grf bitmask holds 32 values per lane");
  int c_aidx[NCI][CW];                        // ctrl elements of this lane: LDS index of the
  int c_j[NCI][CW];                           // action value (-1: actuator not driven), actuator
#pragma unroll
  for (int it = 0; it < NCI; ++it)
#pragma unroll
    for (int q = 0; q < CW; ++q) {
      const int e = (tid + it * THREADS) * CW + q;
      const int r = e / NU, j = e - r * NU;
      int k = -1;
      if (e < ROWS * NU) k = md->ctrl_src[j];
      c_j[it][q] = j;
      c_aidx[it][q] = k < 0 ? -1 : r * NA + k;
    }
  // per-row tests: wave-uniform thresholds; unused slots can never trigger
  const int nf = md->n_fall;
  int f_off[FAST_FALL], f_str[FAST_FALL];
  double f_lo[FAST_FALL], f_hi[FAST_FALL];
  unsigned f_grf = 0;
#pragma unroll
  for (int k = 0; k < FAST_FALL; ++k) {
    const int kk = k < nf ? k : 0;
    const int sidx = md->fall_sidx[kk];
    f_off[k] = col_off(sidx);
    f_str[k] = col_str(sidx);
    if (is_grf(sidx)) f_grf |= 1u << k;
    f_lo[k] = k < nf ? md->fall_lo[kk] : -__builtin_huge_val();
    f_hi[k] = k < nf ? md->fall_hi[kk] : __builtin_huge_val();
  }
  const int rt = md->reward_type;
  const int rew_off = col_off(md->reward_sidx), rew_str = col_str(md->reward_sidx);
  const bool rew_grf = is_grf(md->reward_sidx);
  const double tvel = md->target_velocity;
  const int use_abs = md->use_absorbing;
  auto rew_f = [&](double s) -> float {
    if (rt == OLY_REWARD_TARGET_VELOCITY) {
      const double dv = s - tvel;
      return (float)exp(-(dv * dv));
    }
    return (float)s;  // PosReward
  };

  const long ntiles = p.tile0;  // number of full tiles
  u32x4 rq[KQ], rv[KV], rg[KG > 0 ? KG : 1], ra[KA];
  auto issue_loads = [&](long t) {
    const long r0 = t * ROWS;
    const u32x4* gq = reinterpret_cast<const u32x4*>(p.qpos + r0 * NQ) + tid;
    const u32x4* gv = reinterpret_cast<const u32x4*>(p.qvel + r0 * NV) + tid;
#pragma unroll
    for (int k = 0; k < KQ; ++k)
      if (k * THREADS + THREADS <= CQ || tid + k * THREADS < CQ) rq[k] = ld16(gq + k * THREADS);
#pragma unroll
    for (int k = 0; k < KV; ++k)
      if (k * THREADS + THREADS <= CV || tid + k * THREADS < CV) rv[k] = ld16(gv + k * THREADS);
    if (NG > 0) {
      const u32x4* gg = reinterpret_cast<const u32x4*>(p.grf + r0 * NG) + tid;
#pragma unroll
      for (int k = 0; k < KG; ++k)
        if (k * THREADS + THREADS <= CG || tid + k * THREADS < CG) rg[k] = ld16(gg + k * THREADS);
    }
    if (with_ctrl) {
      const u32x4* ga = reinterpret_cast<const u32x4*>(p.action + r0 * NA) + tid;
#pragma unroll
      for (int k = 0; k < KA; ++k)
        if (k * THREADS + THREADS <= CA || tid + k * THREADS < CA) ra[k] = ld16(ga + k * THREADS);
    }
  };

  long tile = blockIdx.x;
  if (tile < ntiles) issue_loads(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    const long row0 = tile * ROWS;
    // ---- land this tile (global order == LDS order)
    {
      u32x4* lq = reinterpret_cast<u32x4*>(sq) + tid;
      u32x4* lv = reinterpret_cast<u32x4*>(sv) + tid;
      u32x4* la = reinterpret_cast<u32x4*>(sa) + tid;
#pragma unroll
      for (int k = 0; k < KQ; ++k)
        if (k * THREADS + THREADS <= CQ || tid + k * THREADS < CQ) lq[k * THREADS] = rq[k];
#pragma unroll
      for (int k = 0; k < KV; ++k)
        if (k * THREADS + THREADS <= CV || tid + k * THREADS < CV) lv[k * THREADS] = rv[k];
      if (NG > 0) {
        u32x4* lg = reinterpret_cast<u32x4*>(sg) + tid;
#pragma unroll
        for (int k = 0; k < KG; ++k)
          if (k * THREADS + THREADS <= CG || tid + k * THREADS < CG) lg[k * THREADS] = rg[k];
      }
      if (with_ctrl) {
#pragma unroll
        for (int k = 0; k < KA; ++k)
          if (k * THREADS + THREADS <= CA || tid + k * THREADS < CA) la[k * THREADS] = ra[k];
      }
    }
    __syncthreads();
    // ---- the next tile's loads fly while this one is processed
    const long next = tile + gridDim.x;
    if (next < ntiles) issue_loads(next);

    // ---- per-row scalars (waves 0 .. ROWS/64-1): fall tests, reward of the NEXT step, flags
    if (tid < ROWS) {
      const int r = tid;
      const long gr = row0 + r;
      double fv[FAST_FALL];
#pragma unroll
      for (int k = 0; k < FAST_FALL; ++k) {
        fv[k] = sq[f_off[k] + r * f_str[k]];
        if (NG > 0 && ((f_grf >> k) & 1u)) fv[k] = fv[k] / 1000.0;
      }
      double x = sq[rew_off + r * rew_str];
      if (NG > 0 && rew_grf) x = x / 1000.0;
      unsigned code = 0;
#pragma unroll
      for (int k = FAST_FALL - 1; k >= 0; --k)
        if (fv[k] < f_lo[k] || fv[k] > f_hi[k]) code = (unsigned)(k + 1);  // lowest k wins
      const unsigned ab = (code != 0 && use_abs) ? 1u : 0u;
      if (rt == OLY_REWARD_NONE) {
        p.reward[gr] = 0.0f;
      } else {
        // step-0 rows read the carried state BEFORE the last-step row of the same env could
        // overwrite it (prev_in may alias prev_out only when T == 1: same lane, this order)
        if (gr < p.N) p.reward[gr] = rew_f(p.prev_in[gr]);
        if (gr + p.N < p.R)
          p.reward[gr + p.N] = rew_f(x);     // reward(t+1) reads obs(t), utils/reward.py:73
        else
          p.prev_out[gr - (p.R - p.N)] = x;  // self._obs of the last step
      }
      // flags: 4 rows per dword (lane i < 16 gathers lanes 4i .. 4i+3 of its wave)
      unsigned a4 = 0, c4 = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int srcl = (lane & 15) * 4 + k;
        a4 |= (unsigned)__shfl((int)ab, srcl, 64) << (8 * k);
        c4 |= (unsigned)__shfl((int)code, srcl, 64) << (8 * k);
      }
      if (lane < 16) {
        const long q4 = (row0 + (r - lane)) / 4 + lane;
        reinterpret_cast<unsigned*>(p.absorbing)[q4] = a4;
        if (p.fall_code) reinterpret_cast<unsigned*>(p.fall_code)[q4] = c4;
      }
    }

    // ---- observation tile: NIT dense vector stores per lane, one LDS round trip
    {
      double v[NIT][OW];
#pragma unroll
      for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int k = 0; k < OW; ++k) {
          v[it][k] = sq[e_off[it][k]];
          if (NG > 0 && ((e_grf >> (it * OW + k)) & 1u)) v[it][k] = v[it][k] / 1000.0;  // mean_grf / 1000.0
        }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = tid + it * THREADS;
        if (it * THREADS + THREADS <= NOV || i < NOV) {
          if (OBS64) {
            f64x2 o;
            o.x = v[it][0]; o.y = v[it][OW - 1];
            st16(reinterpret_cast<f64x2*>(static_cast<double*>(p.obs) + row0 * NO) + i, o);
          } else {
            f32x4 o;
            o.x = (float)v[it][0]; o.y = (float)v[it][1 % OW];
            o.z = (float)v[it][2 % OW]; o.w = (float)v[it][3 % OW];
            st16(reinterpret_cast<f32x4*>(static_cast<float*>(p.obs) + row0 * NO) + i, o);
          }
        }
      }
    }

    // ---- control tile: un-normalise in fp64, clamp to ctrlrange, actuator order
    if (with_ctrl) {
#pragma unroll
      for (int it = 0; it < NCI; ++it) {
        const int i = tid + it * THREADS;
        if (it * THREADS + THREADS <= NCV || i < NCV) {
          double u[CW];
#pragma unroll
          for (int q = 0; q < CW; ++q) {
            const int ai = c_aidx[it][q], j = c_j[it][q];
            const double a = (double)sa[ai < 0 ? 0 : ai];
            double x = a * t_act[NU + j] + t_act[j];
            const double lo = t_act[2 * NU + j], hi = t_act[3 * NU + j];
            if (x < lo) x = lo;
            if (x > hi) x = hi;
            u[q] = ai < 0 ? 0.0 : x;
          }
          if (CTRL64) {
            f64x2 o;
            o.x = u[0]; o.y = u[CW - 1];
            st16(reinterpret_cast<f64x2*>(static_cast<double*>(p.ctrl) + row0 * NU) + i, o);
          } else {
            f32x4 o;
            o.x = (float)u[0]; o.y = (float)u[1 % CW]; o.z = (float)u[2 % CW]; o.w = (float)u[3 % CW];
            st16(reinterpret_cast<f32x4*>(static_cast<float*>(p.ctrl) + row0 * NU) + i, o);
          }
        }
      }
    }
    __syncthreads();  // all LDS reads of this tile are done before the next one lands
  }
}

// K5 alone: un-normalise, clamp, actuator order, for N rows (elementwise over N*nu outputs).
template <bool CTRL64>
__global__ __launch_bounds__(THREADS) void il_ctrl_kernel(const IlDev* __restrict__ md, long total,
                                                          const float* __restrict__ action,
                                                          void* __restrict__ ctrl) {
  const int nu = md->nu, n_act = md->n_act;
  const long stride = (long)gridDim.x * THREADS;
  for (long e = (long)blockIdx.x * THREADS + threadIdx.x; e < total; e += stride) {
    const long r = e / nu;
    const int j = (int)(e - r * nu);
    const int k = md->ctrl_src[j];
    double u = 0.0;
    if (k >= 0) {
      u = (double)action[r * n_act + k] * md->act_delta[k] + md->act_mean[k];
      if (u < md->ctrl_lo[k]) u = md->ctrl_lo[k];
      if (u > md->ctrl_hi[k]) u = md->ctrl_hi[k];
    }
    if (CTRL64)
      static_cast<double*>(ctrl)[e] = u;
    else
      static_cast<float*>(ctrl)[e] = (float)u;
  }
}

// ------------------------------------------------------------------------------------
// Persistent tile kernel for RUNTIME robot dimensions (any table-driven robot, foot-force
// columns included): same life cycle as il_tile_kernel, but the per-element source offsets
// live in LDS tables built once per workgroup (eo: observation element -> staged element,
// co: control element -> staged action element | actuator) instead of registers.
// ------------------------------------------------------------------------------------
constexpr int DYN_ROWS = 64;

struct DynCarve {
  int a, eo, co, tact, total;
  __host__ __device__ DynCarve(int nq, int nv, int n_grf, int n_act, int nu, int n_obs) {
    auto al = [](int x) { return (x + 15) & ~15; };
    a = al(DYN_ROWS * (nq + nv + n_grf) * 8);
    eo = al(a + DYN_ROWS * n_act * 4);
    co = al(eo + DYN_ROWS * n_obs * 4);
    tact = al(co + DYN_ROWS * nu * 4);
    total = al(tact + 4 * nu * 8);
  }
};

template <bool OBS64, bool CTRL64>
__global__ __launch_bounds__(THREADS) void il_dyn_tile_kernel(IlArgs p) {
  constexpr int ROWS = DYN_ROWS;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const IlDev* __restrict__ md = p.md;
  const int nq = md->nq, nv = md->nv, n_grf = md->n_grf, n_act = md->n_act, nu = md->nu, n_obs = md->n_obs;
  const DynCarve cv(nq, nv, n_grf, n_act, nu, n_obs);
  const int tid = threadIdx.x, lane = tid & 63;
  double* sq = reinterpret_cast<double*>(lds);
  float* sa = reinterpret_cast<float*>(lds + cv.a);
  unsigned* eo = reinterpret_cast<unsigned*>(lds + cv.eo);   // bit 31: divide by 1000 (grf column)
  int* co = reinterpret_cast<int*>(lds + cv.co);             // (action element << 8) | actuator, -1: none
  double* t_act = reinterpret_cast<double*>(lds + cv.tact);  // per ACTUATOR: mean|delta|lo|hi
  const bool with_ctrl = p.ctrl != nullptr;
  const int first_grf = n_obs - n_grf;

  auto col_off = [&](int sidx) -> int {
    if (sidx < nq) return sidx;
    if (sidx < nq + nv) return ROWS * nq + (sidx - nq);
    return ROWS * (nq + nv) + (sidx - nq - nv);
  };
  auto col_str = [&](int sidx) -> int { return sidx < nq ? nq : (sidx < nq + nv ? nv : n_grf); };

  // ---- once per workgroup: element tables
  for (int e = tid; e < ROWS * n_obs; e += THREADS) {
    const int r = e / n_obs, c = e - r * n_obs;
    const int sidx = md->src[c];
    eo[e] = (unsigned)(col_off(sidx) + r * col_str(sidx)) | (c >= first_grf && n_grf > 0 ? 0x80000000u : 0u);
  }
  for (int e = tid; e < ROWS * nu; e += THREADS) {
    const int r = e / nu, j = e - r * nu;
    const int k = md->ctrl_src[j];
    co[e] = k < 0 ? -1 : (((r * n_act + k) << 8) | j);
  }
  for (int j = tid; j < nu; j += THREADS) {
    const int k = md->ctrl_src[j];
    const int kk = k < 0 ? 0 : k;
    t_act[j] = md->act_mean[kk];
    t_act[nu + j] = md->act_delta[kk];
    t_act[2 * nu + j] = md->ctrl_lo[kk];
    t_act[3 * nu + j] = md->ctrl_hi[kk];
  }
  const int nf = md->n_fall;
  const int rt = md->reward_type;
  const int rew_sidx = md->reward_sidx;
  const bool rew_grf = n_grf > 0 && md->reward_idx >= first_grf;
  const double tvel = md->target_velocity;
  const int use_abs = md->use_absorbing;
  auto rew_f = [&](double s) -> float {
    if (rt == OLY_REWARD_TARGET_VELOCITY) {
      const double dv = s - tvel;
      return (float)exp(-(dv * dv));
    }
    return (float)s;
  };

  // 16-B chunks per array and tile; each array is prefetched by its own short register run
  // (base pointer + lane + k * THREADS: no per-chunk pointer arithmetic)
  const int cq = ROWS * nq / 2, cvv = ROWS * nv / 2, cg = ROWS * n_grf / 2;
  const int ca = with_ctrl ? ROWS * n_act / 4 : 0;
  constexpr int KQ = 5, KG = 1, KA = 3;   // covers nq,nv <= 40, n_grf <= 8, n_act <= 48 at 64 rows
  u32x4 rq[KQ], rv[KQ], rg[KG], ra[KA];
  u32x4* lq = reinterpret_cast<u32x4*>(sq);
  u32x4* lv = lq + cq;
  u32x4* lg = lv + cvv;
  u32x4* la = reinterpret_cast<u32x4*>(sa);
  auto issue_loads = [&](long t) {
    const long r0 = t * ROWS;
    const u32x4* gq = reinterpret_cast<const u32x4*>(p.qpos + r0 * nq) + tid;
    const u32x4* gv = reinterpret_cast<const u32x4*>(p.qvel + r0 * nv) + tid;
#pragma unroll
    for (int k = 0; k < KQ; ++k)
      if (tid + k * THREADS < cq) rq[k] = ld16(gq + k * THREADS);
#pragma unroll
    for (int k = 0; k < KQ; ++k)
      if (tid + k * THREADS < cvv) rv[k] = ld16(gv + k * THREADS);
    if (cg > 0) {
      const u32x4* gg = reinterpret_cast<const u32x4*>(p.grf + r0 * n_grf) + tid;
#pragma unroll
      for (int k = 0; k < KG; ++k)
        if (tid + k * THREADS < cg) rg[k] = ld16(gg + k * THREADS);
    }
    if (ca > 0) {
      const u32x4* ga = reinterpret_cast<const u32x4*>(p.action + r0 * n_act) + tid;
#pragma unroll
      for (int k = 0; k < KA; ++k)
        if (tid + k * THREADS < ca) ra[k] = ld16(ga + k * THREADS);
    }
  };
  auto land = [&](long row0) {
#pragma unroll
    for (int k = 0; k < KQ; ++k)
      if (tid + k * THREADS < cq) lq[tid + k * THREADS] = rq[k];
#pragma unroll
    for (int k = 0; k < KQ; ++k)
      if (tid + k * THREADS < cvv) lv[tid + k * THREADS] = rv[k];
#pragma unroll
    for (int k = 0; k < KG; ++k)
      if (tid + k * THREADS < cg) lg[tid + k * THREADS] = rg[k];
#pragma unroll
    for (int k = 0; k < KA; ++k)
      if (tid + k * THREADS < ca) la[tid + k * THREADS] = ra[k];
    // chunks beyond the register runs (very wide robots): plain staged copy
    for (int g = tid + KQ * THREADS; g < cq; g += THREADS)
      lq[g] = ld16(reinterpret_cast<const u32x4*>(p.qpos + row0 * nq) + g);
    for (int g = tid + KQ * THREADS; g < cvv; g += THREADS)
      lv[g] = ld16(reinterpret_cast<const u32x4*>(p.qvel + row0 * nv) + g);
    for (int g = tid + KG * THREADS; g < cg; g += THREADS)
      lg[g] = ld16(reinterpret_cast<const u32x4*>(p.grf + row0 * n_grf) + g);
    for (int g = tid + KA * THREADS; g < ca; g += THREADS)
      la[g] = ld16(reinterpret_cast<const u32x4*>(p.action + row0 * n_act) + g);
  };

  const long ntiles = p.tile0;
  long tile = blockIdx.x;
  if (tile < ntiles) issue_loads(tile);
  __syncthreads();  // tables
  for (; tile < ntiles; tile += gridDim.x) {
    const long row0 = tile * ROWS;
    land(row0);
    __syncthreads();
    const long next = tile + gridDim.x;
    if (next < ntiles) issue_loads(next);

    // ---- per-row scalars (wave 0: ROWS == 64)
    if (tid < ROWS) {
      const int r = tid;
      const long gr = row0 + r;
      unsigned code = 0;
      for (int k = 0; k < nf; ++k) {
        const int sidx = md->fall_sidx[k];
        double v = sq[col_off(sidx) + r * col_str(sidx)];
        if (n_grf > 0 && md->fall_idx[k] >= first_grf) v = v / 1000.0;
        if (code == 0 && (v < md->fall_lo[k] || v > md->fall_hi[k])) code = (unsigned)(k + 1);
      }
      const unsigned ab = (code != 0 && use_abs) ? 1u : 0u;
      if (rt == OLY_REWARD_NONE) {
        p.reward[gr] = 0.0f;
      } else {
        double x = sq[col_off(rew_sidx) + r * col_str(rew_sidx)];
        if (rew_grf) x = x / 1000.0;
        if (gr < p.N) p.reward[gr] = rew_f(p.prev_in[gr]);
        if (gr + p.N < p.R)
          p.reward[gr + p.N] = rew_f(x);
        else
          p.prev_out[gr - (p.R - p.N)] = x;
      }
      unsigned a4 = 0, c4 = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int srcl = (lane & 15) * 4 + k;
        a4 |= (unsigned)__shfl((int)ab, srcl, 64) << (8 * k);
        c4 |= (unsigned)__shfl((int)code, srcl, 64) << (8 * k);
      }
      if (lane < 16) {
        const long q4 = row0 / 4 + lane;
        reinterpret_cast<unsigned*>(p.absorbing)[q4] = a4;
        if (p.fall_code) reinterpret_cast<unsigned*>(p.fall_code)[q4] = c4;
      }
    }

    // ---- observation tile
    {
      constexpr int OW = OBS64 ? 2 : 4;
      const int nvec = ROWS * n_obs / OW;
#pragma unroll 2
      for (int i = tid; i < nvec; i += THREADS) {
        unsigned o4[OW];
        double v[OW];
#pragma unroll
        for (int k = 0; k < OW; ++k) o4[k] = eo[i * OW + k];
#pragma unroll
        for (int k = 0; k < OW; ++k) {
          v[k] = sq[o4[k] & 0x7fffffffu];
          if (o4[k] & 0x80000000u) v[k] = v[k] / 1000.0;
        }
        if (OBS64) {
          f64x2 o;
          o.x = v[0]; o.y = v[OW - 1];
          st16(reinterpret_cast<f64x2*>(static_cast<double*>(p.obs) + row0 * n_obs) + i, o);
        } else {
          f32x4 o;
          o.x = (float)v[0]; o.y = (float)v[1 % OW]; o.z = (float)v[2 % OW]; o.w = (float)v[3 % OW];
          st16(reinterpret_cast<f32x4*>(static_cast<float*>(p.obs) + row0 * n_obs) + i, o);
        }
      }
    }

    // ---- control tile
    if (with_ctrl) {
      constexpr int CW = CTRL64 ? 2 : 4;
      const int ncv = ROWS * nu / CW;
      for (int i = tid; i < ncv; i += THREADS) {
        double u[CW];
#pragma unroll
        for (int q = 0; q < CW; ++q) {
          const int c = co[i * CW + q];
          const int j = c & 0xff, ai = c >> 8;
          const double a = (double)sa[c < 0 ? 0 : ai];
          double x = a * t_act[nu + (c < 0 ? 0 : j)] + t_act[c < 0 ? 0 : j];
          const double lo = t_act[2 * nu + (c < 0 ? 0 : j)], hi = t_act[3 * nu + (c < 0 ? 0 : j)];
          if (x < lo) x = lo;
          if (x > hi) x = hi;
          u[q] = c < 0 ? 0.0 : x;
        }
        if (CTRL64) {
          f64x2 o;
          o.x = u[0]; o.y = u[CW - 1];
          st16(reinterpret_cast<f64x2*>(static_cast<double*>(p.ctrl) + row0 * nu) + i, o);
        } else {
          f32x4 o;
          o.x = (float)u[0]; o.y = (float)u[1 % CW]; o.z = (float)u[2 % CW]; o.w = (float)u[3 % CW];
          st16(reinterpret_cast<f32x4*>(static_cast<float*>(p.ctrl) + row0 * nu) + i, o);
        }
      }
    }
    __syncthreads();
  }
}

using H1Dims = StaticDims<17, 17, 0, 11, 11, 32>;      // UnitreeH1, arms removed
using AtlasDims = StaticDims<16, 16, 0, 10, 10, 30>;   // Atlas, arms + back removed (default)
using TalosDims = StaticDims<18, 18, 0, 12, 12, 34>;   // Talos, arms removed (default)
using H1FFDims = StaticDims<17, 17, 6, 11, 11, 38>;    // UnitreeH1, use_foot_forces=True
using TalosFFDims = StaticDims<18, 18, 6, 12, 12, 40>; // Talos, use_foot_forces=True
using H1ArmsDims = StaticDims<25, 25, 0, 19, 19, 48>;  // UnitreeH1, disable_arms=False
// the remaining (disable_arms, disable_back_joint) combinations of the three robots
using H1ArmsNoBackDims = StaticDims<24, 24, 0, 18, 18, 46>;
using AtlasBackDims = StaticDims<19, 19, 0, 13, 13, 36>;
using ArmsNoBack28Dims = StaticDims<28, 28, 0, 22, 22, 54>;   // Atlas / Talos with arms, back removed
using AtlasFullDims = StaticDims<31, 31, 0, 25, 25, 60>;
using TalosFullDims = StaticDims<30, 30, 0, 24, 24, 58>;

template <int ROWS, class D>
int launch_generic(oly_ctx* ctx, IlArgs a, long tile0, int out_flags, hipStream_t s) {
  const IlDev& h = ctx->il_host;
  Carve<ROWS, D> cv(h.nq, h.nv, h.n_grf, h.n_act, h.nu, h.n_obs);
  const long tiles = (a.R + ROWS - 1) / ROWS - tile0;
  if (tiles <= 0) return OLY_OK;
  if (tiles > 0x7fffffffL) OLY_FAIL(ctx, OLY_EINVAL, "oly_il_step: too many rows");
  a.tile0 = tile0;
  dim3 grid((unsigned)tiles), block(THREADS);
  const bool o64 = out_flags & OLY_OUT_OBS_F64, c64 = out_flags & OLY_OUT_CTRL_F64;
#define OLY_K1(O, C_)                                                                       \
  do {                                                                                      \
    auto k = il_step_kernel<ROWS, D, O, C_>;                                                \
    if (cv.total > 48 * 1024)                                                               \
      OLY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, cv.total)); \
    hipLaunchKernelGGL(k, grid, block, cv.total, s, a);                                     \
  } while (0)
  if (o64 && c64) OLY_K1(true, true);
  else if (o64) OLY_K1(true, false);
  else if (c64) OLY_K1(false, true);
  else OLY_K1(false, false);
#undef OLY_K1
  OLY_LAUNCH_CHECK(ctx, "il_step_kernel");
  return OLY_OK;
}

// Full aligned tiles through the persistent tile kernel, the ragged remainder (< ROWS rows)
// through the generic kernel.
template <int ROWS, class D>
int launch_fast(oly_ctx* ctx, IlArgs a, int out_flags, int wg_per_cu, hipStream_t s) {
  const long nfull = a.R / ROWS;
  if (nfull > 0) {
    const int lds = ROWS * (D::nq + D::nv + D::n_grf) * 8 + ROWS * D::n_act * 4 + 4 * D::nu * 8;
    int per_cu = (160 * 1024) / lds;
    if (per_cu > 8) per_cu = 8;
    if (wg_per_cu > 0) per_cu = wg_per_cu;
    long want = (long)ctx->num_cu * per_cu;
    if (want > nfull) want = nfull;
    IlArgs b = a;
    b.tile0 = nfull;
    dim3 grid((unsigned)want), block(THREADS);
    const bool o64 = out_flags & OLY_OUT_OBS_F64, c64 = out_flags & OLY_OUT_CTRL_F64;
#define OLY_K1T(O, C_)                                                                      \
  do {                                                                                      \
    auto k = il_tile_kernel<ROWS, D, O, C_>;                                                \
    if (lds > 48 * 1024)                                                                    \
      OLY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));   \
    hipLaunchKernelGGL(k, grid, block, lds, s, b);                                          \
  } while (0)
    if (o64 && c64) OLY_K1T(true, true);
    else if (o64) OLY_K1T(true, false);
    else if (c64) OLY_K1T(false, true);
    else OLY_K1T(false, false);
#undef OLY_K1T
    OLY_LAUNCH_CHECK(ctx, "il_tile_kernel");
  }
  if (nfull * ROWS < a.R) return launch_generic<ROWS, D>(ctx, a, nfull, out_flags, s);
  return OLY_OK;
}

// Runtime-shaped robots: full aligned tiles through the persistent dyn tile kernel, the
// remainder through the generic per-tile kernel.
int launch_dyn(oly_ctx* ctx, IlArgs a, int out_flags, hipStream_t s) {
  const IlDev& h = ctx->il_host;
  const DynCarve cv(h.nq, h.nv, h.n_grf, h.n_act, h.nu, h.n_obs);
  const long nfull = (a.fast && cv.total <= 150 * 1024) ? a.R / DYN_ROWS : 0;
  if (nfull > 0) {
    int per_cu = (160 * 1024) / cv.total;
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    long want = (long)ctx->num_cu * per_cu;
    if (want > nfull) want = nfull;
    IlArgs b = a;
    b.tile0 = nfull;
    dim3 grid((unsigned)want), block(THREADS);
    const bool o64 = out_flags & OLY_OUT_OBS_F64, c64 = out_flags & OLY_OUT_CTRL_F64;
#define OLY_K1D(O, C_)                                                                      \
  do {                                                                                      \
    auto k = il_dyn_tile_kernel<O, C_>;                                                     \
    if (cv.total > 48 * 1024)                                                               \
      OLY_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, cv.total)); \
    hipLaunchKernelGGL(k, grid, block, cv.total, s, b);                                     \
  } while (0)
    if (o64 && c64) OLY_K1D(true, true);
    else if (o64) OLY_K1D(true, false);
    else if (c64) OLY_K1D(false, true);
    else OLY_K1D(false, false);
#undef OLY_K1D
    OLY_LAUNCH_CHECK(ctx, "il_dyn_tile_kernel");
  }
  if (nfull * DYN_ROWS < a.R) return launch_generic<DYN_ROWS, DynDims>(ctx, a, nfull, out_flags, s);
  return OLY_OK;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int oly_il_configure(oly_ctx* ctx, const oly_il_model* m) {
  if (!ctx || !m) return OLY_EINVAL;
  const int n_spec = m->n_pos + m->n_vel;
  const int n_obs = n_spec - m->n_drop + m->n_grf;
  if (m->nq <= 0 || m->nv <= 0 || m->n_pos < 0 || m->n_vel < 0 || m->n_drop < 0 || m->n_grf < 0 ||
      m->n_drop > n_spec || n_obs <= 0 || n_obs > OLY_MAX_OBS || m->n_act < 0 ||
      m->n_act > OLY_MAX_ACT || m->nu < m->n_act || m->nu > OLY_MAX_ACT || m->n_fall < 0 ||
      m->n_fall > OLY_MAX_FALL)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_il_configure: bad shape (n_obs=%d n_act=%d nu=%d n_fall=%d)",
             n_obs, m->n_act, m->nu, m->n_fall);
  if (!m->qpos_adr || !m->qvel_adr || (m->n_act && (!m->act_to_ctrl || !m->act_mean ||
      !m->act_delta || !m->ctrl_lo || !m->ctrl_hi)) || (m->n_fall && (!m->fall_idx || !m->fall_lo ||
      !m->fall_hi)))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_il_configure: NULL table");
  ctx->il_ok = false;  // a failed re-configure must not leave a half-written model usable
  IlDev& h = ctx->il_host;
  memset(&h, 0, sizeof(h));
  h.nq = m->nq; h.nv = m->nv; h.n_pos = m->n_pos; h.n_vel = m->n_vel; h.n_drop = m->n_drop;
  h.n_grf = m->n_grf; h.n_act = m->n_act; h.nu = m->nu; h.n_obs = n_obs; h.n_fall = m->n_fall;
  h.reward_type = m->reward_type; h.reward_idx = m->reward_idx;
  h.use_absorbing = m->use_absorbing_states; h.target_velocity = m->target_velocity;
  if (h.reward_type < OLY_REWARD_NONE || h.reward_type > OLY_REWARD_X_POS)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_il_configure: unknown reward_type %d", h.reward_type);
  if (h.reward_type != OLY_REWARD_NONE && (h.reward_idx < 0 || h.reward_idx >= n_obs))
    OLY_FAIL(ctx, OLY_ERANGE, "oly_il_configure: reward_idx %d outside [0,%d)", h.reward_idx, n_obs);
  for (int c = 0; c < n_obs; ++c) {
    const int slot = c + m->n_drop;
    int s;
    if (slot < m->n_pos) {
      s = m->qpos_adr[slot];
      if (s < 0 || s >= m->nq) OLY_FAIL(ctx, OLY_ERANGE, "qpos_adr[%d]=%d outside nq=%d", slot, s, m->nq);
    } else if (slot < n_spec) {
      s = m->qvel_adr[slot - m->n_pos];
      if (s < 0 || s >= m->nv) OLY_FAIL(ctx, OLY_ERANGE, "qvel_adr[%d]=%d outside nv=%d", slot - m->n_pos, s, m->nv);
      s += m->nq;
    } else {
      s = m->nq + m->nv + (slot - n_spec);
    }
    h.src[c] = (short)s;
  }
  for (int k = 0; k < m->n_fall; ++k) {
    if (m->fall_idx[k] < 0 || m->fall_idx[k] >= n_obs)
      OLY_FAIL(ctx, OLY_ERANGE, "fall_idx[%d]=%d outside [0,%d)", k, m->fall_idx[k], n_obs);
    h.fall_idx[k] = m->fall_idx[k]; h.fall_lo[k] = m->fall_lo[k]; h.fall_hi[k] = m->fall_hi[k];
    h.fall_sidx[k] = h.src[m->fall_idx[k]];
  }
  h.reward_sidx = (h.reward_type != OLY_REWARD_NONE) ? h.src[h.reward_idx] : 0;
  for (int j = 0; j < m->nu; ++j) h.ctrl_src[j] = -1;
  for (int k = 0; k < m->n_act; ++k) {
    const int j = m->act_to_ctrl[k];
    if (j < 0 || j >= m->nu) OLY_FAIL(ctx, OLY_ERANGE, "act_to_ctrl[%d]=%d outside nu=%d", k, j, m->nu);
    h.ctrl_src[j] = (short)k;
    h.act_mean[k] = m->act_mean[k]; h.act_delta[k] = m->act_delta[k];
    h.ctrl_lo[k] = m->ctrl_lo[k]; h.ctrl_hi[k] = m->ctrl_hi[k];
  }
  OLY_HIP(ctx, hipSetDevice(ctx->device));
  OLY_HIP(ctx, hipMemcpy(ctx->il_dev, &h, sizeof(h), hipMemcpyHostToDevice));
  ctx->il_ok = true;
  return OLY_OK;
}

extern "C" int oly_il_obs_dim(const oly_ctx* ctx) {
  if (!ctx || !ctx->il_ok) return OLY_ENOTCONF;
  return ctx->il_host.n_obs;
}

extern "C" int oly_il_step(oly_ctx* ctx, int T, int N, const double* qpos, const double* qvel,
                           const float* action, const double* grf_mean, const double* prev_in,
                           double* prev_out, void* obs, float* reward, uint8_t* absorbing,
                           uint8_t* fall_code, void* ctrl, int out_flags, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->il_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_il_step before oly_il_configure");
  const IlDev& h = ctx->il_host;
  if (T < 0 || N < 0) OLY_FAIL(ctx, OLY_EINVAL, "oly_il_step: negative T or N");
  if (T == 0 || N == 0) return OLY_OK;
  if (!qpos || !qvel || !obs || !reward || !absorbing)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_il_step: NULL required pointer");
  if (h.reward_type != OLY_REWARD_NONE && (!prev_in || !prev_out))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_il_step: prev_in/prev_out required by the reward");
  if (T > 1 && prev_in == prev_out && prev_in)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_il_step: prev_in must not alias prev_out when T > 1");
  if (h.n_grf > 0 && !grf_mean) OLY_FAIL(ctx, OLY_EINVAL, "oly_il_step: grf_mean required (n_grf=%d)", h.n_grf);
  if (ctrl && !action) OLY_FAIL(ctx, OLY_EINVAL, "oly_il_step: ctrl requested without action");
  IlArgs a;
  a.md = ctx->il_dev; a.R = (long)T * N; a.N = N; a.qpos = qpos; a.qvel = qvel; a.action = action;
  a.grf = h.n_grf ? grf_mean : nullptr; a.prev_in = prev_in; a.prev_out = prev_out; a.obs = obs;
  a.reward = reward; a.absorbing = absorbing; a.fall_code = fall_code; a.ctrl = ctrl;
  a.fast = al16(qpos) && al16(qvel) && al16(action) && al16(grf_mean) && al16(obs) && al16(ctrl) &&
           ((reinterpret_cast<uintptr_t>(absorbing) & 3) == 0) &&
           ((reinterpret_cast<uintptr_t>(fall_code) & 3) == 0);
  a.tile0 = 0;
  auto shape = [&](int nq, int na, int no, int ng = 0) {
    return h.nq == nq && h.nv == nq && h.n_grf == ng && h.n_act == na && h.nu == na && h.n_obs == no &&
           h.n_pos == nq && h.n_vel == nq && h.n_drop == 2;
  };
  const bool fast_ok = a.fast && h.n_fall <= FAST_FALL;
  static const int wg_env = [] { const char* e = getenv("OLY_K1_WG_PER_CU"); return e ? atoi(e) : 0; }();
  static const int rows_env = [] { const char* e = getenv("OLY_K1_ROWS"); return e ? atoi(e) : 0; }();
#define OLY_K1_FAST(DIMS, DEFROWS)                                                              \
  do {                                                                                          \
    const int rows = rows_env ? rows_env : DEFROWS;                                             \
    if (rows == 64) return launch_fast<64, DIMS>(ctx, a, out_flags, wg_env, oly_s(stream));     \
    return launch_fast<128, DIMS>(ctx, a, out_flags, wg_env, oly_s(stream));                    \
  } while (0)
  if (fast_ok && shape(17, 11, 32)) OLY_K1_FAST(H1Dims, 128);
  if (fast_ok && shape(16, 10, 30)) OLY_K1_FAST(AtlasDims, 128);
  if (fast_ok && shape(18, 12, 34)) OLY_K1_FAST(TalosDims, 128);
  if (fast_ok && shape(17, 11, 38, 6)) OLY_K1_FAST(H1FFDims, 128);
  if (fast_ok && shape(18, 12, 40, 6)) OLY_K1_FAST(TalosFFDims, 128);
  if (fast_ok && shape(25, 19, 48)) OLY_K1_FAST(H1ArmsDims, 128);
  if (fast_ok && shape(24, 18, 46)) return launch_fast<128, H1ArmsNoBackDims>(ctx, a, out_flags, wg_env, oly_s(stream));
  if (fast_ok && shape(19, 13, 36)) return launch_fast<128, AtlasBackDims>(ctx, a, out_flags, wg_env, oly_s(stream));
  if (fast_ok && shape(28, 22, 54)) return launch_fast<128, ArmsNoBack28Dims>(ctx, a, out_flags, wg_env, oly_s(stream));
  if (fast_ok && shape(31, 25, 60)) return launch_fast<128, AtlasFullDims>(ctx, a, out_flags, wg_env, oly_s(stream));
  if (fast_ok && shape(30, 24, 58)) return launch_fast<128, TalosFullDims>(ctx, a, out_flags, wg_env, oly_s(stream));
#undef OLY_K1_FAST
  static const int dyn_env = [] { const char* e = getenv("OLY_K1_DYN_TILE"); return e ? atoi(e) : 1; }();
  if (dyn_env) return launch_dyn(ctx, a, out_flags, oly_s(stream));
  return launch_generic<64, DynDims>(ctx, a, 0, out_flags, oly_s(stream));
}

extern "C" int oly_il_ctrl(oly_ctx* ctx, int N, const float* action, void* ctrl, int out_flags,
                           oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->il_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_il_ctrl before oly_il_configure");
  if (N < 0 || (N > 0 && (!action || !ctrl))) OLY_FAIL(ctx, OLY_EINVAL, "oly_il_ctrl: bad argument");
  if (N == 0) return OLY_OK;
  const long total = (long)N * ctx->il_host.nu;
  long nb = (total + THREADS - 1) / THREADS;
  if (nb > 4096) nb = 4096;
  if (out_flags & OLY_OUT_CTRL_F64)
    hipLaunchKernelGGL(il_ctrl_kernel<true>, dim3((unsigned)nb), dim3(THREADS), 0, oly_s(stream), ctx->il_dev,
                       total, action, ctrl);
  else
    hipLaunchKernelGGL(il_ctrl_kernel<false>, dim3((unsigned)nb), dim3(THREADS), 0, oly_s(stream), ctx->il_dev,
                       total, action, ctrl);
  OLY_LAUNCH_CHECK(ctx, "il_ctrl_kernel");
  return OLY_OK;
}
