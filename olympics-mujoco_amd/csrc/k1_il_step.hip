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
#include "oly_common.h"

namespace {

constexpr int THREADS = 256;

template <int NQ, int NV, int NGRF, int NACT, int NU, int NOBS>
struct StaticDims {
  static constexpr int nq = NQ, nv = NV, n_grf = NGRF, n_act = NACT, nu = NU, n_obs = NOBS;
  __device__ explicit StaticDims(const IlDev*) {}
};
struct DynDims {
  int nq, nv, n_grf, n_act, nu, n_obs;
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
};

// LDS carve (all offsets in bytes, every region 16-B aligned):
//   [ staged doubles: q | v | g ][ staged action floats ][ tables ][ row codes ]
template <int ROWS, class D>
struct Carve {
  int q, v, g, a, tab_src, tab_csrc, tab_act, codes, total;
  __host__ __device__ Carve(int nq, int nv, int n_grf, int n_act, int nu, int n_obs) {
    auto al = [](int x) { return (x + 15) & ~15; };
    q = 0;
    v = q + ROWS * nq * 8;
    g = v + ROWS * nv * 8;
    a = al(g + ROWS * n_grf * 8);
    tab_src = al(a + ROWS * n_act * 4);
    tab_csrc = al(tab_src + n_obs * 2);
    tab_act = al(tab_csrc + nu * 2);
    codes = al(tab_act + 4 * n_act * 8);
    total = al(codes + 2 * ROWS);
  }
};

template <int ROWS, class D, bool OBS64, bool CTRL64>
__global__ __launch_bounds__(THREADS) void il_step_kernel(IlArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const IlDev* __restrict__ md = p.md;
  const D d(md);
  const Carve<ROWS, D> cv(d.nq, d.nv, d.n_grf, d.n_act, d.nu, d.n_obs);
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * ROWS;
  const int rows = (p.R - row0 < ROWS) ? (int)(p.R - row0) : ROWS;
  const bool full = (rows == ROWS) && p.fast;

  double* sq = reinterpret_cast<double*>(lds + cv.q);
  double* sv = reinterpret_cast<double*>(lds + cv.v);
  double* sg = reinterpret_cast<double*>(lds + cv.g);
  float* sa = reinterpret_cast<float*>(lds + cv.a);
  short* t_src = reinterpret_cast<short*>(lds + cv.tab_src);
  short* t_csrc = reinterpret_cast<short*>(lds + cv.tab_csrc);
  double* t_act = reinterpret_cast<double*>(lds + cv.tab_act);  // mean|delta|lo|hi
  unsigned char* s_abs = lds + cv.codes;
  unsigned char* s_code = s_abs + ROWS;

  // ---- stage inputs (global order == LDS order)
  if (full) {
    const int cq = ROWS * d.nq / 2, cvv = ROWS * d.nv / 2, cg = ROWS * d.n_grf / 2;
    const int ca = p.action ? ROWS * d.n_act / 4 : 0;
    const uint4* gq = reinterpret_cast<const uint4*>(p.qpos + row0 * d.nq);
    const uint4* gv = reinterpret_cast<const uint4*>(p.qvel + row0 * d.nv);
    const uint4* gg = reinterpret_cast<const uint4*>(p.grf ? p.grf + row0 * d.n_grf : nullptr);
    const uint4* ga = reinterpret_cast<const uint4*>(p.action ? p.action + row0 * d.n_act : nullptr);
    uint4* lq = reinterpret_cast<uint4*>(sq);
    uint4* lv = reinterpret_cast<uint4*>(sv);
    uint4* lg = reinterpret_cast<uint4*>(sg);
    uint4* la = reinterpret_cast<uint4*>(sa);
#pragma unroll 4
    for (int i = tid; i < cq; i += THREADS) lq[i] = gq[i];
#pragma unroll 4
    for (int i = tid; i < cvv; i += THREADS) lv[i] = gv[i];
    if (p.grf)
      for (int i = tid; i < cg; i += THREADS) lg[i] = gg[i];
#pragma unroll 2
    for (int i = tid; i < ca; i += THREADS) la[i] = ga[i];
  } else {
    for (int i = tid; i < rows * d.nq; i += THREADS) sq[i] = p.qpos[row0 * d.nq + i];
    for (int i = tid; i < rows * d.nv; i += THREADS) sv[i] = p.qvel[row0 * d.nv + i];
    if (p.grf)
      for (int i = tid; i < rows * d.n_grf; i += THREADS) sg[i] = p.grf[row0 * d.n_grf + i];
    if (p.action)
      for (int i = tid; i < rows * d.n_act; i += THREADS) sa[i] = p.action[row0 * d.n_act + i];
  }
  // ---- per-column tables
  for (int i = tid; i < d.n_obs; i += THREADS) t_src[i] = md->src[i];
  for (int i = tid; i < d.nu; i += THREADS) t_csrc[i] = md->ctrl_src[i];
  for (int i = tid; i < d.n_act; i += THREADS) {
    t_act[i] = md->act_mean[i];
    t_act[d.n_act + i] = md->act_delta[i];
    t_act[2 * d.n_act + i] = md->ctrl_lo[i];
    t_act[3 * d.n_act + i] = md->ctrl_hi[i];
  }
  __syncthreads();

  // created-observation column c of local row r, in float64
  auto obs_val = [&](int r, int c) -> double {
    const int s = t_src[c];
    if (s < d.nq) return sq[r * d.nq + s];
    if (s < d.nq + d.nv) return sv[r * d.nv + (s - d.nq)];
    return sg[r * d.n_grf + (s - d.nq - d.nv)] / 1000.0;
  };

  // ---- per-row scalars: fall tests, reward of the NEXT step, carried state
  for (int r = tid; r < ROWS; r += THREADS) {
    unsigned char code = 0, ab = 0;
    if (r < rows) {
      const int nf = md->n_fall;
      for (int k = 0; k < nf; ++k) {
        const double v = obs_val(r, md->fall_idx[k]);
        if (code == 0 && (v < md->fall_lo[k] || v > md->fall_hi[k])) code = (unsigned char)(k + 1);
      }
      ab = (code != 0 && md->use_absorbing) ? 1 : 0;
      const long gr = row0 + r;
      const int rt = md->reward_type;
      if (rt == OLY_REWARD_NONE) {
        p.reward[gr] = 0.0f;
      } else {
        const double x = obs_val(r, md->reward_idx);
        auto f = [&](double s) -> float {
          if (rt == OLY_REWARD_TARGET_VELOCITY) {
            const double dv = s - md->target_velocity;
            return (float)exp(-(dv * dv));
          }
          return (float)s;  // PosReward
        };
        // step-0 rows read the carried state BEFORE last-step rows may overwrite it
        // (prev_in may alias prev_out when T == 1: then both are this very lane)
        if (gr < p.N) p.reward[gr] = f(p.prev_in[gr]);
        if (gr + p.N < p.R)
          p.reward[gr + p.N] = f(x);  // reward(t+1) reads obs(t), utils/reward.py:73
        else
          p.prev_out[gr - (p.R - p.N)] = x;  // self._obs of the last step
      }
    }
    s_abs[r] = ab;
    s_code[r] = code;
  }

  // ---- observation tile
  if (full) {
    if (OBS64) {
      double2* out = reinterpret_cast<double2*>(static_cast<double*>(p.obs) + row0 * d.n_obs);
      const int n2 = ROWS * d.n_obs / 2;
      for (int i = tid; i < n2; i += THREADS) {
        int e = 2 * i, r = e / d.n_obs, c = e - r * d.n_obs;
        double2 o;
        o.x = obs_val(r, c);
        if (++c == d.n_obs) { c = 0; ++r; }
        o.y = obs_val(r, c);
        out[i] = o;
      }
    } else {
      float4* out = reinterpret_cast<float4*>(static_cast<float*>(p.obs) + row0 * d.n_obs);
      const int n4 = ROWS * d.n_obs / 4;
      for (int i = tid; i < n4; i += THREADS) {
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

  // ---- control tile: un-normalise, clamp to ctrlrange, scatter to actuator order
  if (p.ctrl) {
    auto ctrl_val = [&](int r, int j) -> double {
      const int k = t_csrc[j];
      if (k < 0) return 0.0;
      double u = (double)sa[r * d.n_act + k] * t_act[d.n_act + k] + t_act[k];
      const double lo = t_act[2 * d.n_act + k], hi = t_act[3 * d.n_act + k];
      if (u < lo) u = lo;
      if (u > hi) u = hi;
      return u;
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

  // ---- flags: 4 rows per dword
  __syncthreads();
  if (full) {
    const unsigned* a4 = reinterpret_cast<const unsigned*>(s_abs);
    const unsigned* c4 = reinterpret_cast<const unsigned*>(s_code);
    for (int i = tid; i < ROWS / 4; i += THREADS) {
      reinterpret_cast<unsigned*>(p.absorbing + row0)[i] = a4[i];
      if (p.fall_code) reinterpret_cast<unsigned*>(p.fall_code + row0)[i] = c4[i];
    }
  } else {
    for (int r = tid; r < rows; r += THREADS) {
      p.absorbing[row0 + r] = s_abs[r];
      if (p.fall_code) p.fall_code[row0 + r] = s_code[r];
    }
  }
}

using H1Dims = StaticDims<17, 17, 0, 11, 11, 32>;

template <int ROWS, class D>
int launch(oly_ctx* ctx, const IlArgs& a, int out_flags, hipStream_t s) {
  const IlDev& h = ctx->il_host;
  Carve<ROWS, D> cv(h.nq, h.nv, h.n_grf, h.n_act, h.nu, h.n_obs);
  const long tiles = (a.R + ROWS - 1) / ROWS;
  if (tiles > 0x7fffffffL) OLY_FAIL(ctx, OLY_EINVAL, "oly_il_step: too many rows");
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
  }
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
  const bool is_h1 = h.nq == 17 && h.nv == 17 && h.n_grf == 0 && h.n_act == 11 && h.nu == 11 && h.n_obs == 32;
  if (is_h1) return launch<128, H1Dims>(ctx, a, out_flags, oly_s(stream));
  return launch<64, DynDims>(ctx, a, out_flags, oly_s(stream));
}
