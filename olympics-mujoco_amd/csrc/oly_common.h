// Internal definitions shared by the HIP translation units of libolympic_hip.so (gfx950).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/olympic_hip.h"

// ---------------------------------------------------------------- device-side tables
// One copy per ctx in device memory; kernels receive a pointer and read the fields with
// wave-uniform (scalar) loads, or stage the per-column tables in LDS.

struct IlDev {
  int nq, nv, n_pos, n_vel, n_drop, n_grf, n_act, nu, n_obs, n_fall;
  int reward_type, reward_idx, use_absorbing, reward_sidx;  // *_sidx: staged-row element
  double target_velocity;
  // created-observation column c reads staged row element src[c]; the staged row is
  // [qpos (nq) | qvel (nv) | grf (n_grf)] and grf columns are divided by 1000.
  short src[OLY_MAX_OBS];
  int fall_idx[OLY_MAX_FALL], fall_sidx[OLY_MAX_FALL];
  double fall_lo[OLY_MAX_FALL], fall_hi[OLY_MAX_FALL];
  short ctrl_src[OLY_MAX_ACT];  // actuator j <- action slot, or -1
  double act_mean[OLY_MAX_ACT], act_delta[OLY_MAX_ACT], ctrl_lo[OLY_MAX_ACT], ctrl_hi[OLY_MAX_ACT];
};

struct A3Dev {
  int nq, nv, nu, period, delay_frames, n_obs, pad0, pad1;
  double target_radius, mass, goal_height_ref, goal_speed_ref;
  double clock_lut[4 * OLY_MAX_PERIOD];
  double motor_offset[16], gear[16];
};

struct ContactDev {
  int ngeom, floor_body, rfoot_body, lfoot_body;
  int* geom_bodyid;  // device [ngeom]
};

struct GrfDev {
  int ngeom, n_pairs;
  int pair_a[OLY_MAX_GRF_PAIRS], pair_b[OLY_MAX_GRF_PAIRS];
  int* geom_group;  // device [ngeom]
};

struct TrajDev {
  int n_keys, n_traj, len, pad;
  double* rows;  // device [n_traj, len, n_keys] (sample-major copy of the reference table)
};

#define OLY_STATS_MAX_BLOCKS 1024

struct oly_ctx {
  int device;
  char err[512];
  IlDev* il_dev;
  IlDev il_host;
  bool il_ok;
  A3Dev* a3_dev;
  A3Dev a3_host;
  bool a3_ok;
  ContactDev contact;
  bool contact_ok;
  TrajDev traj;
  bool traj_ok;
  GrfDev grf;
  bool grf_ok;
  int* grf_group_host;          // host copy of grf.geom_group (the batcher packs contacts on the host)
  double* stats_ws;  // device [OLY_STATS_MAX_BLOCKS * 2 * OLY_MAX_OBS... ] partial sums
  size_t stats_ws_bytes;
  bool mlp_attr_done = false;   // dynamic-LDS limit of the fused MLP kernel raised on this device
  bool disc_attr_done = false;  // same for the fused discriminator kernel (K12)
  bool roll_attr_done = false;  // same for the persistent rollout kernel (K13)
  unsigned upd_attr_done = 0;   // same for the update kernel's instantiations (K14)
  unsigned scan_attr_done = 0;  // dynamic-LDS limit of the pipelined scan kernels raised on this device
  int num_cu;
};

#define OLY_FAIL(ctx, code, ...)                                \
  do {                                                          \
    if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); \
    return (code);                                              \
  } while (0)

#define OLY_HIP(ctx, expr)                                                             \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess)                                                             \
      OLY_FAIL(ctx, OLY_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
               __FILE__, __LINE__);                                                    \
  } while (0)

#define OLY_LAUNCH_CHECK(ctx, what)                                                           \
  do {                                                                                        \
    hipError_t e__ = hipGetLastError();                                                       \
    if (e__ != hipSuccess)                                                                    \
      OLY_FAIL(ctx, OLY_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e__));       \
  } while (0)

// K7's finishing step over ctx->stats_ws[0 .. 2*nblocks) -> (n, sum, sumsq); used by K6's fused statistics
int oly_stats_finish(oly_ctx* ctx, int nblocks, int64_t n, double* stats3_out, oly_stream stream);
// oly_a3_step over the host batcher's compact staging (compact_base: qpos = [N,4] base quaternion, qvel = [N,3])
int oly_a3_step_strided(oly_ctx* ctx, int N, const oly_a3_inputs* in, const oly_a3_state* st, void* obs, float* rew6,
                        float* reward, uint8_t* done, int out_flags, int compact_base, oly_stream stream);

static inline hipStream_t oly_s(oly_stream s) { return reinterpret_cast<hipStream_t>(s); }

// 64-wide wavefront helpers (gfx950).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
