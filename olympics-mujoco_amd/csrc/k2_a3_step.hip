// K2 + K5: StickFigureA3 reinforcement-learning step (A1 / Jvrc share the code).
//   WalkingTask.step           tasks/walking_task.py:246-293 (+ :228-244, :184-225)
//   WalkingTask.calc_reward    :74-110, step_reward :56-72, tasks/rewards.py:27-40,65-102,121-126
//   WalkingTask.done           :298-319
//   StickFigureA3.get_obs      real_humanoid_robots/StickFigureA3.py:144-178
//   robot.JVRC.step target     environments/robot.py:88-95, PD torque :109-115 with
//                              MujocoRobotInterface.step_pd mujoco_robot_interface.py:425-443
//
// One lane per environment for the task logic (integer state machine + ~10 transcendental
// calls in fp64); the 41-wide observation rows are assembled in LDS and leave as a dense
// stream.  Integer state (phase, t1, t2, frame counters, flags) is bit-exact; every fp64
// expression keeps the reference's evaluation order (compiled with -ffp-contract=off).
// Bound: HBM, ~930 B read + ~257 B written per env.
#include <cstdlib>
#include <type_traits>

#include "a3_vec_core.h"
#include "oly_common.h"

namespace {
constexpr int THREADS = 128;
constexpr double PI = 3.141592653589793;
constexpr int K2_LANES16_MAX_N = 16384;    // up to here the 16-lane kernel wins (tools/time_k2.py, profiles/r03/k2_lane_layouts.csv)

struct A3Args {
  const A3Dev* md;
  int N;
  // where the base quaternion / base angular velocity of env n sit inside in.qpos / in.qvel: row stride and
  // offset in doubles.  (nq, 3) / (nv, 3) for the full rows of the C ABI; (4, 0) / (3, 0) for the host batcher's
  // compact staging, which ships only the seven numbers get_obs reads.
  int qpos_stride, qpos_off, qvel_stride, qvel_off;
  oly_a3_inputs in;
  oly_a3_state st;
  void* obs;
  float* rew6;
  float* reward;
  uint8_t* done;
};

__device__ __forceinline__ void quat2mat(const double* q, double R[3][3]) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double nq = w * w + x * x + y * y + z * z;
  if (nq < 2.220446049250313e-16) {
    R[0][0] = 1; R[0][1] = 0; R[0][2] = 0;
    R[1][0] = 0; R[1][1] = 1; R[1][2] = 0;
    R[2][0] = 0; R[2][1] = 0; R[2][2] = 1;
    return;
  }
  const double s = 2.0 / nq;
  const double X = x * s, Y = y * s, Z = z * s;
  const double wX = w * X, wY = w * Y, wZ = w * Z;
  const double xX = x * X, xY = x * Y, xZ = x * Z;
  const double yY = y * Y, yZ = y * Z, zZ = z * Z;
  R[0][0] = 1.0 - (yY + zZ); R[0][1] = xY - wZ;         R[0][2] = xZ + wY;
  R[1][0] = xY + wZ;         R[1][1] = 1.0 - (xX + zZ); R[1][2] = yZ - wX;
  R[2][0] = xZ - wY;         R[2][1] = yZ + wX;         R[2][2] = 1.0 - (xX + yY);
}

__device__ __forceinline__ double norm3d(double a0, double a1, double a2) {
  return sqrt(a0 * a0 + a1 * a1 + a2 * a2);
}

template <bool OBS64>
__global__ __launch_bounds__(THREADS) void a3_step_kernel(A3Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // the tile is staged in the OUTPUT type (the narrowing happens once, at the LDS store): half
  // the LDS of a double tile for the default f32 observation -> twice the resident workgroups
  using obs_t = typename std::conditional<OBS64, double, float>::type;
  obs_t* s_obs = reinterpret_cast<obs_t*>(lds_raw);  // [THREADS][n_obs]
  const A3Dev* __restrict__ m = p.md;
  const int n_obs = m->n_obs, nu = m->nu, period = m->period;
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * THREADS;
  const int n = n0 + tid;
  if (n < p.N) {
    const double* lf = p.in.lf_pos + 3 * (size_t)n;
    const double* rf = p.in.rf_pos + 3 * (size_t)n;
    const double lf0 = lf[0], lf1 = lf[1], lf2 = lf[2], rf0 = rf[0], rf1 = rf[1], rf2 = rf[2];
    const double* seq = p.st.sequence + (size_t)n * OLY_MAX_SEQ * 4;
    const int mode = p.st.mode[n];
    const int seq_len = p.st.seq_len[n];

    // ---- WalkingTask.step
    int phase = p.st.phase[n] + 1;
    if (phase >= period) phase = 0;
    int t1 = p.st.t1[n], t2 = p.st.t2[n];
    t1 = min(max(t1, 0), OLY_MAX_SEQ - 1);
    t2 = min(max(t2, 0), OLY_MAX_SEQ - 1);
    double tx = seq[4 * t1], ty = seq[4 * t1 + 1], tz = seq[4 * t1 + 2];
    const double dl = norm3d(lf0 - tx, lf1 - ty, lf2 - tz);
    const double dr = norm3d(rf0 - tx, rf1 - ty, rf2 - tz);
    int reached, frames = p.st.reached_frames[n];
    if (dl < m->target_radius || dr < m->target_radius) {
      reached = 1;
      frames += 1;
    } else {
      reached = 0;
      frames = 0;
    }
    if (reached && frames >= m->delay_frames) {  // update_target_steps
      t1 = t2;
      t2 += 1;
      if (t2 == seq_len) t2 = seq_len - 1;
      t2 = min(max(t2, 0), OLY_MAX_SEQ - 1);
      reached = 0;
      frames = 0;
    }
    p.st.phase[n] = phase;
    p.st.t1[n] = t1;
    p.st.t2[n] = t2;
    p.st.reached_frames[n] = frames;
    p.st.target_reached[n] = (uint8_t)reached;

    // ---- update_goal_steps
    const double* rp = p.in.root_pos + 3 * (size_t)n;
    const double* rqp = p.in.root_quat + 4 * (size_t)n;
    const double rp0 = rp[0], rp1 = rp[1], rp2 = rp[2];
    double rq[4] = {rqp[0], rqp[1], rqp[2], rqp[3]};
    double goal[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (mode != OLY_MODE_STANDING) {
      double R[3][3];
      quat2mat(rq, R);
      const int tt[2] = {t1, t2};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const double* s = seq + 4 * tt[i];
        const double d0 = s[0] - rp0, d1 = s[1] - rp1, d2 = s[2] - rp2;
        goal[0 + i] = R[0][0] * d0 + R[1][0] * d1 + R[2][0] * d2;
        goal[2 + i] = R[0][1] * d0 + R[1][1] * d1 + R[2][1] * d2;
        goal[4 + i] = R[0][2] * d0 + R[1][2] * d1 + R[2][2] * d2;
        const double c = cos(s[3]), sn = sin(s[3]);
        const double m00 = R[0][0] * c + R[1][0] * sn;
        const double m10 = R[0][1] * c + R[1][1] * sn;
        goal[6 + i] = atan2(m10, m00);
      }
    }
    double* gout = p.st.goal + 8 * (size_t)n;
#pragma unroll
    for (int i = 0; i < 8; ++i) gout[i] = goal[i];

    // ---- calc_reward
    double c_rfrc, c_rvel, c_lfrc, c_lvel;
    if (mode == OLY_MODE_STANDING) {
      c_rfrc = 1.0; c_lfrc = 1.0; c_rvel = -1.0; c_lvel = -1.0;
    } else {
      c_rfrc = m->clock_lut[0 * period + phase];
      c_rvel = m->clock_lut[1 * period + phase];
      c_lfrc = m->clock_lut[2 * period + phase];
      c_lvel = m->clock_lut[3 * period + phase];
    }
    const double max_frc = m->mass * 9.8 * 0.5;
    double nl = fmin(p.in.grf_l[n], max_frc) / max_frc;
    double nr = fmin(p.in.grf_r[n], max_frc) / max_frc;
    nl *= 2; nl -= 1; nr *= 2; nr -= 1;
    const double frc = (tan(PI / 4 * c_lfrc * nl) + tan(PI / 4 * c_rfrc * nr)) / 2;
    const double* lv = p.in.lf_vel + 3 * (size_t)n;
    const double* rv = p.in.rf_vel + 3 * (size_t)n;
    double vl = fmin(norm3d(lv[0], lv[1], lv[2]), 0.2) / 0.2;
    double vr = fmin(norm3d(rv[0], rv[1], rv[2]), 0.2) / 0.2;
    vl *= 2; vl -= 1; vr *= 2; vr -= 1;
    const double vel = (tan(PI / 4 * c_lvel * vl) + tan(PI / 4 * c_rvel * vr)) / 2;
    const double yaw = seq[4 * t1 + 3];
    const double tq0 = cos(yaw / 2.0), tq3 = sin(yaw / 2.0);
    const double ip = tq0 * rq[0] + 0.0 * rq[1] + 0.0 * rq[2] + tq3 * rq[3];
    const double orient = exp(-(10 * (1 - ip * ip)));
    const double contact_point = (p.in.n_r[n] > 0 || p.in.n_l[n] > 0) ? p.in.min_z[n] : 0.0;
    double err = fabs((rp2 - contact_point) - m->goal_height_ref);
    const double deadzone = 0.01 + 0.05 * m->goal_speed_ref;
    if (err < deadzone) err = 0;
    const double height = exp(-40 * (err * err));
    tx = seq[4 * t1]; ty = seq[4 * t1 + 1]; tz = seq[4 * t1 + 2];
    const double fd = fmin(norm3d(lf0 - tx, lf1 - ty, lf2 - tz), norm3d(rf0 - tx, rf1 - ty, rf2 - tz));
    const double hit = reached ? exp(-fd / 0.25) : 0.0;
    const double mpx = (seq[4 * t1] + seq[4 * t2]) / 2, mpy = (seq[4 * t1 + 1] + seq[4 * t2 + 1]) / 2;
    const double rx = rp0 - mpx, ry = rp1 - mpy;
    const double progress = exp(-sqrt(rx * rx + ry * ry) / 2);
    const double step_r = 0.8 * hit + 0.2 * progress;
    const double* hp = p.in.head_pos + 3 * (size_t)n;
    const double hx = hp[0] - rp0, hy = hp[1] - rp1;
    const double hn = sqrt(hx * hx + hy * hy);
    const double upper = exp(-10 * (hn * hn));
    double rew[6];
    rew[0] = 0.150 * frc;
    rew[1] = 0.150 * vel;
    rew[2] = 0.050 * orient;
    rew[3] = 0.050 * height;
    rew[4] = 0.450 * step_r;
    rew[5] = 0.050 * upper;
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      tot += rew[i];
      p.rew6[6 * (size_t)n + i] = (float)rew[i];
    }
    p.reward[n] = (float)tot;

    // ---- done
    const double foot_z = fmin(lf2, rf2);
    p.done[n] = (uint8_t)(((rp2 - foot_z) < 0.6) || p.in.bad[n]);

    // ---- get_obs (root part; motor columns are filled cooperatively below)
    const double* qpos = p.in.qpos + (size_t)n * p.qpos_stride + p.qpos_off - 3;
    const double* qvel = p.in.qvel + (size_t)n * p.qvel_stride + p.qvel_off - 3;
    double bq[4] = {qpos[3], qpos[4], qpos[5], qpos[6]};
    double Rb[3][3];
    quat2mat(bq, Rb);
    const double cy = sqrt(Rb[0][0] * Rb[0][0] + Rb[1][0] * Rb[1][0]);
    double roll, pitch;
    if (cy > 4.0 * 2.220446049250313e-16) {
      roll = atan2(Rb[2][1], Rb[2][2]);
      pitch = atan2(-Rb[2][0], cy);
    } else {
      roll = atan2(-Rb[1][2], Rb[1][1]);
      pitch = atan2(-Rb[2][0], cy);
    }
    const double ci = cos(roll / 2.0), si = sin(roll / 2.0), cj = cos(pitch / 2.0), sj = sin(pitch / 2.0);
    obs_t* o = s_obs + (size_t)tid * n_obs;
    o[0] = (obs_t)(ci * cj);
    o[1] = (obs_t)(si * cj);
    o[2] = (obs_t)(ci * sj);
    o[3] = (obs_t)(-(si * sj));
    o[4] = (obs_t)qvel[3]; o[5] = (obs_t)qvel[4]; o[6] = (obs_t)qvel[5];
    const double ang = 2 * PI * phase / (double)period;
    o[7 + 2 * nu] = (obs_t)sin(ang);
    o[8 + 2 * nu] = (obs_t)cos(ang);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[9 + 2 * nu + i] = (obs_t)goal[i];
  }
  // motor position / velocity columns: dense [N,nu] streams, one element per lane
  const int rows = min(THREADS, p.N - n0);
  for (int e = tid; e < rows * nu; e += THREADS) {
    const int r = e / nu, i = e - r * nu;
    const double g = m->gear[i];
    s_obs[(size_t)r * n_obs + 7 + i] = (obs_t)(p.in.act_len[(size_t)n0 * nu + e] / g);
    s_obs[(size_t)r * n_obs + 7 + nu + i] = (obs_t)(p.in.act_vel[(size_t)n0 * nu + e] / g);
  }
  __syncthreads();
  for (int e = tid; e < rows * n_obs; e += THREADS) {
    static_cast<obs_t*>(p.obs)[(size_t)n0 * n_obs + e] = s_obs[e];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same step with SIXTEEN lanes per environment (K10's layout) for the sizes a vectorised env.step() really has.
// One lane per environment is a dependent chain of ~27 fp64 libm calls and ~1100 libm-free fp64 instructions on 128
// threads per workgroup: 4096 environments are 32 workgroups on 256 CUs and 13.9 us (16384: 14.9).  Here a
// 256-thread workgroup owns 16 environments: the libm-free arithmetic runs as four per-wave tasks (a3_vec_core.h:
// level1_tasks), every libm call is one lane's job in two rounds regrouped by function across the waves, the
// observation row is assembled by the environment's 16 lanes: 8.4 us at 4096, 12.9 at 16384.  Past that the 37 KB
// of LDS per 16 environments limits the resident groups and the lane-per-environment kernel is ahead (65536: 23.5 us
// against 40.7), so oly_a3_step picks by N.  Same expressions on the same inputs as a3_step_kernel (the arithmetic
// K10 and K13 share): identical results, byte for byte (tests/test_gpu_parity.py: the two kernels against each
// other, and every K2 test under either).
template <bool OBS64>
__global__ __launch_bounds__(256) void a3_step16_kernel(A3Args p) {
  using namespace oly_a3v;
  using obs_t = typename std::conditional<OBS64, double, float>::type;
  constexpr int SLOTS = A3V_SLOTS, EPW = A3V_EPW, MAXOBS = 7 + 2 * 16 + 10;
  __shared__ double s_env[EPW * L_ENV];
  __shared__ double seqs[EPW * A3V_SEQW];
  __shared__ double s_arg[EPW * SLOTS * 2];
  __shared__ uint8_t s_cls[EPW * SLOTS];
  __shared__ int s_int[EPW * SI_N];
  __shared__ obs_t s_obs[EPW * (MAXOBS + 1)];
  const A3Dev* __restrict__ m = p.md;
  const int nu = m->nu, n_obs = m->n_obs, period = m->period;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, slot = lane & (SLOTS - 1);
  const int el = wave * 4 + grp;
  const int row0 = blockIdx.x * EPW;
  const int n = row0 + el;
  const bool env_ok = n < p.N;
  const int rows = min(EPW, p.N - row0);
  double* se = s_env + el * L_ENV;
  double* sq = seqs + el * A3V_SEQW;

  // ---- every load up front, one piece per lane; staged in LDS
  if (env_ok) {
    const size_t r3 = (size_t)n * 3, r4 = (size_t)n * 4;
    double va;
    if (slot < 4) va = p.in.root_quat[r4 + slot];
    else if (slot < 7) va = p.in.root_pos[r3 + slot - 4];
    else if (slot < 10) va = p.in.head_pos[r3 + slot - 7];
    else if (slot < 13) va = p.in.lf_pos[r3 + slot - 10];
    else va = p.in.rf_pos[r3 + slot - 13];
    se[slot] = va;
    if (slot < 3) se[L_LV + slot] = p.in.lf_vel[r3 + slot];
    else if (slot < 6) se[L_RV + slot - 3] = p.in.rf_vel[r3 + slot - 3];
    else if (slot < 10) se[L_BQ + slot - 6] = p.in.qpos[(size_t)n * p.qpos_stride + p.qpos_off + slot - 6];
    else if (slot < 13) se[L_AV + slot - 10] = p.in.qvel[(size_t)n * p.qvel_stride + p.qvel_off + slot - 10];
    if (slot < nu) {
      se[L_AL + slot] = p.in.act_len[(size_t)n * nu + slot];
      se[L_AVL + slot] = p.in.act_vel[(size_t)n * nu + slot];
    }
#pragma unroll
    for (int q = 0; q < A3V_SEQW / SLOTS; ++q) sq[slot + SLOTS * q] = p.st.sequence[(size_t)n * A3V_SEQW + slot + SLOTS * q];
    if (slot == 0) {
      int* si = s_int + el * SI_N;
      si[I_PHASE0] = p.st.phase[n];
      si[I_T1] = min(max(p.st.t1[n], 0), OLY_MAX_SEQ - 1);
      si[I_T2] = min(max(p.st.t2[n], 0), OLY_MAX_SEQ - 1);
      si[I_FRAMES] = p.st.reached_frames[n];
      si[I_MODE] = p.st.mode[n];
      si[I_SEQLEN] = p.st.seq_len[n];
      si[I_TLEN] = 0;
      si[I_RC] = 0;
      si[I_BAD] = p.in.bad[n];
      si[I_HAVEC] = (p.in.n_r[n] > 0 || p.in.n_l[n] > 0);
      se[L_GR] = p.in.grf_r[n];
      se[L_GL] = p.in.grf_l[n];
      se[L_MZ] = p.in.min_z[n];
    }
  }
  __syncthreads();

  // ---- level 1 (no rollout bookkeeping: last_step = true means no cut-driven reset)
  {
    Level1Ctx lc;
    lc.m = m; lc.s_env = s_env; lc.seqs = seqs; lc.s_int = s_int; lc.s_arg = s_arg; lc.s_cls = s_cls;
    lc.s_lut = m->clock_lut; lc.period = period; lc.rows = rows; lc.last_step = true; lc.max_traj_len = 0x7fffffff;
    lc.pool = nullptr; lc.pool_depth = 1; lc.row0 = row0;
    level1_tasks(lc, wave, lane);
  }
  __syncthreads();
  double r0, r1;
  const int ee = lane & 15;
  {  // ---- libm round 1: wave 0 sin-cos, 1 tan, 2 exp, 3 atan2 of the group's 16 environments
    constexpr int R1_TASK[4][4] = {{0, 1, 6, 13}, {2, 3, 4, 5}, {7, 8, 9, 10}, {11, 12, -1, -1}};
    const int task = R1_TASK[wave][lane >> 4];
    if (task >= 0) {
      eval_task(s_cls[ee * SLOTS + task], s_arg[(ee * SLOTS + task) * 2], s_arg[(ee * SLOTS + task) * 2 + 1], r0, r1);
      s_env[ee * L_ENV + L_R1 + 2 * task] = r0;
      s_env[ee * L_ENV + L_R1 + 2 * task + 1] = r1;
    }
  }
  __syncthreads();

  // ---- back on the environment's own lanes
  const int* si_ = s_int + el * SI_N;
  const int t1 = env_ok ? si_[O_T1] : 0, t2 = env_ok ? si_[O_T2] : 0;      // (rows past N hold nothing)
  const bool walking = env_ok && si_[I_MODE] != OLY_MODE_STANDING;
  const double rq0 = se[L_RQ], rq1 = se[L_RQ + 1], rq2 = se[L_RQ + 2], rq3 = se[L_RQ + 3];
  const double rp0 = se[L_RP], rp1 = se[L_RP + 1], rp2 = se[L_RP + 2];
  double R[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R[i][j] = se[L_ROT + 3 * i + j];
  double goal[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (walking) {
    const int selA = 4 * t1, selB = 4 * t2;
    const double s1x = sq[selA], s1y = sq[selA + 1], s1z = sq[selA + 2];
    const double s2x = sq[selB], s2y = sq[selB + 1], s2z = sq[selB + 2];
    const double a0 = s1x - rp0, a1 = s1y - rp1, a2 = s1z - rp2;
    const double b0 = s2x - rp0, b1 = s2y - rp1, b2 = s2z - rp2;
    goal[0] = R[0][0] * a0 + R[1][0] * a1 + R[2][0] * a2;
    goal[2] = R[0][1] * a0 + R[1][1] * a1 + R[2][1] * a2;
    goal[4] = R[0][2] * a0 + R[1][2] * a1 + R[2][2] * a2;
    goal[1] = R[0][0] * b0 + R[1][0] * b1 + R[2][0] * b2;
    goal[3] = R[0][1] * b0 + R[1][1] * b1 + R[2][1] * b2;
    goal[5] = R[0][2] * b0 + R[1][2] * b1 + R[2][2] * b2;
  }
  {  // ---- round 2 arguments
    int cls = F_NONE;
    double a = 0.0, b = 0.0;
    switch (slot) {
      case 0:
      case 1:
        if (walking) {   // theta = mat2euler(R^T Rz(yaw))[2] = atan2(M10, M00)
          const double c = se[L_R1 + 2 * slot + 1], sn = se[L_R1 + 2 * slot];
          cls = F_ATAN2;
          a = R[0][1] * c + R[1][1] * sn;
          b = R[0][0] * c + R[1][0] * sn;
        }
        break;
      case 2: {          // body orientation: exp(-10 (1 - <q_ref, q>^2))
        const double tq0 = se[L_R1 + 2 * 6 + 1], tq3 = se[L_R1 + 2 * 6];
        const double ip = tq0 * rq0 + 0.0 * rq1 + 0.0 * rq2 + tq3 * rq3;
        cls = F_EXP;
        a = -(10 * (1 - ip * ip));
      } break;
      case 3: cls = F_SINCOS; a = se[L_R1 + 2 * 11] / 2.0; break;                    // roll / 2
      case 4: cls = F_SINCOS; a = se[L_R1 + 2 * 12] / 2.0; break;                    // pitch / 2
      default: break;
    }
    if (slot < 5) {
      s_arg[(el * SLOTS + slot) * 2] = a;
      s_arg[(el * SLOTS + slot) * 2 + 1] = b;
      s_cls[el * SLOTS + slot] = (uint8_t)(env_ok ? cls : F_NONE);
    }
  }
  __syncthreads();
  {  // ---- libm round 2: wave 0 sin-cos (roll / 2, pitch / 2), wave 1 atan2 (the goal yaws), wave 2 exp (orientation)
    constexpr int R2_TASK[4][4] = {{3, 4, -1, -1}, {0, 1, -1, -1}, {2, -1, -1, -1}, {-1, -1, -1, -1}};
    const int task = R2_TASK[wave][lane >> 4];
    if (task >= 0) {
      eval_task(s_cls[ee * SLOTS + task], s_arg[(ee * SLOTS + task) * 2], s_arg[(ee * SLOTS + task) * 2 + 1], r0, r1);
      s_env[ee * L_ENV + L_R2 + 2 * task] = r0;
      s_env[ee * L_ENV + L_R2 + 2 * task + 1] = r1;
    }
  }
  __syncthreads();

  // ---- combine: observation row (the output type is formed once, as a3_step_kernel does), rewards, state
  if (env_ok) {
    if (walking) {
      goal[6] = se[L_R2 + 0];
      goal[7] = se[L_R2 + 2];
    }
    obs_t* op = s_obs + el * (MAXOBS + 1);
    // every LDS value a lane needs is read up front, unconditionally (a read inside each `if (slot == k)` is a
    // serialised LDS round trip per branch)
    const double ci = se[L_R2 + 2 * 3 + 1], si = se[L_R2 + 2 * 3], cj = se[L_R2 + 2 * 4 + 1], sj = se[L_R2 + 2 * 4];
    const int xi = slot < 7 ? L_AV + max(slot, 4) - 4 : L_R1 + 2 * 13 + min(slot, 8) - 7;   // slots 4..6: qvel[3:6]; 7, 8: clock
    const double xv = se[xi];
    const int mi = min(slot, nu - 1);
    const double ql = se[L_AL + mi], qv = se[L_AVL + mi];
    const double lo = slot == 0 ? ci * cj : slot == 1 ? si * cj : slot == 2 ? ci * sj : slot == 3 ? -(si * sj) : xv;
    if (slot < 7) op[slot] = (obs_t)lo;
    if (slot < nu) {
      const double g = m->gear[slot];
      const bool unit = __ballot(g != 1.0) == 0;     // x / 1.0 == x: no fp64 division chain when every gear is 1
      op[7 + slot] = (obs_t)(unit ? ql : ql / g);
      op[7 + nu + slot] = (obs_t)(unit ? qv : qv / g);
    }
    if (slot == 7 || slot == 8) op[2 * nu + slot] = (obs_t)xv;
    if (slot >= 8) op[9 + 2 * nu + slot - 8] = (obs_t)goal[slot - 8];
    if (slot < 8) p.st.goal[8 * (size_t)n + slot] = goal[slot];
    if (slot == 0) {
      const int reached = si_[O_REACHED];
      const double frc = (se[L_R1 + 2 * 2] + se[L_R1 + 2 * 3]) / 2;
      const double vel = (se[L_R1 + 2 * 4] + se[L_R1 + 2 * 5]) / 2;
      const double orient = se[L_R2 + 2 * 2];
      const double height = se[L_R1 + 2 * 7];
      const double hit = reached ? se[L_R1 + 2 * 8] : 0.0;
      const double progress = se[L_R1 + 2 * 9];
      const double step_r = 0.8 * hit + 0.2 * progress;
      const double upper = se[L_R1 + 2 * 10];
      double rew[6];
      rew[0] = 0.150 * frc;
      rew[1] = 0.150 * vel;
      rew[2] = 0.050 * orient;
      rew[3] = 0.050 * height;
      rew[4] = 0.450 * step_r;
      rew[5] = 0.050 * upper;
      double tot = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        tot += rew[i];
        p.rew6[6 * (size_t)n + i] = (float)rew[i];
      }
      p.reward[n] = (float)tot;
      p.done[n] = (uint8_t)(si_[O_DONE] != 0);
      p.st.phase[n] = si_[O_PHASE];
      p.st.t1[n] = t1;
      p.st.t2[n] = t2;
      p.st.reached_frames[n] = si_[O_FRAMES];
      p.st.target_reached[n] = (uint8_t)reached;
    }
  }
  __syncthreads();
  for (int e = tid; e < rows * n_obs; e += 256) {
    const int r = e / n_obs, c = e - r * n_obs;
    static_cast<obs_t*>(p.obs)[(size_t)row0 * n_obs + e] = s_obs[r * (MAXOBS + 1) + c];
  }
}

__global__ void pd_target_kernel(const A3Dev* __restrict__ m, long total, const float* __restrict__ action,
                                 double* __restrict__ target) {
  const long stride = (long)gridDim.x * blockDim.x;
  const int nu = m->nu;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride)
    target[e] = (double)action[e] + m->motor_offset[e % nu];
}

__global__ void pd_torque_kernel(const A3Dev* __restrict__ m, long total, const double* __restrict__ kp,
                                 const double* __restrict__ kd, const double* __restrict__ target,
                                 const double* __restrict__ act_len, const double* __restrict__ act_vel,
                                 double* __restrict__ tau) {
  const long stride = (long)gridDim.x * blockDim.x;
  const int nu = m->nu;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int i = (int)(e % nu);
    const double g = m->gear[i];
    const double q = act_len[e] / g, qd = act_vel[e] / g;
    const double perror = target[e] - q, verror = 0.0 - qd;
    tau[e] = (kp[i] * perror + kd[i] * verror) / g;
  }
}

inline int blocks_for(long n, int t) {
  long b = (n + t - 1) / t;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
}  // namespace

extern "C" int oly_a3_configure(oly_ctx* ctx, const oly_a3_model* m) {
  if (!ctx || !m) return OLY_EINVAL;
  if (m->nu <= 0 || m->nu > 16 || m->nq < 7 || m->nv < 6 || m->period <= 0 || m->period > OLY_MAX_PERIOD ||
      !m->clock_lut || !m->motor_offset || !m->gear)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_a3_configure: bad model (nu=%d period=%d)", m->nu, m->period);
  ctx->a3_ok = false;
  A3Dev& h = ctx->a3_host;
  memset(&h, 0, sizeof(h));
  h.nq = m->nq; h.nv = m->nv; h.nu = m->nu; h.period = m->period; h.delay_frames = m->delay_frames;
  h.n_obs = 7 + 2 * m->nu + 10;
  h.target_radius = m->target_radius; h.mass = m->mass; h.goal_height_ref = m->goal_height_ref;
  h.goal_speed_ref = m->goal_speed_ref;
  memcpy(h.clock_lut, m->clock_lut, sizeof(double) * 4 * m->period);
  for (int i = 0; i < m->nu; ++i) {
    h.motor_offset[i] = m->motor_offset[i];
    h.gear[i] = m->gear[i];
    if (m->gear[i] == 0.0) OLY_FAIL(ctx, OLY_ERANGE, "oly_a3_configure: gear[%d] is zero", i);
  }
  OLY_HIP(ctx, hipSetDevice(ctx->device));
  OLY_HIP(ctx, hipMemcpy(ctx->a3_dev, &h, sizeof(h), hipMemcpyHostToDevice));
  ctx->a3_ok = true;
  return OLY_OK;
}

int oly_a3_step_strided(oly_ctx* ctx, int N, const oly_a3_inputs* in, const oly_a3_state* st, void* obs,
                        float* rew6, float* reward, uint8_t* done, int out_flags, int compact_base,
                        oly_stream stream);

extern "C" int oly_a3_step(oly_ctx* ctx, int N, const oly_a3_inputs* in, const oly_a3_state* st, void* obs,
                           float* rew6, float* reward, uint8_t* done, int out_flags, oly_stream stream) {
  return oly_a3_step_strided(ctx, N, in, st, obs, rew6, reward, done, out_flags, 0, stream);
}

// compact_base != 0: in->qpos is [N,4] (the base quaternion qpos[3:7]) and in->qvel [N,3] (qvel[3:6])
int oly_a3_step_strided(oly_ctx* ctx, int N, const oly_a3_inputs* in, const oly_a3_state* st, void* obs,
                        float* rew6, float* reward, uint8_t* done, int out_flags, int compact_base,
                        oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->a3_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_a3_step before oly_a3_configure");
  if (N < 0 || !in || !st) OLY_FAIL(ctx, OLY_EINVAL, "oly_a3_step: bad argument");
  if (N == 0) return OLY_OK;
  const void* req[] = {in->qpos, in->qvel, in->act_len, in->act_vel, in->lf_pos, in->rf_pos, in->lf_vel,
                       in->rf_vel, in->root_pos, in->root_quat, in->head_pos, in->grf_l, in->grf_r,
                       in->min_z, in->n_r, in->n_l, in->bad, st->phase, st->t1, st->t2,
                       st->reached_frames, st->target_reached, st->mode, st->seq_len, st->sequence,
                       st->goal, obs, rew6, reward, done};
  for (const void* q : req)
    if (!q) OLY_FAIL(ctx, OLY_EINVAL, "oly_a3_step: NULL pointer in inputs/state/outputs");
  A3Args a;
  a.md = ctx->a3_dev; a.N = N; a.in = *in; a.st = *st; a.obs = obs; a.rew6 = rew6; a.reward = reward;
  a.done = done;
  a.qpos_stride = compact_base ? 4 : ctx->a3_host.nq; a.qpos_off = compact_base ? 0 : 3;
  a.qvel_stride = compact_base ? 3 : ctx->a3_host.nv; a.qvel_off = compact_base ? 0 : 3;
  // 16 lanes per environment while that fills the chip better than one (OLY_K2_LANES = 1 / 16 forces either kernel)
  const char* lanes_env = getenv("OLY_K2_LANES");
  const int force_lanes = lanes_env ? atoi(lanes_env) : 0;
  const bool lanes16 = force_lanes == 16 || (force_lanes != 1 && N <= K2_LANES16_MAX_N);
  if (lanes16) {
    dim3 grid16((N + 15) / 16), block16(256);
    if (out_flags & OLY_OUT_OBS_F64) hipLaunchKernelGGL(a3_step16_kernel<true>, grid16, block16, 0, oly_s(stream), a);
    else hipLaunchKernelGGL(a3_step16_kernel<false>, grid16, block16, 0, oly_s(stream), a);
    OLY_LAUNCH_CHECK(ctx, "a3_step16_kernel");
    return OLY_OK;
  }
  const size_t lds = ((out_flags & OLY_OUT_OBS_F64) ? sizeof(double) : sizeof(float)) * THREADS * ctx->a3_host.n_obs;
  dim3 grid((N + THREADS - 1) / THREADS), block(THREADS);
  if (out_flags & OLY_OUT_OBS_F64)
    hipLaunchKernelGGL(a3_step_kernel<true>, grid, block, lds, oly_s(stream), a);
  else
    hipLaunchKernelGGL(a3_step_kernel<false>, grid, block, lds, oly_s(stream), a);
  OLY_LAUNCH_CHECK(ctx, "a3_step_kernel");
  return OLY_OK;
}

extern "C" int oly_a3_pd_target(oly_ctx* ctx, int N, const float* action, double* target,
                                oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->a3_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_a3_pd_target before oly_a3_configure");
  if (N < 0 || (N > 0 && (!action || !target))) OLY_FAIL(ctx, OLY_EINVAL, "oly_a3_pd_target: bad argument");
  if (N == 0) return OLY_OK;
  const long total = (long)N * ctx->a3_host.nu;
  hipLaunchKernelGGL(pd_target_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, oly_s(stream), ctx->a3_dev,
                     total, action, target);
  OLY_LAUNCH_CHECK(ctx, "pd_target_kernel");
  return OLY_OK;
}

extern "C" int oly_a3_pd_torque(oly_ctx* ctx, int N, const double* kp, const double* kd,
                                const double* target, const double* act_len, const double* act_vel,
                                double* tau, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->a3_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_a3_pd_torque before oly_a3_configure");
  if (N < 0 || !kp || !kd || (N > 0 && (!target || !act_len || !act_vel || !tau)))
    OLY_FAIL(ctx, OLY_EINVAL, "oly_a3_pd_torque: bad argument");
  if (N == 0) return OLY_OK;
  const long total = (long)N * ctx->a3_host.nu;
  hipLaunchKernelGGL(pd_torque_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, oly_s(stream), ctx->a3_dev,
                     total, kp, kd, target, act_len, act_vel, tau);
  OLY_LAUNCH_CHECK(ctx, "pd_torque_kernel");
  return OLY_OK;
}
