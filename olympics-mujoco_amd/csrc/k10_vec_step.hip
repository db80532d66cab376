// K10: one launch per vec step of the PPO rollout loop for N StickFigureA3 environments
// (rl/algos/ppo.py:169-196): the Gaussian sample from pre-drawn noise, memory.store, the PD
// target (K5), the foot-contact reduction (K3), WalkingTask.step / calc_reward / done / get_obs
// (K2), the episode-cut bookkeeping, the bootstrap side list and the device-side env.reset()
// of the cut environments (WalkingTask.reset from a pre-drawn record + transform_sequence).
//
// Regime: N = 4096 environments move ~10 MB per step = ~2 us of HBM time; the launch is
// LATENCY-bound, not bandwidth-bound.  The stand-alone K2 gives one lane a whole environment:
// ~27 dependent fp64 transcendental calls per lane = 24 us on 64 waves.  Here an environment is
// 16 lanes (the K3 layout: one lane per contact slot, 4 environments per wave, 16 per
// workgroup), and the transcendental calls are spread over lanes in two dependency rounds: the
// environment's lanes form ONE argument each, then the evaluation is regrouped by function over the
// workgroup's four waves (sin/cos, tan, exp, atan2), so a round costs ONE libm evaluation per wave
// instead of 18 / 9 in sequence on one lane.  The arithmetic per value is K2's, expression for expression (same libm entry
// points, -ffp-contract=off): observations, rewards and task state are bit-identical to
// oly_contact_reduce + oly_a3_step.
//
// The step index t and the readback row k are DEVICE counters (one private copy per workgroup, so
// no workgroup ever reads a counter another one has already advanced): the launch arguments are
// the same for every step and the launch can be replayed from a HIP graph.
#include "oly_common.h"
#include "a3_vec_core.h"

using namespace oly_a3v;
namespace {
#ifndef OLY_K10_THREADS
#define OLY_K10_THREADS 256   // 128 (8 environments, two workgroups per CU) measured slower: 15.7 vs 14.1 us
#endif
constexpr int THREADS = OLY_K10_THREADS;
constexpr int SLOTS = 16;               // lanes per environment
constexpr int EPW = THREADS / SLOTS;    // environments per workgroup
constexpr int MAX_NU = 16;
constexpr int MAX_NOBS = 7 + 2 * MAX_NU + 10;

// per-environment LDS scratch (doubles)
enum {
  L_RQ = 0,      // root quat 4
  L_RP = 4,      // root pos 3
  L_HP = 7,      // head pos 3
  L_LF = 10,     // lf pos 3
  L_RF = 13,     // rf pos 3
  L_LV = 16,     // lf vel 3
  L_RV = 19,     // rf vel 3
  L_BQ = 22,     // body quat qpos[3:7]
  L_AV = 26,     // qvel[3:6]
  L_AL = 29,     // act_len 16
  L_AVL = 45,    // act_vel 16
  L_SEQ = 61,    // the whole step sequence [20][4]
  L_R1 = 141,    // round-1 results [16][2]
  L_R2 = 173,    // round-2 results [16][2]
  L_ENV = 205
};

struct VecArgs {
  const A3Dev* md;
  ContactDev cd;
  int N, flags;
  oly_a3_blocks b;
  oly_a3_state st;
  oly_a3_rollout ro;
};

__global__ __launch_bounds__(THREADS) void a3_vec_kernel(VecArgs p) {
  __shared__ double s_env[EPW][L_ENV];
  __shared__ double s_arg[EPW][SLOTS][2];   // libm arguments of the current round, by (environment, task)
  __shared__ uint8_t s_cls[EPW][SLOTS];
  __shared__ float s_pre[EPW][MAX_NOBS + 1], s_post[EPW][MAX_NOBS + 1];
  const A3Dev* __restrict__ m = p.md;
  const int nu = m->nu, n_obs = m->n_obs, period = m->period, nq = m->nq, nv = m->nv;
  const int N = p.N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, slot = lane & (SLOTS - 1);
  const int el = wave * 4 + grp;
  const int row0 = blockIdx.x * EPW;
  const int n = row0 + el;
  const bool env_ok = n < N;
  const int rows = min(EPW, N - row0);
  const bool reset_all = (p.flags & OLY_VSTEP_RESET_ALL) != 0;
  const int t = p.ro.ctr[2 * blockIdx.x];
  const int k = p.ro.ctr[2 * blockIdx.x + 1];
  const int kk = (int)((unsigned)k % (unsigned)p.b.K);
  const int T = p.ro.T;
  // t lives on the device, so the host cannot check it: a replay past the rollout's T rows (or a caller that forgot
  // to rewind the counters) must not write beyond the buffers.  Touch nothing, advance nothing, leave a sticky mark
  // in the word behind the per-workgroup counters (vecstep._finalize turns it into an error).
  if (!reset_all && (unsigned)t >= (unsigned)T) {
    if (tid == 0) p.ro.ctr[2 * gridDim.x] = 1;
    return;
  }
  const size_t tN = (size_t)t * N;
  const size_t kN = (size_t)kk * N;
  double* se = s_env[el];

  // ---------------------------------------------------------------- every independent global load, up front
  // The launch is latency-bound: nothing below waits for a load that could have been issued here.  Contact
  // slots and the whole step sequence are fetched unconditionally (a few hundred bytes per environment)
  // instead of behind the counts / indices that select from them.
  const int C = p.b.C;
  const int passes = (C + SLOTS - 1) / SLOTS;
  int nc_raw = 0, phase0 = 0, t1 = 0, t2 = 0, frames = 0, mode = OLY_MODE_STANDING, seq_len = 1, tlen = 0, rc = 0;
  double va = 0.0, vb = 0.0, v_len = 0.0, v_vel = 0.0, vs[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  int g1_0 = -1, g2_0 = -1;
  double f0[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, pz0 = 0.0;
  int dst_b = -1;
  // the environment's NEXT reset record (header + its rows of the local step sequence), fetched for every
  // environment whether it will reset or not: behind `need_reset` these were two more exposed memory
  // latencies on the slowest workgroup of nearly every launch (some environment resets almost every step)
  int rec_mode = OLY_MODE_STANDING, rec_phase = 0, rec_len = 1;
  double rec_seq[(OLY_MAX_SEQ + SLOTS - 1) / SLOTS][4];
#pragma unroll
  for (int q = 0; q < (OLY_MAX_SEQ + SLOTS - 1) / SLOTS; ++q)
    rec_seq[q][0] = rec_seq[q][1] = rec_seq[q][2] = rec_seq[q][3] = 0.0;
  if (env_ok) {
    rc = p.ro.pool_count[n];
    const oly_a3_reset_record* rec = p.ro.pool + (size_t)n * p.ro.pool_depth + (unsigned)rc % (unsigned)p.ro.pool_depth;
    rec_mode = rec->mode;
    rec_phase = rec->phase;
    rec_len = rec->seq_len;
#pragma unroll
    for (int q = 0; q < (OLY_MAX_SEQ + SLOTS - 1) / SLOTS; ++q) {
      const int r = slot + SLOTS * q;
      if (r < OLY_MAX_SEQ) {
        rec_seq[q][0] = rec->seq[r][0]; rec_seq[q][1] = rec->seq[r][1];
        rec_seq[q][2] = rec->seq[r][2]; rec_seq[q][3] = rec->seq[r][3];
      }
    }
    // (everything above and the task state below is addressed without the step / readback counters, so it is
    // in flight before their values are needed; the readback rows, which need `kk`, follow)
    phase0 = p.st.phase[n];
    t1 = p.st.t1[n];
    t2 = p.st.t2[n];
    frames = p.st.reached_frames[n];
    mode = p.st.mode[n];
    seq_len = p.st.seq_len[n];
    tlen = p.ro.traj_len[n];
#pragma unroll
    for (int q = 0; q < 5; ++q) vs[q] = p.st.sequence[(size_t)n * OLY_MAX_SEQ * 4 + slot + SLOTS * q];
    nc_raw = p.b.ncon[kN + n];
    const size_t r3 = (kN + n) * 3, r4 = (kN + n) * 4;
    if (slot < 4) va = p.b.root_quat[r4 + slot];
    else if (slot < 7) va = p.b.root_pos[r3 + slot - 4];
    else if (slot < 10) va = p.b.head_pos[r3 + slot - 7];
    else if (slot < 13) va = p.b.lf_pos[r3 + slot - 10];
    else va = p.b.rf_pos[r3 + slot - 13];
    if (slot < 3) { vb = p.b.lf_vel[r3 + slot]; dst_b = L_LV + slot; }
    else if (slot < 6) { vb = p.b.rf_vel[r3 + slot - 3]; dst_b = L_RV + slot - 3; }
    else if (slot < 10) { vb = p.b.qpos[(kN + n) * nq + 3 + slot - 6]; dst_b = L_BQ + slot - 6; }
    else if (slot < 13) { vb = p.b.qvel[(kN + n) * nv + 3 + slot - 10]; dst_b = L_AV + slot - 10; }
    if (slot < nu) {
      v_len = p.b.act_len[(kN + n) * nu + slot];
      v_vel = p.b.act_vel[(kN + n) * nu + slot];
    }
    if (slot < C) {
      const size_t e0 = (kN + n) * C + slot;
      g1_0 = p.b.geom1[e0];
      g2_0 = p.b.geom2[e0];
#pragma unroll
      for (int q = 0; q < 6; ++q) f0[q] = p.b.force6[e0 * 6 + q];
      pz0 = p.b.cpos_z[e0];
    }
  }

  // ---------------------------------------------------------------- policy tail: loads now, stores at the very end
  // (the noise row eps[t] is touched for the first time here: its HBM latency hides behind the whole step)
  constexpr int ACT_PT = (EPW * MAX_NU + THREADS - 1) / THREADS;       // action elements per thread
  constexpr int OBS_PT = (EPW * MAX_NOBS + THREADS - 1) / THREADS;     // observation elements per thread
  float pt_act[ACT_PT], pt_mu[ACT_PT], pt_obs[OBS_PT], pt_val = 0.f;
  if (!reset_all) {
    const bool det = p.ro.deterministic != 0;
#pragma unroll
    for (int q = 0; q < ACT_PT; ++q) {
      const int e = tid + q * THREADS;
      pt_act[q] = pt_mu[q] = 0.f;
      if (e < rows * nu) {
        const int r = e / nu, j = e - r * nu;
        const size_t nn = (size_t)(row0 + r);
        const float mu = p.ro.mu[nn * nu + j];
        float a = mu;
        if (!det) {
          const float sc = p.ro.scale[j] * p.ro.eps[(tN + nn) * nu + j];
          a = mu + sc;
        }
        pt_act[q] = a;
        pt_mu[q] = mu;
      }
    }
#pragma unroll
    for (int q = 0; q < OBS_PT; ++q) {
      const int e = tid + q * THREADS;
      pt_obs[q] = (e < rows * n_obs) ? p.ro.state[(size_t)row0 * n_obs + e] : 0.f;
    }
    if (tid < rows) pt_val = p.ro.value[row0 + tid];
  }

  // ---------------------------------------------------------------- stage the environment's inputs in LDS
  if (env_ok) {
    se[slot] = va;
    if (dst_b >= 0) se[dst_b] = vb;
    if (slot < nu) {
      se[L_AL + slot] = v_len;
      se[L_AVL + slot] = v_vel;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) se[L_SEQ + slot + SLOTS * q] = vs[q];
  }
  t1 = min(max(t1, 0), OLY_MAX_SEQ - 1);
  t2 = min(max(t2, 0), OLY_MAX_SEQ - 1);

  // ---------------------------------------------------------------- K3: foot contacts of row kk
  const int nc = min(max(nc_raw, 0), C);
  int cnt_r = 0, cnt_l = 0;
  double sum_r = 0.0, sum_l = 0.0, mz = 0.0;
  bool have = false;
  for (int ps = 0; ps < passes; ++ps) {
    const int i = ps * SLOTS + slot;
    bool is_r = false, is_l = false;
    double nrm = 0.0, pz = 0.0;
    if (env_ok && i < nc) {
      int g1 = g1_0, g2 = g2_0;
      double f[6] = {f0[0], f0[1], f0[2], f0[3], f0[4], f0[5]};
      pz = pz0;
      if (ps > 0) {            // more than 16 contact slots: the later passes load on demand
        const size_t e = (kN + n) * C + i;
        g1 = p.b.geom1[e];
        g2 = p.b.geom2[e];
#pragma unroll
        for (int q = 0; q < 6; ++q) f[q] = p.b.force6[e * 6 + q];
        pz = p.b.cpos_z[e];
      }
      if (g1 >= 0 && g1 < p.cd.ngeom && g2 >= 0 && g2 < p.cd.ngeom) {
        const int b1 = p.cd.geom_bodyid[g1], b2 = p.cd.geom_bodyid[g2];
        is_r = (b1 == p.cd.floor_body) && (b2 == p.cd.rfoot_body);
        is_l = (b1 == p.cd.floor_body) && (b2 == p.cd.lfoot_body);
      }
      if (is_r || is_l) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) s += f[q] * f[q];
        nrm = sqrt(s);
      } else {
        pz = 0.0;
      }
    }
    const unsigned long long br = __ballot(is_r), bl = __ballot(is_l);
    const unsigned mr = (unsigned)((br >> (grp * SLOTS)) & 0xffffu);
    const unsigned ml = (unsigned)((bl >> (grp * SLOTS)) & 0xffffu);
    cnt_r += __popc(mr);
    cnt_l += __popc(ml);
    unsigned rem = mr | ml;   // in-order chain over the matching slots (contact order), as contact_kernel
    while (__any(rem != 0u)) {
      const int q = rem ? (__ffs((int)rem) - 1) : 0;
      const double vk = __shfl(nrm, grp * SLOTS + q, 64);
      const double zk = __shfl(pz, grp * SLOTS + q, 64);
      if (rem) {
        if ((mr >> q) & 1u) sum_r += vk;
        if ((ml >> q) & 1u) sum_l += vk;
        if (!have || zk < mz) mz = zk;
        have = true;
        rem &= rem - 1u;
      }
    }
  }
  const double grf_r = sum_r, grf_l = sum_l;
  const double min_z = have ? mz : 0.0;
  const bool bad = (cnt_r + cnt_l) != nc_raw;
  __syncthreads();

  // ---------------------------------------------------------------- level 1: everything without libm
  const double rq0 = se[L_RQ], rq1 = se[L_RQ + 1], rq2 = se[L_RQ + 2], rq3 = se[L_RQ + 3];
  const double rp0 = se[L_RP], rp1 = se[L_RP + 1], rp2 = se[L_RP + 2];
  const double lf0 = se[L_LF], lf1 = se[L_LF + 1], lf2 = se[L_LF + 2];
  const double rf0 = se[L_RF], rf1 = se[L_RF + 1], rf2 = se[L_RF + 2];

  // WalkingTask.step (walking_task.py:246-293)
  int phase = phase0 + 1;
  if (phase >= period) phase = 0;
  const double tx = se[L_SEQ + 4 * t1], ty = se[L_SEQ + 4 * t1 + 1], tz = se[L_SEQ + 4 * t1 + 2];
  const double dl = vnorm3(lf0 - tx, lf1 - ty, lf2 - tz);
  const double dr = vnorm3(rf0 - tx, rf1 - ty, rf2 - tz);
  int reached;
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
  const int selA = L_SEQ + 4 * t1, selB = L_SEQ + 4 * t2;   // sequence[t1] / sequence[t2] after the update
  const double s1x = se[selA], s1y = se[selA + 1], s1z = se[selA + 2], s1w = se[selA + 3];
  const double s2x = se[selB], s2y = se[selB + 1], s2z = se[selB + 2], s2w = se[selB + 3];

  double R[3][3];
  quat2mat(rq0, rq1, rq2, rq3, R);
  double goal[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool walking = mode != OLY_MODE_STANDING;
  if (walking) {
    const double a0 = s1x - rp0, a1 = s1y - rp1, a2 = s1z - rp2;
    const double b0 = s2x - rp0, b1 = s2y - rp1, b2 = s2z - rp2;
    goal[0] = R[0][0] * a0 + R[1][0] * a1 + R[2][0] * a2;
    goal[2] = R[0][1] * a0 + R[1][1] * a1 + R[2][1] * a2;
    goal[4] = R[0][2] * a0 + R[1][2] * a1 + R[2][2] * a2;
    goal[1] = R[0][0] * b0 + R[1][0] * b1 + R[2][0] * b2;
    goal[3] = R[0][1] * b0 + R[1][1] * b1 + R[2][1] * b2;
    goal[5] = R[0][2] * b0 + R[1][2] * b1 + R[2][2] * b2;
  }

  // calc_reward arguments (walking_task.py:74-110, tasks/rewards.py:27-40,65-102,121-126)
  double c_rfrc, c_rvel, c_lfrc, c_lvel;
  if (!walking) {
    c_rfrc = 1.0; c_lfrc = 1.0; c_rvel = -1.0; c_lvel = -1.0;
  } else {
    c_rfrc = m->clock_lut[0 * period + phase];
    c_rvel = m->clock_lut[1 * period + phase];
    c_lfrc = m->clock_lut[2 * period + phase];
    c_lvel = m->clock_lut[3 * period + phase];
  }
  const double max_frc = m->mass * 9.8 * 0.5;
  double nl = fmin(grf_l, max_frc) / max_frc;
  double nr = fmin(grf_r, max_frc) / max_frc;
  nl *= 2; nl -= 1; nr *= 2; nr -= 1;
  double vl = fmin(vnorm3(se[L_LV], se[L_LV + 1], se[L_LV + 2]), 0.2) / 0.2;
  double vr = fmin(vnorm3(se[L_RV], se[L_RV + 1], se[L_RV + 2]), 0.2) / 0.2;
  vl *= 2; vl -= 1; vr *= 2; vr -= 1;
  const double contact_point = (cnt_r > 0 || cnt_l > 0) ? min_z : 0.0;
  double err = fabs((rp2 - contact_point) - m->goal_height_ref);
  const double deadzone = 0.01 + 0.05 * m->goal_speed_ref;
  if (err < deadzone) err = 0;
  const double fd = fmin(vnorm3(lf0 - s1x, lf1 - s1y, lf2 - s1z), vnorm3(rf0 - s1x, rf1 - s1y, rf2 - s1z));
  const double mpx = (s1x + s2x) / 2, mpy = (s1y + s2y) / 2;
  const double rx = rp0 - mpx, ry = rp1 - mpy;
  const double hx = se[L_HP] - rp0, hy = se[L_HP + 1] - rp1;
  const double hn = sqrt(hx * hx + hy * hy);

  // done (walking_task.py:298-319) and the rollout's cut rule (ppo.py:178,189-196)
  const double foot_z = fmin(lf2, rf2);
  const bool done = ((rp2 - foot_z) < 0.6) || bad;
  const int len = tlen + 1;
  const bool cut = done || len >= p.ro.max_traj_len || t == T - 1;
  const bool need_reset = env_ok && (reset_all || (cut && t < T - 1));

  // get_obs: quat2euler(qpos[3:7]) (StickFigureA3.py:160)
  double Rb[3][3];
  quat2mat(se[L_BQ], se[L_BQ + 1], se[L_BQ + 2], se[L_BQ + 3], Rb);
  const double cyb = sqrt(Rb[0][0] * Rb[0][0] + Rb[1][0] * Rb[1][0]);
  const bool regular = cyb > 4.0 * EPS;
  const double roll_y = regular ? Rb[2][1] : -Rb[1][2];
  const double roll_x = regular ? Rb[2][2] : Rb[1][1];

  // env.reset(): the next pool record (mode / phase / local sequence), drawn on the host
  int new_mode = mode, new_phase = 0, new_len = seq_len;
  if (need_reset) {
    new_mode = rec_mode;
    new_phase = rec_phase;
    new_len = min(max(rec_len, 1), OLY_MAX_SEQ);
  }
  // root yaw for transform_sequence: quat2euler(root xquat)[2] = mat2euler's ak
  const double cyr = sqrt(R[0][0] * R[0][0] + R[1][0] * R[1][0]);

  // ---------------------------------------------------------------- round 1: one libm call per lane
  int cls = F_NONE;
  double a = 0.0, b = 0.0;
  switch (slot) {
    case 0: if (walking) { cls = F_SINCOS; a = s1w; } break;                       // goal yaw 1: cos / sin(theta)
    case 1: if (walking) { cls = F_SINCOS; a = s2w; } break;
    case 2: cls = F_TAN; a = PI / 4 * c_lfrc * nl; break;                          // foot-force clock terms
    case 3: cls = F_TAN; a = PI / 4 * c_rfrc * nr; break;
    case 4: cls = F_TAN; a = PI / 4 * c_lvel * vl; break;                          // foot-velocity clock terms
    case 5: cls = F_TAN; a = PI / 4 * c_rvel * vr; break;
    case 6: cls = F_SINCOS; a = s1w / 2.0; break;                                  // euler2quat(0,0,yaw) of the target
    case 7: cls = F_EXP; a = -40 * (err * err); break;                             // height
    case 8: cls = F_EXP; a = -fd / 0.25; break;                                    // target hit
    case 9: cls = F_EXP; a = -sqrt(rx * rx + ry * ry) / 2; break;                  // progress
    case 10: cls = F_EXP; a = -10 * (hn * hn); break;                              // upper body
    case 11: cls = F_ATAN2; a = roll_y; b = roll_x; break;                         // roll
    case 12: cls = F_ATAN2; a = -Rb[2][0]; b = cyb; break;                         // pitch
    case 13: cls = F_SINCOS; a = 2 * PI * phase / (double)period; break;           // clock
    case 14: if (need_reset && cyr > 4.0 * EPS) { cls = F_ATAN2; a = R[1][0]; b = R[0][0]; } break;   // root yaw
    default: if (need_reset) { cls = F_SINCOS; a = 2 * PI * new_phase / (double)period; } break;      // clock after reset
  }
  // The environment's lanes only FORM the arguments.  The evaluation is regrouped by function so that a
  // wave runs ONE libm body instead of diverging over five: wave 0 takes the sin/cos tasks of the
  // workgroup's 16 environments, wave 1 the tan tasks, wave 2 the exp tasks, wave 3 the atan2 tasks.
  s_arg[el][slot][0] = a;
  s_arg[el][slot][1] = b;
  s_cls[el][slot] = (uint8_t)(env_ok ? cls : F_NONE);
  __syncthreads();
  double r0, r1;
  {
#if OLY_K10_THREADS == 256   // 16 environments: one function per wave
    constexpr int R1_TASK[4][4] = {{0, 1, 6, 13}, {2, 3, 4, 5}, {7, 8, 9, 10}, {11, 12, 14, -1}};
    const int ee = lane & 15, task = R1_TASK[wave][lane >> 4];
#else                        // 8 environments, two workgroups per CU: two functions per wave
    constexpr int R1_TASK[2][8] = {{0, 1, 6, 13, 2, 3, 4, 5}, {7, 8, 9, 10, 11, 12, 14, -1}};
    const int ee = lane & 7, task = R1_TASK[wave][lane >> 3];
#endif
    if (task >= 0) {
      eval_task(s_cls[ee][task], s_arg[ee][task][0], s_arg[ee][task][1], r0, r1);
      s_env[ee][L_R1 + 2 * task] = r0;
      s_env[ee][L_R1 + 2 * task + 1] = r1;
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- round 2
  const double root_yaw = se[L_R1 + 2 * 14];
  cls = F_NONE;
  a = 0.0;
  b = 0.0;
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
    case 5: if (need_reset) { cls = F_SINCOS; a = root_yaw; } break;               // transform_sequence rotation
    default: break;
  }
  if (slot < 6) {
    s_arg[el][slot][0] = a;
    s_arg[el][slot][1] = b;
    s_cls[el][slot] = (uint8_t)(env_ok ? cls : F_NONE);
  }
  __syncthreads();
  {
    // wave 0: sin/cos (roll / 2, pitch / 2, root yaw, and round 1's clock-after-reset, slot 15, which only
    // needed level-1 values); wave 1: atan2 (the two goal yaws); wave 2: exp (orientation); wave 3: idle
#if OLY_K10_THREADS == 256
    constexpr int R2_TASK[4][4] = {{3, 4, 5, 15}, {0, 1, -1, -1}, {2, -1, -1, -1}, {-1, -1, -1, -1}};
    const int ee = lane & 15, task = R2_TASK[wave][lane >> 4];
#else
    constexpr int R2_TASK[2][8] = {{3, 4, 5, 15, -1, -1, -1, -1}, {0, 1, 2, -1, -1, -1, -1, -1}};
    const int ee = lane & 7, task = R2_TASK[wave][lane >> 3];
#endif
    if (task >= 0) {
      eval_task(s_cls[ee][task], s_arg[ee][task][0], s_arg[ee][task][1], r0, r1);
      const int dst = task == 15 ? L_R1 + 2 * 15 : L_R2 + 2 * task;
      s_env[ee][dst] = r0;
      s_env[ee][dst + 1] = r1;
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- combine, observation rows
  if (env_ok) {
    if (walking) {
      goal[6] = se[L_R2 + 0];
      goal[7] = se[L_R2 + 2];
    }
    float* op = s_pre[el];
    const double ci = se[L_R2 + 2 * 3 + 1], si = se[L_R2 + 2 * 3], cj = se[L_R2 + 2 * 4 + 1], sj = se[L_R2 + 2 * 4];
    if (slot == 0) op[0] = (float)(ci * cj);
    if (slot == 1) op[1] = (float)(si * cj);
    if (slot == 2) op[2] = (float)(ci * sj);
    if (slot == 3) op[3] = (float)(-(si * sj));
    if (slot >= 4 && slot < 7) op[slot] = (float)se[L_AV + slot - 4];
    if (slot < nu) {
      const double g = m->gear[slot];
      op[7 + slot] = (float)(se[L_AL + slot] / g);
      op[7 + nu + slot] = (float)(se[L_AVL + slot] / g);
    }
    if (slot == 7) op[7 + 2 * nu] = (float)se[L_R1 + 2 * 13];
    if (slot == 8) op[8 + 2 * nu] = (float)se[L_R1 + 2 * 13 + 1];
    if (slot >= 8) op[9 + 2 * nu + slot - 8] = (float)goal[slot - 8];
  }
  __syncthreads();
  if (env_ok) {
    float* op = s_pre[el];
    float* oq = s_post[el];
    for (int c = slot; c < n_obs; c += SLOTS) {
      float v = op[c];
      if (need_reset) {   // get_obs of the freshly reset task: goal steps zero, clock of the drawn phase
        if (c == 7 + 2 * nu) v = (float)se[L_R1 + 2 * 15];
        else if (c == 8 + 2 * nu) v = (float)se[L_R1 + 2 * 15 + 1];
        else if (c >= 9 + 2 * nu) v = 0.0f;
      }
      oq[c] = v;
    }

    // ------------------------------------------------------------ rewards, flags, task state
    if (!reset_all && slot == 0) {
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
        if (p.ro.buf_rew6) p.ro.buf_rew6[(tN + n) * 6 + i] = (float)rew[i];
      }
      p.ro.buf_rewards[tN + n] = tot;
      p.ro.buf_flags[tN + n] = (uint8_t)((cut ? OLY_FLAG_LAST : 0) | (done ? OLY_FLAG_ABSORBING : 0));
      p.ro.traj_len[n] = cut ? 0 : len;
    }
    // bootstrap row: finish_path's last_val = (not done) * V(state) needs V of THIS observation
    if (!reset_all && cut && !done) {
      const int sc = p.ro.side_count[n];
      if (sc < p.ro.side_slots) {
        const size_t srow = (size_t)n * p.ro.side_slots + sc;
        for (int c = slot; c < n_obs; c += SLOTS) p.ro.side_obs[srow * n_obs + c] = s_pre[el][c];
        if (slot == 0) p.ro.side_t[srow] = t;
      }
    }
  }
  __syncthreads();   // side_count is read above by all 16 lanes and bumped below by one
  if (env_ok) {
    if (!reset_all && cut && !done && slot == 0) p.ro.side_count[n] += 1;
    if (need_reset) {
      // WalkingTask.reset (walking_task.py:321-397) + transform_sequence (:113-135)
      const double cyw = se[L_R2 + 2 * 5 + 1], syw = se[L_R2 + 2 * 5];
      const double mid0 = (lf0 + rf0) / 2, mid1 = (lf1 + rf1) / 2;
      double* seq_out = const_cast<double*>(p.st.sequence) + (size_t)n * OLY_MAX_SEQ * 4;
#pragma unroll
      for (int q = 0; q < (OLY_MAX_SEQ + SLOTS - 1) / SLOTS; ++q) {
        const int r = slot + SLOTS * q;
        if (r >= OLY_MAX_SEQ) continue;
        double o0 = 0.0, o1 = 0.0, o2 = 0.0, o3 = 0.0;
        if (r < new_len) {
          const double x = rec_seq[q][0], y = rec_seq[q][1], z = rec_seq[q][2], th = rec_seq[q][3];
          o0 = mid0 + x * cyw - y * syw;
          o1 = mid1 + x * syw + y * cyw;
          o2 = z;
          o3 = root_yaw + th;
        }
        seq_out[4 * r] = o0; seq_out[4 * r + 1] = o1; seq_out[4 * r + 2] = o2; seq_out[4 * r + 3] = o3;
      }
      if (slot == 0) {
        p.st.phase[n] = new_phase;
        p.st.t1[n] = 0;
        p.st.t2[n] = (new_len == 1) ? 0 : 1;        // t1 = t2 = 0, then update_target_steps
        p.st.reached_frames[n] = 0;
        p.st.target_reached[n] = 0;
        const_cast<int32_t*>(p.st.mode)[n] = new_mode;
        const_cast<int32_t*>(p.st.seq_len)[n] = new_len;
        p.ro.pool_count[n] = rc + 1;
      }
      if (slot < 8) p.st.goal[8 * (size_t)n + slot] = 0.0;
    } else if (!reset_all) {
      if (slot == 0) {
        p.st.phase[n] = phase;
        p.st.t1[n] = t1;
        p.st.t2[n] = t2;
        p.st.reached_frames[n] = frames;
        p.st.target_reached[n] = (uint8_t)reached;
      }
      if (slot < 8) p.st.goal[8 * (size_t)n + slot] = goal[slot];
    }
  }
  // memory.store(state, action, value) and the PD target (ppo.py:186, robot.py:88-95)
  if (!reset_all) {
#pragma unroll
    for (int q = 0; q < ACT_PT; ++q) {
      const int e = tid + q * THREADS;
      if (e < rows * nu) {
        const int r = e / nu, j = e - r * nu;
        const size_t nn = (size_t)(row0 + r);
        p.ro.buf_actions[(tN + nn) * nu + j] = pt_act[q];
        if (p.ro.buf_mu) p.ro.buf_mu[(tN + nn) * nu + j] = pt_mu[q];
        p.ro.pd_target[nn * nu + j] = (double)pt_act[q] + m->motor_offset[j];
      }
    }
#pragma unroll
    for (int q = 0; q < OBS_PT; ++q) {
      const int e = tid + q * THREADS;
      if (e < rows * n_obs) p.ro.buf_states[(tN + row0) * n_obs + e] = pt_obs[q];
    }
    if (tid < rows) p.ro.buf_values[tN + row0 + tid] = pt_val;
  }
  // next observation for the policy: dense [rows, n_obs] stream
  for (int e = tid; e < rows * n_obs; e += THREADS) {
    const int r = e / n_obs, c = e - r * n_obs;
    p.ro.state[(size_t)row0 * n_obs + e] = s_post[r][c];
  }
  if (tid == 0 && !reset_all) {
    p.ro.ctr[2 * blockIdx.x] = t + 1;
    p.ro.ctr[2 * blockIdx.x + 1] = k + 1;
  }
}
}  // namespace

// (t, k) per 16-environment workgroup, then the sticky overrun word (+ one pad word)
extern "C" int oly_a3_vec_ctr_len(int N) { return N <= 0 ? 0 : 2 * ((N + EPW - 1) / EPW) + 2; }

extern "C" int oly_a3_vec_step(oly_ctx* ctx, int N, const oly_a3_blocks* blocks, const oly_a3_state* st,
                               const oly_a3_rollout* ro, int flags, oly_stream stream) {
  if (!ctx) return OLY_EINVAL;
  if (!ctx->a3_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_a3_vec_step before oly_a3_configure");
  if (!ctx->contact_ok) OLY_FAIL(ctx, OLY_ENOTCONF, "oly_a3_vec_step before oly_contact_configure");
  if (N < 0 || !blocks || !st || !ro) OLY_FAIL(ctx, OLY_EINVAL, "oly_a3_vec_step: bad argument");
  if (N == 0) return OLY_OK;
  if (blocks->K <= 0 || blocks->C <= 0 || ro->T <= 0 || ro->pool_depth <= 0 || ro->side_slots < 0)
    OLY_FAIL(ctx, OLY_EINVAL, "oly_a3_vec_step: bad sizes (K=%d C=%d T=%d pool_depth=%d)", blocks->K, blocks->C,
             ro->T, ro->pool_depth);
  const void* req[] = {blocks->qpos, blocks->qvel, blocks->act_len, blocks->act_vel, blocks->lf_pos, blocks->rf_pos,
                       blocks->lf_vel, blocks->rf_vel, blocks->root_pos, blocks->root_quat, blocks->head_pos,
                       blocks->ncon, blocks->geom1, blocks->geom2, blocks->force6, blocks->cpos_z,
                       st->phase, st->t1, st->t2, st->reached_frames, st->target_reached, st->mode, st->seq_len,
                       st->sequence, st->goal, ro->state, ro->pool, ro->pool_count, ro->ctr, ro->traj_len};
  for (const void* q : req)
    if (!q) OLY_FAIL(ctx, OLY_EINVAL, "oly_a3_vec_step: NULL pointer in blocks / state / rollout");
  if (!(flags & OLY_VSTEP_RESET_ALL)) {
    const void* req2[] = {ro->mu, ro->value, ro->pd_target, ro->buf_states, ro->buf_actions, ro->buf_rewards,
                          ro->buf_values, ro->buf_flags, ro->side_obs, ro->side_t, ro->side_count};
    for (const void* q : req2)
      if (!q) OLY_FAIL(ctx, OLY_EINVAL, "oly_a3_vec_step: NULL pointer in the rollout buffers");
    if (!ro->deterministic && (!ro->scale || !ro->eps))
      OLY_FAIL(ctx, OLY_EINVAL, "oly_a3_vec_step: stochastic step without scale / eps");
  }
  if (ctx->a3_host.nu > MAX_NU) OLY_FAIL(ctx, OLY_ERANGE, "oly_a3_vec_step: nu > %d", MAX_NU);
  VecArgs a;
  a.md = ctx->a3_dev;
  a.cd = ctx->contact;
  a.N = N;
  a.flags = flags;
  a.b = *blocks;
  a.st = *st;
  a.ro = *ro;
  hipLaunchKernelGGL(a3_vec_kernel, dim3((N + EPW - 1) / EPW), dim3(THREADS), 0, oly_s(stream), a);
  OLY_LAUNCH_CHECK(ctx, "a3_vec_kernel");
  return OLY_OK;
}
