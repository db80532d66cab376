// Device helpers shared by the one-launch-per-step vec kernel (K10, csrc/k10_vec_step.hip) and the persistent
// rollout kernel (K13, csrc/k13_rollout.hip): both evaluate WalkingTask / get_obs with these very functions on the
// same arguments (same libm entry points, -ffp-contract=off), which is what makes their results bit-identical.
#pragma once
#include "oly_common.h"

namespace oly_a3v {
constexpr double PI = 3.141592653589793;
constexpr double EPS = 2.220446049250313e-16;
enum { F_NONE = 0, F_SINCOS = 1, F_TAN = 2, F_EXP = 3, F_ATAN2 = 4 };

__device__ __forceinline__ void quat2mat(double w, double x, double y, double z, double R[3][3]) {
  const double nq = w * w + x * x + y * y + z * z;
  if (nq < EPS) {
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

__device__ __forceinline__ double vnorm3(double a0, double a1, double a2) {
  return sqrt(a0 * a0 + a1 * a1 + a2 * a2);
}

// one function on one argument per lane; lanes of different classes diverge, so a call costs one
// evaluation per class present in the wave
__device__ __forceinline__ void eval_task(int cls, double a, double b, double& r0, double& r1) {
  r0 = 0.0;
  r1 = 0.0;
  if (cls == F_SINCOS) {
    sincos(a, &r0, &r1);   // bit-identical to sin() / cos() on gfx950 (tools/hip/check_sincos.hip: 2^24 arguments)
  } else if (cls == F_TAN) {
    r0 = tan(a);
  } else if (cls == F_EXP) {
    r0 = exp(a);
  } else if (cls == F_ATAN2) {
    r0 = atan2(a, b);
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The environment step for 16 environments per 256-thread group, 16 lanes per environment (K13, and the 16-lane form
// of oly_a3_step): per-environment LDS scratch, and its libm-free "level 1" arithmetic as FOUR TASKS.
constexpr int A3V_SLOTS = 16;                 // lanes per environment
constexpr int A3V_EPW = 16;                   // environments per group
constexpr int A3V_SEQW = OLY_MAX_SEQ * 4;     // doubles of one environment's step sequence

// per-environment LDS scratch of the environment step (doubles)
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
  L_R1 = 61,     // round-1 results [16][2]
  L_R2 = 93,     // round-2 results [16][2]
  L_GR = 125,    // contact reduction: grf_r, grf_l, min_z
  L_GL = 126,
  L_MZ = 127,
  L_ROT = 128,   // root rotation matrix R[3][3] (level-1 task 1 -> round 2 / goal steps)
  L_ENV = 137
};
// per-environment ints in LDS: what the level-1 tasks read (written by the environment's own lanes) and what they
// hand back
enum {
  I_PHASE0 = 0, I_T1, I_T2, I_FRAMES, I_MODE, I_SEQLEN, I_TLEN, I_RC, I_BAD, I_HAVEC,           // inputs
  O_PHASE, O_T1, O_T2, O_FRAMES, O_REACHED, O_DONE, O_CUT, O_RESET, O_NEWMODE, O_NEWPHASE, O_NEWLEN,   // outputs
  SI_N = 24
};


struct Level1Ctx {
  const A3Dev* m;
  double* s_env;            // [A3V_EPW][L_ENV]
  const double* seqs;       // [A3V_EPW][A3V_SEQW]
  int* s_int;               // [A3V_EPW][SI_N]
  double* s_arg;            // [A3V_EPW][A3V_SLOTS][2]
  uint8_t* s_cls;           // [A3V_EPW][A3V_SLOTS]
  const double* s_lut;      // [4][period] clock LUT (LDS or the model's own)
  int period, rows;
  // rollout bookkeeping (ppo.py:178,189-196); a plain env.step passes last_step = true (no cut-driven reset)
  bool last_step;
  int max_traj_len;
  const oly_a3_reset_record* pool;
  int pool_depth;
  long row0;
};

// An environment's 16 lanes used to run the whole level-1 block redundantly: ~1100 instructions issued by every wave
// (3.3 us of a step).  Issue time follows the LENGTH of a wave's instruction stream, not its active lanes, so the
// block is cut by FUNCTION across the group's four waves (as the libm rounds are), and inside a wave its independent
// long operations (fp64 square roots and divisions, ~30 dependent instructions each) go to the wave's four 16-lane
// groups, lane group s taking piece s of all 16 environments: the longest dependent chain of a wave is two such
// operations instead of five or six.  Same expressions on the same inputs as K10 / K2: bit-identical.  Each piece
// writes the round-1 libm arguments of the slots it owns (s_arg / s_cls) and its part of the per-environment
// results (s_int outputs, the root rotation at L_ROT).  Call from all four waves (every lane) between two workgroup
// barriers; inputs: the staged rows of s_env, seqs, the I_* ints and L_GR / L_GL / L_MZ.
__device__ __forceinline__ void level1_tasks(const Level1Ctx& c, int wave, int lane) {
  const int e = lane & (A3V_EPW - 1), sub = lane >> 4;
  const double* ee_ = c.s_env + e * L_ENV;
  const double* esq = c.seqs + e * A3V_SEQW;
  int* si = c.s_int + e * SI_N;
  double* arg = c.s_arg + e * A3V_SLOTS * 2;
  uint8_t* acl = c.s_cls + e * A3V_SLOTS;
  auto put = [&](int task, int cls, double a, double b) {
    arg[2 * task] = a;
    arg[2 * task + 1] = b;
    acl[task] = (uint8_t)cls;
  };
  const bool live = e < c.rows;
  // shared, cheap: done / cut / need_reset (walking_task.py:298-319, ppo.py:178,189-196)
  const double rp2 = ee_[L_RP + 2];
  const double foot_z = fmin(ee_[L_LF + 2], ee_[L_RF + 2]);
  const bool bad_e = si[I_BAD] != 0;
  const bool done = ((rp2 - foot_z) < 0.6) || bad_e;
  const int len = si[I_TLEN] + 1;
  const bool cut = done || len >= c.max_traj_len || c.last_step;
  const bool need_reset = live && cut && !c.last_step;
  const int mode_e = si[I_MODE];
  const bool walking = mode_e != OLY_MODE_STANDING;
  int phase = si[I_PHASE0] + 1;
  if (phase >= c.period) phase = 0;
  if (!live) {     // rows past N: every slot of the round-1 table says "nothing to evaluate" (lane group 0 writes them)
    if (sub == 0) {
      if (wave == 0) { put(0, F_NONE, 0, 0); put(1, F_NONE, 0, 0); put(6, F_NONE, 0, 0); put(8, F_NONE, 0, 0); put(9, F_NONE, 0, 0); }
      else if (wave == 1) { put(13, F_NONE, 0, 0); put(14, F_NONE, 0, 0); put(15, F_NONE, 0, 0); }
      else if (wave == 2) { put(7, F_NONE, 0, 0); put(10, F_NONE, 0, 0); put(11, F_NONE, 0, 0); put(12, F_NONE, 0, 0); }
      else { put(2, F_NONE, 0, 0); put(3, F_NONE, 0, 0); put(4, F_NONE, 0, 0); put(5, F_NONE, 0, 0); }
    }
    return;
  }
  if (wave == 0) {
    // WalkingTask.step (walking_task.py:246-293): target reached / delay / update_target_steps; then the
    // arguments that read the selected sequence rows.  Lane groups 0 / 1: the left / right foot's distances.
    const double rp0 = ee_[L_RP], rp1 = ee_[L_RP + 1];
    const int fo = (sub & 1) ? L_RF : L_LF;
    const double f0 = ee_[fo], f1 = ee_[fo + 1], f2 = ee_[fo + 2];
    int t1e = si[I_T1], t2e = si[I_T2], fr = si[I_FRAMES];
    const int seq_len_e = si[I_SEQLEN];
    const double tx = esq[4 * t1e], ty = esq[4 * t1e + 1], tz = esq[4 * t1e + 2];
    const double d_own = vnorm3(f0 - tx, f1 - ty, f2 - tz);            // groups 0, 2: left; 1, 3: right
    const double d_oth = __shfl_xor(d_own, 16);
    const double dl = (sub & 1) ? d_oth : d_own, dr = (sub & 1) ? d_own : d_oth;
    int reached;
    if (dl < c.m->target_radius || dr < c.m->target_radius) {
      reached = 1;
      fr += 1;
    } else {
      reached = 0;
      fr = 0;
    }
    if (reached && fr >= c.m->delay_frames) {  // update_target_steps
      t1e = t2e;
      t2e += 1;
      if (t2e == seq_len_e) t2e = seq_len_e - 1;
      t2e = min(max(t2e, 0), OLY_MAX_SEQ - 1);
      reached = 0;
      fr = 0;
    }
    const int selA = 4 * t1e, selB = 4 * t2e;   // sequence[t1] / sequence[t2] after the update
    const double s1x = esq[selA], s1y = esq[selA + 1], s1z = esq[selA + 2], s1w = esq[selA + 3];
    const double s2x = esq[selB], s2y = esq[selB + 1], s2w = esq[selB + 3];
    const double n_own = vnorm3(f0 - s1x, f1 - s1y, f2 - s1z);
    const double n_oth = __shfl_xor(n_own, 16);
    if (sub == 0) {
      const double fd = fmin(n_own, n_oth);                                        // fmin(left, right)
      put(8, F_EXP, -fd / 0.25, 0.0);                                              // target hit
      si[O_PHASE] = phase; si[O_T1] = t1e; si[O_T2] = t2e; si[O_FRAMES] = fr; si[O_REACHED] = reached;
      si[O_DONE] = done; si[O_CUT] = cut; si[O_RESET] = need_reset;
    } else if (sub == 1) {
      put(0, walking ? F_SINCOS : F_NONE, walking ? s1w : 0.0, 0.0);               // goal yaw 1: cos / sin(theta)
      put(1, walking ? F_SINCOS : F_NONE, walking ? s2w : 0.0, 0.0);
      put(6, F_SINCOS, s1w / 2.0, 0.0);                                            // euler2quat(0,0,yaw) of the target
    } else if (sub == 2) {
      const double mpx = (s1x + s2x) / 2, mpy = (s1y + s2y) / 2;
      const double rx = rp0 - mpx, ry = rp1 - mpy;
      put(9, F_EXP, -sqrt(rx * rx + ry * ry) / 2, 0.0);                            // progress
    }
  } else if (wave == 1) {
    if (sub == 0) {
      // root rotation (goal steps, round 2) and its yaw for transform_sequence: quat2euler(root xquat)[2]
      double R[3][3];
      quat2mat(ee_[L_RQ], ee_[L_RQ + 1], ee_[L_RQ + 2], ee_[L_RQ + 3], R);
      double* ro_ = c.s_env + e * L_ENV + L_ROT;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) ro_[3 * i + j] = R[i][j];
      const double cyr = sqrt(R[0][0] * R[0][0] + R[1][0] * R[1][0]);
      const bool yaw = need_reset && cyr > 4.0 * EPS;
      put(14, yaw ? F_ATAN2 : F_NONE, yaw ? R[1][0] : 0.0, yaw ? R[0][0] : 0.0);
    } else if (sub == 1) {
      // the clock observation, and env.reset()'s record header: mode / phase / length of the next pool record
      put(13, F_SINCOS, 2 * PI * phase / (double)c.period, 0.0);                     // clock
      int new_mode = mode_e, new_phase = 0, new_len = si[I_SEQLEN];
      if (need_reset) {
        const oly_a3_reset_record* rec = c.pool + (size_t)(c.row0 + e) * c.pool_depth +
                                         (unsigned)si[I_RC] % (unsigned)c.pool_depth;
        new_mode = rec->mode;
        new_phase = rec->phase;
        new_len = min(max(rec->seq_len, 1), OLY_MAX_SEQ);
      }
      put(15, need_reset ? F_SINCOS : F_NONE, need_reset ? 2 * PI * new_phase / (double)c.period : 0.0, 0.0);   // clock after reset
      si[O_NEWMODE] = new_mode; si[O_NEWPHASE] = new_phase; si[O_NEWLEN] = new_len;
    }
  } else if (wave == 2) {
    if (sub == 0) {
      // get_obs: quat2euler(qpos[3:7]) (StickFigureA3.py:160)
      double Rb[3][3];
      quat2mat(ee_[L_BQ], ee_[L_BQ + 1], ee_[L_BQ + 2], ee_[L_BQ + 3], Rb);
      const double cyb = sqrt(Rb[0][0] * Rb[0][0] + Rb[1][0] * Rb[1][0]);
      const bool regular = cyb > 4.0 * EPS;
      const double roll_y = regular ? Rb[2][1] : -Rb[1][2];
      const double roll_x = regular ? Rb[2][2] : Rb[1][1];
      put(11, F_ATAN2, roll_y, roll_x);                                            // roll
      put(12, F_ATAN2, -Rb[2][0], cyb);                                            // pitch
    } else if (sub == 1) {
      const double hx = ee_[L_HP] - ee_[L_RP], hy = ee_[L_HP + 1] - ee_[L_RP + 1];
      const double hn = sqrt(hx * hx + hy * hy);
      put(10, F_EXP, -10 * (hn * hn), 0.0);                                        // upper body
    } else if (sub == 2) {
      const double contact_point = si[I_HAVEC] ? ee_[L_MZ] : 0.0;
      double err = fabs((rp2 - contact_point) - c.m->goal_height_ref);
      const double deadzone = 0.01 + 0.05 * c.m->goal_speed_ref;
      if (err < deadzone) err = 0;
      put(7, F_EXP, -40 * (err * err), 0.0);                                       // height
    }
  } else {
    // calc_reward clock terms (walking_task.py:74-110, tasks/rewards.py:65-102): lane group 0 / 1 the left / right
    // foot's force term, 2 / 3 the left / right foot's velocity term
    const bool right = sub & 1, velocity = sub >= 2;
    const int lut_row = velocity ? (right ? 1 : 3) : (right ? 0 : 2);          // r_frc 0, r_vel 1, l_frc 2, l_vel 3
    double coef;
    if (!walking) coef = velocity ? -1.0 : 1.0;
    else coef = c.s_lut[lut_row * c.period + phase];
    double x;
    if (!velocity) {
      const double max_frc = c.m->mass * 9.8 * 0.5;
      x = fmin(ee_[right ? L_GR : L_GL], max_frc) / max_frc;
    } else {
      const int vo = right ? L_RV : L_LV;
      x = fmin(vnorm3(ee_[vo], ee_[vo + 1], ee_[vo + 2]), 0.2) / 0.2;
    }
    x *= 2; x -= 1;
    // slots: 2 left force, 3 right force, 4 left velocity, 5 right velocity
    put(2 + sub, F_TAN, PI / 4 * coef * x, 0.0);
  }
}
}  // namespace oly_a3v
