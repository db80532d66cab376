/*
 * oly_oracle.c - CPU restatement of the reference's per-environment hot path.
 * TEST INFRASTRUCTURE (see oly_oracle.h): the checker for the HIP kernels and the
 * "port" CPU baseline of bench.py.  Never linked into or loaded by the product.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no fused multiply-add, so that
 * every fp64 expression rounds exactly as numpy evaluates it in the reference).
 * Citations are file:line under /root/reference.
 */
#include "oly_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI 3.141592653589793 /* numpy.pi */

/* ===================================================================== K1 + K5 (IL) */

/* One (step, env) row.  `prev` is the reward input carried by self._obs. */
static void il_row(const oly_il_model* m, const double* qpos, const double* qvel,
                   const float* action, const double* grf, double* prev, void* obs_out,
                   float* reward, uint8_t* absorbing, uint8_t* fall_code, void* ctrl_out,
                   int out_flags, double* reward_f64) {
  double full[OLY_MAX_OBS + 16];
  double obs[OLY_MAX_OBS + 16];
  int n_spec = m->n_pos + m->n_vel;
  /* mushroom ObservationHelper._build_obs driven by the spec, UnitreeH1.py:303-355:
     data.joint(name).qpos for JOINT_POS entries, .qvel for JOINT_VEL, in spec order.  */
  for (int i = 0; i < m->n_pos; ++i) full[i] = qpos[m->qpos_adr[i]];
  for (int i = 0; i < m->n_vel; ++i) full[m->n_pos + i] = qvel[m->qvel_adr[i]];
  /* LocoEnvBase._create_observation, loco_env_base.py:737-767: obs[2:], (+ mean_grf/1000) */
  int n_obs = 0;
  for (int i = m->n_drop; i < n_spec; ++i) obs[n_obs++] = full[i];
  for (int i = 0; i < m->n_grf; ++i) obs[n_obs++] = grf[i] / 1000.0;

  /* BaseHumanoidRobot.is_absorbing -> UnitreeH1._has_fallen, base_humanoid_robot.py:246-260,
     UnitreeH1.py:176-198: strict comparisons on the float64 values; first violated test.  */
  int code = 0;
  for (int k = 0; k < m->n_fall && !code; ++k) {
    double v = obs[m->fall_idx[k]];
    if (v < m->fall_lo[k] || v > m->fall_hi[k]) code = k + 1;
  }
  *absorbing = (uint8_t)((code != 0) && m->use_absorbing_states);
  if (fall_code) *fall_code = (uint8_t)code;

  /* LocoEnvBase.reward(state=self._obs, ...), loco_env_base.py:776-781 -> the reward object
     reads `state`, the PREVIOUS observation (utils/reward.py:41,50,73-74).                */
  double r = 0.0;
  if (m->reward_type == OLY_REWARD_TARGET_VELOCITY) {
    double d = *prev - m->target_velocity;
    r = exp(-(d * d));
  } else if (m->reward_type == OLY_REWARD_X_POS) {
    r = *prev;
  }
  *reward = (float)r;
  if (reward_f64) *reward_f64 = r;
  if (m->reward_type != OLY_REWARD_NONE) *prev = obs[m->reward_idx]; /* self._obs = cur_obs */

  if (out_flags & OLY_OUT_OBS_F64) {
    memcpy(obs_out, obs, sizeof(double) * (size_t)n_obs);
  } else {
    float* o = (float*)obs_out;
    for (int i = 0; i < n_obs; ++i) o[i] = (float)obs[i];
  }

  /* LocoEnvBase._preprocess_action, loco_env_base.py:1066: a*delta + mean in float64;
     mushroom scatters into data.ctrl[action_indices]; MuJoCo clamps to ctrlrange.       */
  if (ctrl_out) {
    double c[OLY_MAX_ACT];
    for (int j = 0; j < m->nu; ++j) c[j] = 0.0;
    for (int k = 0; k < m->n_act; ++k) {
      double u = (double)action[k] * m->act_delta[k] + m->act_mean[k];
      if (u < m->ctrl_lo[k]) u = m->ctrl_lo[k];
      if (u > m->ctrl_hi[k]) u = m->ctrl_hi[k];
      c[m->act_to_ctrl[k]] = u;
    }
    if (out_flags & OLY_OUT_CTRL_F64) {
      memcpy(ctrl_out, c, sizeof(double) * (size_t)m->nu);
    } else {
      float* o = (float*)ctrl_out;
      for (int j = 0; j < m->nu; ++j) o[j] = (float)c[j];
    }
  }
}

static int il_check(const oly_il_model* m) {
  if (!m || m->n_pos + m->n_vel > OLY_MAX_OBS || m->n_act > OLY_MAX_ACT || m->nu > OLY_MAX_ACT ||
      m->n_fall > OLY_MAX_FALL)
    return OLY_EINVAL;
  return OLY_OK;
}

int oly_il_step_cpu(const oly_il_model* m, int T, int N, const double* qpos, const double* qvel,
                    const float* action, const double* grf_mean, double* prev_inout, void* obs,
                    float* reward, uint8_t* absorbing, uint8_t* fall_code, void* ctrl,
                    int out_flags, double* reward_f64) {
  if (il_check(m) || T < 0 || N < 0 || !qpos || !qvel || !prev_inout || !obs || !reward || !absorbing)
    return OLY_EINVAL;
  int n_obs = m->n_pos + m->n_vel - m->n_drop + m->n_grf;
  size_t osz = (out_flags & OLY_OUT_OBS_F64) ? 8 : 4, csz = (out_flags & OLY_OUT_CTRL_F64) ? 8 : 4;
  for (int t = 0; t < T; ++t)
    for (int n = 0; n < N; ++n) {
      size_t r = (size_t)t * N + n;
      il_row(m, qpos + r * m->nq, qvel + r * m->nv, action ? action + r * m->n_act : NULL,
             grf_mean ? grf_mean + r * m->n_grf : NULL, prev_inout + n,
             (char*)obs + r * n_obs * osz, reward + r, absorbing + r,
             fall_code ? fall_code + r : NULL, ctrl ? (char*)ctrl + r * m->nu * csz : NULL,
             out_flags, reward_f64 ? reward_f64 + r : NULL);
    }
  return OLY_OK;
}

int oly_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int oly_il_step_cpu_mt(const oly_il_model* m, int T, int N, const double* qpos,
                       const double* qvel, const float* action, double* prev_inout, void* obs,
                       float* reward, uint8_t* absorbing, void* ctrl, int out_flags, int threads) {
  if (il_check(m)) return OLY_EINVAL;
  int n_obs = m->n_pos + m->n_vel - m->n_drop + m->n_grf;
  size_t osz = (out_flags & OLY_OUT_OBS_F64) ? 8 : 4, csz = (out_flags & OLY_OUT_CTRL_F64) ? 8 : 4;
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
  for (int n = 0; n < N; ++n)
    for (int t = 0; t < T; ++t) {
      size_t r = (size_t)t * N + n;
      il_row(m, qpos + r * m->nq, qvel + r * m->nv, action ? action + r * m->n_act : NULL, NULL,
             prev_inout + n, (char*)obs + r * n_obs * osz, reward + r, absorbing + r, NULL,
             ctrl ? (char*)ctrl + r * m->nu * csz : NULL, out_flags, NULL);
    }
  (void)threads;
  return OLY_OK;
}

/* ============================================================================== K4 */

#define TAB(k, j, s) table[((size_t)(k) * n_traj + (j)) * len + (s)]

/* Trajectory.reset_trajectory, utils/trajectory.py:289-323: the chosen sub-trajectory is
   COPIED (_get_subtraj :458-464) and keys 0,1 are shifted by their value at `step`.    */
int oly_traj_reset_cpu(int n_keys, int n_traj, int len, const double* table, int N,
                       const int32_t* traj_no, const int32_t* step, int32_t* cur_traj,
                       int32_t* cur_step, double* origin, double* sample) {
  for (int n = 0; n < N; ++n) {
    int j = traj_no[n], s = step[n];
    if (j < 0 || j >= n_traj || s < 0 || s >= len) return OLY_ERANGE;
    cur_traj[n] = j;
    cur_step[n] = s;
    double x0 = TAB(0, j, s), y0 = TAB(1, j, s);
    origin[2 * n] = x0;
    origin[2 * n + 1] = y0;
    for (int k = 0; k < n_keys; ++k) {
      double v = TAB(k, j, s);
      if (k == 0) v -= x0; /* self.subtraj[0] -= self.subtraj[0][step]  :319 */
      if (k == 1) v -= y0; /*                                           :320 */
      sample[(size_t)n * n_keys + k] = v;
    }
  }
  return OLY_OK;
}

/* Trajectory.get_next_sample, utils/trajectory.py:389-401. */
int oly_traj_next_cpu(int n_keys, int n_traj, int len, const double* table, int N,
                      const uint8_t* active, const int32_t* cur_traj, int32_t* cur_step,
                      const double* origin, double* sample, uint8_t* at_end) {
  for (int n = 0; n < N; ++n) {
    if (active && !active[n]) {
      at_end[n] = 0;
      continue;
    }
    int j = cur_traj[n];
    int s = cur_step[n];
    if (s < len) s += 1; /* self.subtraj_step_no += 1; saturates at len: the reference's
                            callers reset right after a None (loco_env_base.py:534-537), so a
                            call past the end never happens there (it would raise IndexError) */
    cur_step[n] = s;
    if (s >= len) { /* sample = None */
      at_end[n] = 1;
      continue;
    }
    at_end[n] = 0;
    for (int k = 0; k < n_keys; ++k) {
      double v = TAB(k, j, s);
      if (k == 0) v -= origin[2 * n];
      if (k == 1) v -= origin[2 * n + 1];
      sample[(size_t)n * n_keys + k] = v;
    }
  }
  return OLY_OK;
}

/* play_trajectory_from_velocity, loco_env_base.py:515-519:
   qpos = [qp + self.dt * qv for qp, qv in zip(curr_qpos, qvel)]; sample[:len(qpos)] = qpos */
int oly_traj_euler_cpu(int n_keys, int N, int n_qpos, double dt, const double* curr_qpos,
                       double* sample) {
  for (int n = 0; n < N; ++n)
    for (int j = 0; j < n_qpos; ++j) {
      double* s = sample + (size_t)n * n_keys;
      s[j] = curr_qpos[(size_t)n * n_qpos + j] + dt * s[n_qpos + j];
    }
  return OLY_OK;
}

/* ============================================================================== K3 */

/* MujocoRobotInterface.get_{r,l}foot_floor_contacts :245-273 (floor must be geom1),
   get_{r,l}foot_grf :275-297 (np.linalg.norm of the 6-vector, summed in contact order),
   check_bad_collisions :393-399, and the contact point of _calc_height_reward
   (tasks/rewards.py:29-33).                                                            */
int oly_contact_reduce_cpu(int ngeom, const int32_t* geom_bodyid, int floor_body, int rfoot_body,
                           int lfoot_body, int N, int C, const int32_t* ncon,
                           const int32_t* geom1, const int32_t* geom2, const double* force6,
                           const double* pos_z, int32_t* n_r, int32_t* n_l, int32_t* idx_r,
                           int32_t* idx_l, double* grf_r, double* grf_l, double* min_z,
                           uint8_t* bad) {
  for (int n = 0; n < N; ++n) {
    int nc = ncon[n];
    if (nc < 0 || nc > C) return OLY_ERANGE;
    int nr = 0, nl = 0;
    double gr = 0.0, gl = 0.0, mz = 0.0;
    int have = 0;
    /* the reference builds the right list fully, then the left list; min() runs over
       rcontacts + lcontacts.  Sums and the minimum are order-independent across the two
       lists except for fp addition order WITHIN a list, which is contact order.        */
    for (int i = 0; i < nc; ++i) {
      size_t e = (size_t)n * C + i;
      int g1 = geom1[e], g2 = geom2[e];
      if (g1 < 0 || g1 >= ngeom || g2 < 0 || g2 >= ngeom) return OLY_ERANGE;
      int b1 = geom_bodyid[g1], b2 = geom_bodyid[g2];
      int is_r = (b1 == floor_body) && (b2 == rfoot_body);
      int is_l = (b1 == floor_body) && (b2 == lfoot_body);
      if (is_r || is_l) {
        const double* f = force6 + e * 6;
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s += f[k] * f[k];
        double nrm = sqrt(s);
        if (!have || pos_z[e] < mz) mz = pos_z[e];
        have = 1;
        if (is_r) {
          if (idx_r) idx_r[(size_t)n * C + nr] = i;
          gr += nrm;
          ++nr;
        }
        if (is_l) { /* both can hold only if rfoot_body == lfoot_body */
          if (idx_l) idx_l[(size_t)n * C + nl] = i;
          gl += nrm;
          ++nl;
        }
      }
    }
    if (idx_r) for (int i = nr; i < C; ++i) idx_r[(size_t)n * C + i] = -1;
    if (idx_l) for (int i = nl; i < C; ++i) idx_l[(size_t)n * C + i] = -1;
    n_r[n] = nr;
    n_l[n] = nl;
    grf_r[n] = gr;
    grf_l[n] = gl;
    min_z[n] = have ? mz : 0.0;
    bad[n] = (uint8_t)((nr + nl) != nc);
  }
  return OLY_OK;
}

/* ========================================================================== K2 (A3) */

/* transforms3d.quaternions.quat2mat (w,x,y,z) - restated from the package docs. */
static void quat2mat(const double* q, double R[3][3]) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double nq = w * w + x * x + y * y + z * z;
  if (nq < 2.220446049250313e-16) {
    memset(R, 0, sizeof(double) * 9);
    R[0][0] = R[1][1] = R[2][2] = 1.0;
    return;
  }
  double s = 2.0 / nq;
  double X = x * s, Y = y * s, Z = z * s;
  double wX = w * X, wY = w * Y, wZ = w * Z;
  double xX = x * X, xY = x * Y, xZ = x * Z;
  double yY = y * Y, yZ = y * Z, zZ = z * Z;
  R[0][0] = 1.0 - (yY + zZ); R[0][1] = xY - wZ;         R[0][2] = xZ + wY;
  R[1][0] = xY + wZ;         R[1][1] = 1.0 - (xX + zZ); R[1][2] = yZ - wX;
  R[2][0] = xZ - wY;         R[2][1] = yZ + wX;         R[2][2] = 1.0 - (xX + yY);
}

static double norm3(const double* a, const double* b) {
  double d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
  return sqrt(d0 * d0 + d1 * d1 + d2 * d2);
}

static void a3_row(const oly_a3_model* m, int n, const oly_a3_inputs* in, const oly_a3_state* st,
                   double* obs, double* rew, double* total, uint8_t* done) {
  const double* lf = in->lf_pos + 3 * n;
  const double* rf = in->rf_pos + 3 * n;
  const double* seq = st->sequence + (size_t)n * OLY_MAX_SEQ * 4;
  int mode = st->mode[n];

  /* ---- WalkingTask.step, walking_task.py:246-293 */
  int phase = st->phase[n] + 1;
  if (phase >= m->period) phase = 0;
  int t1 = st->t1[n], t2 = st->t2[n];
  const double* target = seq + 4 * t1;
  double dl = norm3(lf, target), dr = norm3(rf, target);
  int reached;
  int frames = st->reached_frames[n];
  if (dl < m->target_radius || dr < m->target_radius) {
    reached = 1;
    frames += 1;
  } else {
    reached = 0;
    frames = 0;
  }
  if (reached && frames >= m->delay_frames) {
    /* update_target_steps :228-244 */
    t1 = t2;
    t2 += 1;
    if (t2 == st->seq_len[n]) t2 = st->seq_len[n] - 1;
    reached = 0;
    frames = 0;
  }
  st->phase[n] = phase;
  st->t1[n] = t1;
  st->t2[n] = t2;
  st->reached_frames[n] = frames;
  st->target_reached[n] = (uint8_t)reached;

  /* ---- update_goal_steps :184-225: relative = inv(T_root) . T_target.  T_root is rigid
     ([R p; 0 1], R from quat2mat), so inv = [R^T, -R^T p]; goal xyz = R^T (target - p),
     theta = mat2euler(R^T Rz(yaw))[2] = atan2(M[1][0], M[0][0]).                        */
  double* goal = st->goal + 8 * n;
  for (int i = 0; i < 8; ++i) goal[i] = 0.0;
  const double* rp = in->root_pos + 3 * n;
  const double* rq = in->root_quat + 4 * n;
  if (mode != OLY_MODE_STANDING) {
    double R[3][3];
    quat2mat(rq, R);
    int tt[2] = {t1, t2};
    for (int i = 0; i < 2; ++i) {
      const double* s = seq + 4 * tt[i];
      double d[3] = {s[0] - rp[0], s[1] - rp[1], s[2] - rp[2]};
      goal[0 + i] = R[0][0] * d[0] + R[1][0] * d[1] + R[2][0] * d[2];
      goal[2 + i] = R[0][1] * d[0] + R[1][1] * d[1] + R[2][1] * d[2];
      goal[4 + i] = R[0][2] * d[0] + R[1][2] * d[1] + R[2][2] * d[2];
      double c = cos(s[3]), sn = sin(s[3]);
      /* M = R^T Rz: M[0][0] = R00 c + R10 s ; M[1][0] = R01 c + R11 s */
      double m00 = R[0][0] * c + R[1][0] * sn;
      double m10 = R[0][1] * c + R[1][1] * sn;
      goal[6 + i] = atan2(m10, m00);
    }
  }

  /* ---- WalkingTask.calc_reward :74-110 */
  double c_rfrc, c_rvel, c_lfrc, c_lvel;
  if (mode == OLY_MODE_STANDING) { /* :82-90 */
    c_rfrc = 1.0; c_lfrc = 1.0; c_rvel = -1.0; c_lvel = -1.0;
  } else {
    c_rfrc = m->clock_lut[0 * m->period + phase];
    c_rvel = m->clock_lut[1 * m->period + phase];
    c_lfrc = m->clock_lut[2 * m->period + phase];
    c_lvel = m->clock_lut[3 * m->period + phase];
  }
  /* _calc_foot_frc_clock_reward, tasks/rewards.py:65-83 */
  double max_frc = m->mass * 9.8 * 0.5;
  double nl = fmin(in->grf_l[n], max_frc) / max_frc;
  double nr = fmin(in->grf_r[n], max_frc) / max_frc;
  nl *= 2; nl -= 1; nr *= 2; nr -= 1;
  double frc = (tan(PI / 4 * c_lfrc * nl) + tan(PI / 4 * c_rfrc * nr)) / 2;
  /* _calc_foot_vel_clock_reward :85-102 */
  static const double zero3[3] = {0, 0, 0};
  double vl = fmin(norm3(in->lf_vel + 3 * n, zero3), 0.2) / 0.2;
  double vr = fmin(norm3(in->rf_vel + 3 * n, zero3), 0.2) / 0.2;
  vl *= 2; vl -= 1; vr *= 2; vr -= 1;
  double vel = (tan(PI / 4 * c_lvel * vl) + tan(PI / 4 * c_rvel * vr)) / 2;
  /* _calc_body_orient_reward :121-126 with quat_ref = euler2quat(0,0,yaw) = (cos(y/2),0,0,sin(y/2)) */
  double yaw = seq[4 * t1 + 3];
  double tq[4] = {cos(yaw / 2.0), 0.0, 0.0, sin(yaw / 2.0)};
  double ip = tq[0] * rq[0] + tq[1] * rq[1] + tq[2] * rq[2] + tq[3] * rq[3];
  double orient = exp(-(10 * (1 - ip * ip)));
  /* _calc_height_reward :27-40 */
  double contact_point = (in->n_r[n] > 0 || in->n_l[n] > 0) ? in->min_z[n] : 0.0;
  double err = fabs((rp[2] - contact_point) - m->goal_height_ref);
  double deadzone = 0.01 + 0.05 * m->goal_speed_ref;
  if (err < deadzone) err = 0;
  double height = exp(-40 * (err * err));
  /* step_reward :56-72 (uses t1/t2/target_reached as updated by step()) */
  const double* tp = seq + 4 * t1;
  double fd = fmin(norm3(lf, tp), norm3(rf, tp));
  double hit = reached ? exp(-fd / 0.25) : 0.0;
  double mpx = (seq[4 * t1] + seq[4 * t2]) / 2, mpy = (seq[4 * t1 + 1] + seq[4 * t2 + 1]) / 2;
  double rx = rp[0] - mpx, ry = rp[1] - mpy;
  double progress = exp(-sqrt(rx * rx + ry * ry) / 2);
  double step_r = 0.8 * hit + 0.2 * progress;
  /* upper body :108 */
  const double* hp = in->head_pos + 3 * n;
  double hx = hp[0] - rp[0], hy = hp[1] - rp[1];
  double hn = sqrt(hx * hx + hy * hy);
  double upper = exp(-10 * (hn * hn));
  rew[0] = 0.150 * frc;
  rew[1] = 0.150 * vel;
  rew[2] = 0.050 * orient;
  rew[3] = 0.050 * height;
  rew[4] = 0.450 * step_r;
  rew[5] = 0.050 * upper;
  double tot = 0.0; /* sum([float(i) for i in rewards.values()]) StickFigureA3.py:194 */
  for (int i = 0; i < 6; ++i) tot += rew[i];
  *total = tot;

  /* ---- WalkingTask.done :298-319 */
  double foot_z = fmin(lf[2], rf[2]);
  *done = (uint8_t)(((rp[2] - foot_z) < 0.6) || in->bad[n]);

  /* ---- StickFigureA3.get_obs, StickFigureA3.py:144-178 */
  const double* qpos = in->qpos + (size_t)n * m->nq;
  const double* qvel = in->qvel + (size_t)n * m->nv;
  /* quat2euler(qpos[3:7])[0:2] (sxyz) then euler2quat(roll, pitch, 0) */
  double Rb[3][3];
  quat2mat(qpos + 3, Rb);
  double cy = sqrt(Rb[0][0] * Rb[0][0] + Rb[1][0] * Rb[1][0]);
  double roll, pitch;
  if (cy > 4.0 * 2.220446049250313e-16) {
    roll = atan2(Rb[2][1], Rb[2][2]);
    pitch = atan2(-Rb[2][0], cy);
  } else {
    roll = atan2(-Rb[1][2], Rb[1][1]);
    pitch = atan2(-Rb[2][0], cy);
  }
  double ci = cos(roll / 2.0), si = sin(roll / 2.0), cj = cos(pitch / 2.0), sj = sin(pitch / 2.0);
  int o = 0;
  obs[o++] = ci * cj;  /* ck = 1, sk = 0 */
  obs[o++] = si * cj;
  obs[o++] = ci * sj;
  obs[o++] = -(si * sj);
  for (int i = 3; i < 6; ++i) obs[o++] = qvel[i];
  for (int i = 0; i < m->nu; ++i) obs[o++] = in->act_len[(size_t)n * m->nu + i] / m->gear[i];
  for (int i = 0; i < m->nu; ++i) obs[o++] = in->act_vel[(size_t)n * m->nu + i] / m->gear[i];
  double ang = 2 * PI * phase / (double)m->period;
  obs[o++] = sin(ang);
  obs[o++] = cos(ang);
  for (int i = 0; i < 8; ++i) obs[o++] = goal[i];
}

int oly_a3_step_cpu(const oly_a3_model* m, int N, const oly_a3_inputs* in, const oly_a3_state* st,
                    void* obs, float* rew6, float* reward, uint8_t* done, int out_flags,
                    double* rew6_f64, double* reward_f64) {
  if (!m || !in || !st || m->nu > 16) return OLY_EINVAL;
  int n_obs = 7 + 2 * m->nu + 10;
  for (int n = 0; n < N; ++n) {
    double o[64], r[6], tot;
    a3_row(m, n, in, st, o, r, &tot, done + n);
    if (out_flags & OLY_OUT_OBS_F64)
      memcpy((double*)obs + (size_t)n * n_obs, o, sizeof(double) * (size_t)n_obs);
    else
      for (int i = 0; i < n_obs; ++i) ((float*)obs)[(size_t)n * n_obs + i] = (float)o[i];
    for (int i = 0; i < 6; ++i) {
      rew6[6 * (size_t)n + i] = (float)r[i];
      if (rew6_f64) rew6_f64[6 * (size_t)n + i] = r[i];
    }
    reward[n] = (float)tot;
    if (reward_f64) reward_f64[n] = tot;
  }
  return OLY_OK;
}

/* robot.JVRC.step, environments/robot.py:88-95 (actuators = identity list 0..nu-1). */
int oly_a3_pd_target_cpu(const oly_a3_model* m, int N, const float* action, double* target) {
  for (int n = 0; n < N; ++n)
    for (int i = 0; i < m->nu; ++i)
      target[(size_t)n * m->nu + i] = (double)action[(size_t)n * m->nu + i] + m->motor_offset[i];
  return OLY_OK;
}

/* MujocoRobotInterface.step_pd :425-443 + robot.JVRC.do_simulation robot.py:109-115. */
int oly_a3_pd_torque_cpu(const oly_a3_model* m, int N, const double* kp, const double* kd,
                         const double* target, const double* act_len, const double* act_vel,
                         double* tau) {
  for (int n = 0; n < N; ++n)
    for (int i = 0; i < m->nu; ++i) {
      size_t e = (size_t)n * m->nu + i;
      double q = act_len[e] / m->gear[i], qd = act_vel[e] / m->gear[i];
      double perror = target[e] - q, verror = 0.0 - qd;
      tau[e] = (kp[i] * perror + kd[i] * verror) / m->gear[i];
    }
  return OLY_OK;
}

/* ====================================================================== K10 (vec step) */

/* One vec step of PPO.sample (rl/algos/ppo.py:169-196) for N StickFigureA3 environments, row by
   row, composed from the pieces above exactly as the reference's loop orders them:
     action = Normal(mu, std*anneal).sample() with the noise given          ppo.py:181
     memory.store(state, action, reward, value)                             ppo.py:186
     JVRC.step target                                                       robot.py:88-95
     contacts -> task.step / calc_reward / done / get_obs                   StickFigureA3.py:187-200
     traj_len / cut, finish_path's bootstrap row                            ppo.py:178,189-196
     env.reset() of a cut environment: WalkingTask.reset from a pre-drawn record
       (walking_task.py:321-397), transform_sequence (:113-135), get_obs with the goal steps
       zero and the clock of the drawn phase                                StickFigureA3.py:205-235
   ctr[0] = t, ctr[1] = k (advanced by one unless OLY_VSTEP_RESET_ALL).  Host pointers.       */
int oly_a3_vec_step_cpu(const oly_a3_model* m, int ngeom, const int32_t* geom_bodyid, int floor_body,
                        int rfoot_body, int lfoot_body, int N, const oly_a3_blocks* b,
                        const oly_a3_state* st, const oly_a3_rollout* ro, int flags) {
  if (!m || !b || !st || !ro || b->K <= 0 || ro->pool_depth <= 0 || m->nu > 16) return OLY_EINVAL;
  const int nu = m->nu, n_obs = 7 + 2 * nu + 10, C = b->C;
  const int reset_all = (flags & OLY_VSTEP_RESET_ALL) != 0;
  const int t = ro->ctr[0], k = ro->ctr[1];
  if (!reset_all && (t < 0 || t >= ro->T)) return OLY_ERANGE;   /* the device kernel: no write, sticky mark */
  const int kk = (int)((unsigned)k % (unsigned)b->K);
  const size_t tN = (size_t)t * N, kN = (size_t)kk * N;
  for (int n = 0; n < N; ++n) {
    if (!reset_all) {
      for (int j = 0; j < nu; ++j) {
        float mu = ro->mu[(size_t)n * nu + j], a = mu;
        if (!ro->deterministic) {
          float sc = ro->scale[j] * ro->eps[(tN + n) * nu + j];
          a = mu + sc;
        }
        ro->buf_actions[(tN + n) * nu + j] = a;
        if (ro->buf_mu) ro->buf_mu[(tN + n) * nu + j] = mu;
        ro->pd_target[(size_t)n * nu + j] = (double)a + m->motor_offset[j];
      }
      memcpy(ro->buf_states + (tN + n) * n_obs, ro->state + (size_t)n * n_obs, sizeof(float) * n_obs);
      ro->buf_values[tN + n] = ro->value[n];
    }
    /* contacts of readback row kk; a count beyond the staged slots is a bad collision (as the kernel) */
    int32_t nc_raw = b->ncon[kN + n], nc = nc_raw < 0 ? 0 : (nc_raw > C ? C : nc_raw);
    int32_t n_r, n_l;
    double grf_r, grf_l, min_z;
    uint8_t bad;
    int rc0 = oly_contact_reduce_cpu(ngeom, geom_bodyid, floor_body, rfoot_body, lfoot_body, 1, C, &nc,
                                     b->geom1 + (kN + n) * C, b->geom2 + (kN + n) * C,
                                     b->force6 + (kN + n) * C * 6, b->cpos_z + (kN + n) * C, &n_r, &n_l, NULL,
                                     NULL, &grf_r, &grf_l, &min_z, &bad);
    if (rc0 != OLY_OK) return rc0;
    if (nc != nc_raw) bad = 1;
    /* a one-env view of the inputs and of the task state for a3_row */
    oly_a3_inputs in;
    in.qpos = b->qpos + (kN + n) * m->nq; in.qvel = b->qvel + (kN + n) * m->nv;
    in.act_len = b->act_len + (kN + n) * nu; in.act_vel = b->act_vel + (kN + n) * nu;
    in.lf_pos = b->lf_pos + (kN + n) * 3; in.rf_pos = b->rf_pos + (kN + n) * 3;
    in.lf_vel = b->lf_vel + (kN + n) * 3; in.rf_vel = b->rf_vel + (kN + n) * 3;
    in.root_pos = b->root_pos + (kN + n) * 3; in.root_quat = b->root_quat + (kN + n) * 4;
    in.head_pos = b->head_pos + (kN + n) * 3;
    in.grf_l = &grf_l; in.grf_r = &grf_r; in.min_z = &min_z; in.n_r = &n_r; in.n_l = &n_l; in.bad = &bad;
    int32_t phase = st->phase[n], t1 = st->t1[n], t2 = st->t2[n], frames = st->reached_frames[n];
    uint8_t reached = st->target_reached[n];
    double goal[8];
    oly_a3_state s1;
    s1.phase = &phase; s1.t1 = &t1; s1.t2 = &t2; s1.reached_frames = &frames; s1.target_reached = &reached;
    s1.mode = st->mode + n; s1.seq_len = st->seq_len + n;
    s1.sequence = st->sequence + (size_t)n * OLY_MAX_SEQ * 4; s1.goal = goal;
    double obs[64], rew[6], tot;
    uint8_t done;
    a3_row(m, 0, &in, &s1, obs, rew, &tot, &done);
    int len = ro->traj_len[n] + 1;
    int cut = done || len >= ro->max_traj_len || t == ro->T - 1;
    int need_reset = reset_all || (cut && t < ro->T - 1);
    if (!reset_all) {
      ro->buf_rewards[tN + n] = tot;
      if (ro->buf_rew6)
        for (int i = 0; i < 6; ++i) ro->buf_rew6[(tN + n) * 6 + i] = (float)rew[i];
      ro->buf_flags[tN + n] = (uint8_t)((cut ? OLY_FLAG_LAST : 0) | (done ? OLY_FLAG_ABSORBING : 0));
      ro->traj_len[n] = cut ? 0 : len;
      if (cut && !done) { /* finish_path(last_val = (not done) * V(state)): V of THIS observation */
        int sc = ro->side_count[n];
        if (sc < ro->side_slots) {
          size_t srow = (size_t)n * ro->side_slots + sc;
          for (int c = 0; c < n_obs; ++c) ro->side_obs[srow * n_obs + c] = (float)obs[c];
          ro->side_t[srow] = t;
        }
        ro->side_count[n] = sc + 1;
      }
    }
    float* next = ro->state + (size_t)n * n_obs;
    for (int c = 0; c < n_obs; ++c) next[c] = (float)obs[c];
    if (need_reset) {
      int rc = ro->pool_count[n];
      const oly_a3_reset_record* rec = ro->pool + (size_t)n * ro->pool_depth + (unsigned)rc % (unsigned)ro->pool_depth;
      int new_len = rec->seq_len < 1 ? 1 : (rec->seq_len > OLY_MAX_SEQ ? OLY_MAX_SEQ : rec->seq_len);
      /* transform_sequence: mid point of the feet, root yaw = quat2euler(root xquat)[2] */
      double R[3][3];
      quat2mat(in.root_quat, R);
      double cy = sqrt(R[0][0] * R[0][0] + R[1][0] * R[1][0]);
      double root_yaw = (cy > 4.0 * 2.220446049250313e-16) ? atan2(R[1][0], R[0][0]) : 0.0;
      double cyw = cos(root_yaw), syw = sin(root_yaw);
      double mid0 = (in.lf_pos[0] + in.rf_pos[0]) / 2, mid1 = (in.lf_pos[1] + in.rf_pos[1]) / 2;
      double* seq = (double*)st->sequence + (size_t)n * OLY_MAX_SEQ * 4;
      for (int r = 0; r < OLY_MAX_SEQ; ++r) {
        if (r < new_len) {
          double x = rec->seq[r][0], y = rec->seq[r][1];
          seq[4 * r] = mid0 + x * cyw - y * syw;
          seq[4 * r + 1] = mid1 + x * syw + y * cyw;
          seq[4 * r + 2] = rec->seq[r][2];
          seq[4 * r + 3] = root_yaw + rec->seq[r][3];
        } else {
          seq[4 * r] = seq[4 * r + 1] = seq[4 * r + 2] = seq[4 * r + 3] = 0.0;
        }
      }
      st->phase[n] = rec->phase;
      st->t1[n] = 0;
      st->t2[n] = (new_len == 1) ? 0 : 1; /* t1 = t2 = 0, then update_target_steps :228-244 */
      st->reached_frames[n] = 0;
      st->target_reached[n] = 0;
      ((int32_t*)st->mode)[n] = rec->mode;
      ((int32_t*)st->seq_len)[n] = new_len;
      for (int i = 0; i < 8; ++i) st->goal[8 * (size_t)n + i] = 0.0;
      ro->pool_count[n] = rc + 1;
      /* get_obs of the un-advanced task: clock of the drawn phase, goal steps zero */
      double ang = 2 * PI * rec->phase / (double)m->period;
      next[7 + 2 * nu] = (float)sin(ang);
      next[8 + 2 * nu] = (float)cos(ang);
      for (int i = 0; i < 8; ++i) next[9 + 2 * nu + i] = 0.0f;
    } else {
      st->phase[n] = phase; st->t1[n] = t1; st->t2[n] = t2;
      st->reached_frames[n] = frames; st->target_reached[n] = reached;
      for (int i = 0; i < 8; ++i) st->goal[8 * (size_t)n + i] = goal[i];
    }
  }
  if (!reset_all) {
    ro->ctr[0] = t + 1;
    ro->ctr[1] = k + 1;
  }
  return OLY_OK;
}

/* ============================================================================== K11 */

/* One ReLU MLP in -> 256 -> 256 -> out as the f32 matrix cores evaluate it: every pre-activation is
   the fmaf chain over k ascending from 0, bias added after the chain; the output layer runs eight
   chains over k in [32w, 32w + 32) and adds them in order w = 0..7, then the bias.
   Gaussian_FF_Actor._get_dist_params (rl/policies/actor.py:180-195) / FF_V.forward (critic.py:62-74)
   up to the summation order of the Linear layers.                                                 */
int oly_mlp_forward_cpu(int N, int in_dim, int out_dim, const float* x, const float* w1, const float* b1,
                        const float* w2, const float* b2, const float* w3, const float* b3,
                        const float* in_mean, const float* in_std, float* y) {
  enum { H = 256 };
  if (in_dim > 64 || out_dim > 32) return OLY_ERANGE;
  for (int n = 0; n < N; ++n) {
    float xin[64], h1[H], h2[H];
    for (int k = 0; k < in_dim; ++k) {
      float v = x[(size_t)n * in_dim + k];
      if (in_mean && in_std) v = (v - in_mean[k]) / in_std[k];
      xin[k] = v;
    }
    for (int j = 0; j < H; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < in_dim; ++k) acc = fmaf(xin[k], w1[(size_t)j * in_dim + k], acc);
      float v = acc + b1[j];
      h1[j] = (v > 0.0f || v != v) ? v : 0.0f;
    }
    for (int j = 0; j < H; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < H; ++k) acc = fmaf(h1[k], w2[(size_t)j * H + k], acc);
      float v = acc + b2[j];
      h2[j] = (v > 0.0f || v != v) ? v : 0.0f;
    }
    for (int j = 0; j < out_dim; ++j) {
      float s = 0.0f;
      for (int w = 0; w < 8; ++w) {
        float acc = 0.0f;
        for (int k = 32 * w; k < 32 * w + 32; ++k) acc = fmaf(h2[k], w3[(size_t)j * H + k], acc);
        s = (w == 0) ? acc : s + acc;
      }
      y[(size_t)n * out_dim + j] = s + b3[j];
    }
  }
  return OLY_OK;
}

/* ============================================================================== K6 */

int oly_return_scan_cpu(int mode, int T, int N, double gamma, double lam, const float* rew,
                        const float* val, const float* next_val, const uint8_t* flags, float* ret,
                        float* adv) {
  if (mode == OLY_SCAN_RETURN) {
    /* PPOBuffer.finish_path, rl/algos/ppo.py:68-84, as numpy 2.x evaluates it:
       R = last_val (float32 array), R = gamma*R + reward: the FIRST product is a python
       float times a float32 array = float32 arithmetic; adding the float64 reward
       promotes to float64, where the scan stays.  torch.Tensor(...) narrows (:328);
       advantages = returns - values in float32 (:335).                                */
    float g32 = (float)gamma;
    for (int n = 0; n < N; ++n) {
      double R = 0.0;
      for (int t = T - 1; t >= 0; --t) {
        size_t e = (size_t)t * N + n;
        uint8_t f = flags[e];
        if ((f & OLY_FLAG_LAST) || t == T - 1) {
          float r0 = (f & OLY_FLAG_ABSORBING) ? 0.0f : next_val[e]; /* (not done) * value */
          float p = g32 * r0;
          R = (double)p + (double)rew[e];
        } else {
          R = gamma * R + (double)rew[e];
        }
        ret[e] = (float)R;
        adv[e] = ret[e] - val[e];
      }
    }
    return OLY_OK;
  }
  if (mode == OLY_SCAN_GAE) {
    /* mushroom_rl.utils.value_functions.compute_gae (mushroom-rl 1.10, absent here:
       parity unpinned), call site imitation_lib/imitation/gail_TRPO.py:126-127.  All
       operands are float32 arrays / python floats, so numpy works in float32.         */
    float g32 = (float)gamma, gl32 = (float)(gamma * lam);
    for (int n = 0; n < N; ++n) {
      float a_next = 0.0f;
      for (int t = T - 1; t >= 0; --t) {
        size_t e = (size_t)t * N + n;
        uint8_t f = flags[e];
        float a;
        if ((f & OLY_FLAG_LAST) || t == T - 1) {
          a = rew[e] - val[e];
          if (!(f & OLY_FLAG_ABSORBING)) a += g32 * next_val[e];
        } else {
          a = rew[e] + g32 * next_val[e] - val[e] + gl32 * a_next;
        }
        adv[e] = a;
        ret[e] = a + val[e];
        a_next = a;
      }
    }
    return OLY_OK;
  }
  return OLY_EINVAL;
}

/* finish_path with the float64 reward un-narrowed (WrapEnv.step hands np.array([reward]), a
   float64 array, to PPOBuffer.store: rl/envs/wrappers.py:14, rl/algos/ppo.py:56-66,74-76). */
int oly_return_scan_r64_cpu(int T, int N, double gamma, const double* rew, const float* val,
                            const float* next_val, const uint8_t* flags, float* ret, float* adv) {
  float g32 = (float)gamma;
  for (int n = 0; n < N; ++n) {
    double R = 0.0;
    for (int t = T - 1; t >= 0; --t) {
      size_t e = (size_t)t * N + n;
      uint8_t f = flags[e];
      if ((f & OLY_FLAG_LAST) || t == T - 1) {
        float r0 = (f & OLY_FLAG_ABSORBING) ? 0.0f : next_val[e];
        float p = g32 * r0;
        R = (double)p + rew[e];
      } else {
        R = gamma * R + rew[e];
      }
      ret[e] = (float)R;
      adv[e] = ret[e] - val[e];
    }
  }
  return OLY_OK;
}

/* ============================================================================== K7 */

int oly_adv_stats_cpu(int64_t n, const float* x, double* stats3_out) {
  double s = 0.0, ss = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    double v = (double)x[i];
    s += v;
    ss += v * v;
  }
  stats3_out[0] = (double)n;
  stats3_out[1] = s;
  stats3_out[2] = ss;
  return OLY_OK;
}

/* rl/algos/ppo.py:335-336 (ddof 1, eps 1e-5) / gail_TRPO.py:128 (ddof 0, eps 1e-8). */
int oly_adv_normalize_cpu(int64_t n, float* x, const double* stats3, int ddof, double eps) {
  double cnt = stats3[0], mean = stats3[1] / cnt;
  double var = (stats3[2] - cnt * mean * mean) / (cnt - (double)ddof);
  if (var < 0.0) var = 0.0;
  double denom = sqrt(var) + eps;
  for (int64_t i = 0; i < n; ++i) x[i] = (float)(((double)x[i] - mean) / denom);
  return OLY_OK;
}

/* The sharded form: one (count, sum, sumsq) triple per rank, added by a balanced pairwise tree in
   rank order (what every rank does after the all-gather), then the normalisation above. */
int oly_adv_normalize_parts_cpu(int64_t n, float* x, const double* parts3, int parts, int ddof, double eps) {
  double tot[3], a[64];
  if (parts < 1 || parts > 64) return OLY_EINVAL;
  for (int k = 0; k < 3; ++k) {
    int m = 1;
    while (m < parts) m <<= 1;
    for (int i = 0; i < m; ++i) a[i] = (i < parts) ? parts3[3 * i + k] : 0.0;
    for (int h = 1; h < m; h <<= 1)
      for (int i = 0; i + h < m; i += 2 * h) a[i] += a[i + h];
    tot[k] = a[0];
  }
  return oly_adv_normalize_cpu(n, x, tot, ddof, eps);
}

/* Standardizer.update_mean_std sums, imitation_lib/utils/networks.py:76-79. */
int oly_col_stats_cpu(int B, int D, const float* x, double* colstats, int accumulate) {
  for (int j = 0; j < D; ++j) {
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < B; ++b) {
      double v = (double)x[(size_t)b * D + j];
      s += v;
      ss += v * v;
    }
    if (accumulate) {
      colstats[j] += (double)B;
      colstats[D + j] += s;
      colstats[2 * D + j] += ss;
    } else {
      colstats[j] = (double)B;
      colstats[D + j] = s;
      colstats[2 * D + j] = ss;
    }
  }
  return OLY_OK;
}

/* ============================================================================== K8 */

/* prepare_discrim_inputs (gail_TRPO.py:297-313) + Standardizer.forward (networks.py:68-74):
   float32 batch minus float64 mean, divided by float64 std, narrowed by .float().       */
int oly_disc_standardize_cpu(int B, int Dx, int D, const float* x, const int32_t* mask,
                             const double* mean, const double* std, float* out) {
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < D; ++j) {
      int c = mask ? mask[j] : j;
      if (c < 0 || c >= Dx) return OLY_ERANGE;
      out[(size_t)b * D + j] = (float)(((double)x[(size_t)b * Dx + c] - mean[j]) / std[j]);
    }
  return OLY_OK;
}

/* reparameterize, networks.py:21-24 (float32 torch ops). */
int oly_disc_reparam_cpu(int64_t n, const float* mu, const float* logvar, const float* eps,
                         float* z) {
  for (int64_t i = 0; i < n; ++i) z[i] = mu[i] + expf(logvar[i] / 2.0f) * eps[i];
  return OLY_OK;
}

/* GAIL.make_discrim_reward, gail_TRPO.py:326-327 on a float32 logit array. */
int oly_disc_reward_cpu(int64_t B, const float* logits, float* reward) {
  for (int64_t i = 0; i < B; ++i) {
    float e = expf(-logits[i]);
    float p = 1.0f / (1.0f + e);
    float q = 1.0f - p + 1e-8f;
    reward[i] = -logf(q);
  }
  return OLY_OK;
}

/* ============================================================================== K12 */

/* exp in float32 from fma / rint / exponent arithmetic only (the same operations, in the same order, as
   exp32 of csrc/k12_disc_forward.hip, so both return the same bits): n = rint(x log2 e), r = x - n ln2 by the
   two-constant Cody-Waite split, e^r = 1 + (r + r^2 P(r)) with a degree-5 minimax P, scaled by 2^n in two
   exact steps.  <= 1 ulp of exp (test_oracle_golden.py::test_exp32_against_fp64_exp), the error class of
   torch's float32 exp that reparameterize (networks.py:21-24) calls.                                     */
static float oly_pow2i(int e) {
  union { uint32_t u; float f; } c;
  c.u = (uint32_t)(e + 127) << 23;
  return c.f;
}
float oly_exp32_cpu(float x) {
  if (x != x) return x;
  if (x > 88.72283935546875f) return INFINITY;
  if (x < -103.97208404541016f) return 0.0f;
  const float n = rintf(x * 1.4426950408889634f);
  float r = fmaf(n, -0.693145751953125f, x);
  r = fmaf(n, -1.428606765330187045e-06f, r);
  float u = 0.000198527617612853646278381f;
  u = fmaf(u, r, 0.00139304355252534151077271f);
  u = fmaf(u, r, 0.00833336077630519866943359f);
  u = fmaf(u, r, 0.0416664853692054748535156f);
  u = fmaf(u, r, 0.166666671633720397949219f);
  u = fmaf(u, r, 0.5f);
  u = 1.0f + fmaf(r * r, u, r);
  const int q = (int)n, q1 = q >> 1;
  return (u * oly_pow2i(q1)) * oly_pow2i(q - q1);
}

/* GAIL.make_discrim_reward (gail_TRPO.py:320-327) through VAIL.discrim_output (vail_TRPO.py:18-21),
   prepare_discrim_inputs (gail_TRPO.py:297-313), VariationalNet.forward (networks.py:258-284):
   Standardizer.forward's (x - mean) / std (networks.py:68-74; statistics given), encoder in -> 256 -> 128 with
   ReLU after both layers (examples/imitation_learning/utils.py:157-159), mu / logvar heads, reparameterize
   (networks.py:21-24) with the caller's eps, decoder 128 -> 1, reward formula.  Linear layers as the f32
   matrix cores evaluate them: fmaf chain over k ascending from 0, bias after the chain; the decoder is the
   chain over k < 64 plus the chain over 64 <= k < 128, then the bias.  The reward takes numpy's float32
   steps with each transcendental correctly rounded (through fp64).
   outputs (each may be NULL): reward [B], logits [B], mu [B,128], logvar [B,128].                       */
int oly_disc_forward_cpu(int64_t B, int Dx, int D, const float* x, const int32_t* mask, const double* mean,
                         const double* sd, const double* colstats, const float* enc_w0, const float* enc_b0, const float* enc_w1,
                         const float* enc_b1, const float* mu_w, const float* mu_b, const float* lv_w,
                         const float* lv_b, const float* dec_w, const float* dec_b, const float* eps,
                         float* reward, float* logits, float* mu_out, float* logvar_out) {
  enum { H1 = 256, H2 = 128, Z = 128 };
  if (D <= 0 || D > 64 || (!mask && D != Dx) || ((mean == NULL) != (sd == NULL)) || (mean && colstats)) return OLY_EINVAL;
  double cmean[64], csd[64];
  if (colstats) {
    /* Standardizer.update_mean_std (networks.py:76-81) on the running sums; _sumsq and _count start at
       1e-2 (networks.py:54-56), the variance is floored at 1e-2                                        */
    for (int k = 0; k < D; ++k) {
      const double cnt = colstats[k] + 1e-2;
      cmean[k] = colstats[D + k] / cnt;
      csd[k] = sqrt(fmax((colstats[2 * D + k] + 1e-2) / cnt - cmean[k] * cmean[k], 1e-2));
    }
    mean = cmean;
    sd = csd;
  }
  for (int64_t n = 0; n < B; ++n) {
    float xs[64], h1[H1], h2[H2], z[Z];
    for (int k = 0; k < D; ++k) {
      const int c = mask ? mask[k] : k;
      if (c < 0 || c >= Dx) return OLY_ERANGE;
      float v = x[(size_t)n * Dx + c];
      if (mean) v = (float)(((double)v - mean[k]) / sd[k]);
      xs[k] = v;
    }
    for (int j = 0; j < H1; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < D; ++k) acc = fmaf(xs[k], enc_w0[(size_t)j * D + k], acc);
      const float v = acc + enc_b0[j];
      h1[j] = (v > 0.0f || v != v) ? v : 0.0f;
    }
    for (int j = 0; j < H2; ++j) {
      float acc = 0.0f;
      for (int k = 0; k < H1; ++k) acc = fmaf(h1[k], enc_w1[(size_t)j * H1 + k], acc);
      const float v = acc + enc_b1[j];
      h2[j] = (v > 0.0f || v != v) ? v : 0.0f;
    }
    for (int j = 0; j < Z; ++j) {
      float am = 0.0f, al = 0.0f;
      for (int k = 0; k < H2; ++k) {
        am = fmaf(h2[k], mu_w[(size_t)j * H2 + k], am);
        al = fmaf(h2[k], lv_w[(size_t)j * H2 + k], al);
      }
      const float m = am + mu_b[j], lv = al + lv_b[j];
      z[j] = eps ? m + oly_exp32_cpu(lv / 2.0f) * eps[(size_t)n * Z + j] : m;
      if (mu_out) mu_out[(size_t)n * Z + j] = m;
      if (logvar_out) logvar_out[(size_t)n * Z + j] = lv;
    }
    float lo = 0.0f, hi = 0.0f;
    for (int k = 0; k < 64; ++k) lo = fmaf(z[k], dec_w[k], lo);
    for (int k = 64; k < 128; ++k) hi = fmaf(z[k], dec_w[k], hi);
    const float d = (lo + hi) + dec_b[0];
    if (logits) logits[n] = d;
    if (reward) {
      const float e = (float)exp(-(double)d);
      const float p = 1.0f / (1.0f + e);
      const float q = 1.0f - p + 1e-8f;
      reward[n] = -(float)log((double)q);
    }
  }
  return OLY_OK;
}

/* Normalize._obfilt, rl/envs/normalize.py:139-147 (obs f32 in, statistics f64). */
int oly_obs_filter_cpu(int B, int D, const float* x, const double* mean, const double* var,
                       double eps, double clip, float* out) {
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < D; ++j) {
      double v = ((double)x[(size_t)b * D + j] - mean[j]) / sqrt(var[j] + eps);
      if (clip > 0.0) v = v < -clip ? -clip : (v > clip ? clip : v);
      out[(size_t)b * D + j] = (float)v;
    }
  return OLY_OK;
}

/* x @ M with M from _get_symmetry_matrix (rl/envs/wrappers.py:51-57,75-82), stated on the
 * (src, sign) form of the signed permutation. */
int oly_signed_perm_cpu(int B, int D, const float* x, const int32_t* src, const float* sign,
                        float* out) {
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < D; ++j) out[(size_t)b * D + j] = sign[j] * x[(size_t)b * D + src[j]];
  return OLY_OK;
}

/* mirror loss, rl/algos/ppo.py:261-268, and its gradients. */
int oly_mirror_loss_cpu(int B, int A, const float* det, const float* mir, const int32_t* src,
                        const float* sign, double* loss_out, float* grad_det, float* grad_mir) {
  double acc = 0.0;
  const float gs = (float)(2.0 / ((double)B * A));
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < A; ++j) {
      const size_t e = (size_t)b * A + j, m = (size_t)b * A + src[j];
      const float d = det[e] - sign[j] * mir[m];
      acc += (double)(d * d);
      if (grad_det) grad_det[e] = gs * d;
      if (grad_mir) grad_mir[m] = -sign[j] * (gs * d);
    }
  loss_out[0] = acc / ((double)B * A);
  return OLY_OK;
}

static float oly_sd_at(const float* sd, int mode, int row, int A, int j) {
  return mode == OLY_STD_SCALAR ? sd[0] : mode == OLY_STD_PER_DIM ? sd[j] : sd[(size_t)row * A + j];
}

/* PPO.update_policy, rl/algos/ppo.py:236-259,270-273 with torch.distributions.Normal's
 * log_prob / entropy written out; fp32 per element as torch evaluates, fp64 sums. */
int oly_ppo_loss_cpu(int B, int A, const float* mu, const float* sd, int sd_mode,
                     const float* old_mu, const float* old_sd, int old_sd_mode,
                     const float* action, const float* adv, const float* ret, const float* value,
                     float clip, float vf_coeff, double* scal_out, float* grad_mu, float* grad_sd,
                     float* grad_value) {
  const float c = 0.9189385332046727f, lo = 1.0f - clip, hi = 1.0f + clip;
  const float invB = 1.0f / (float)B;
  double s_actor = 0, s_ent = 0, s_crit = 0, s_kl = 0, s_cf = 0;
  for (int b = 0; b < B; ++b) {
    float lp = 0.0f, olp = 0.0f, ent = 0.0f;
    for (int j = 0; j < A; ++j) {
      const size_t e = (size_t)b * A + j;
      const float s = oly_sd_at(sd, sd_mode, b, A, j), os = oly_sd_at(old_sd, old_sd_mode, b, A, j);
      const float t = action[e] - mu[e], ot = action[e] - old_mu[e];
      lp += -(t * t) / (2.0f * (s * s)) - logf(s) - c;
      olp += -(ot * ot) / (2.0f * (os * os)) - logf(os) - c;
      ent += (0.5f + c) + logf(s);
    }
    const float lr = lp - olp, ratio = expf(lr);
    const float cpi = ratio * adv[b];
    const float rc = ratio < lo ? lo : (ratio > hi ? hi : ratio);
    const float cl = rc * adv[b];
    s_actor += (double)(cpi < cl ? cpi : cl);
    s_ent += (double)ent;
    const float dv = ret[b] - value[b];
    s_crit += (double)(dv * dv);
    s_kl += (double)((ratio - 1.0f) - lr);
    s_cf += fabsf(ratio - 1.0f) > clip ? 1.0 : 0.0;
    if (grad_value) grad_value[b] = vf_coeff * 2.0f * (value[b] - ret[b]) * invB;
    const float inr = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
    const float w = cpi < cl ? 1.0f : (cpi == cl ? 0.5f + 0.5f * inr : inr);
    const float g_lp = -invB * adv[b] * w * ratio;
    for (int j = 0; j < A; ++j) {
      const size_t e = (size_t)b * A + j;
      const float s = oly_sd_at(sd, sd_mode, b, A, j);
      const float t = action[e] - mu[e];
      if (grad_mu) grad_mu[e] = g_lp * t / (s * s);
      if (grad_sd) grad_sd[e] = g_lp * (t * t / (s * s * s) - 1.0f / s);
    }
  }
  scal_out[0] = -s_actor / B;
  scal_out[1] = -s_ent / ((double)B * A);
  scal_out[2] = (double)vf_coeff * s_crit / B;
  scal_out[3] = s_kl / B;
  scal_out[4] = s_cf / B;
  return OLY_OK;
}

/* UnitreeH1._get_ground_forces (UnitreeH1.py:113-123) per substep + RunningAveragedWindow mean
 * (loco_env_base.py:1072-1084,1163-1174), restating mushroom's _get_collision_force: first
 * contact in contact order between the two geom groups (either order), its force[:3].
 * ncon is the raw data.ncon; only C slots are staged.  overflow[n] = 1 when a substep of env n has
 * ncon > C and a sensor pair without a hit among the staged slots (the reference would have gone
 * on scanning the surplus), or a negative count. */
int oly_il_ground_forces_cpu(int ngeom, const int32_t* geom_group, int n_pairs, const int32_t* pair_a,
                             const int32_t* pair_b, int W, int N, int C, const int32_t* ncon,
                             const int32_t* geom1, const int32_t* geom2, const double* force6,
                             double* grf_step, double* grf_mean, uint8_t* overflow) {
  const int ncomp = 3 * n_pairs;
  for (int n = 0; n < N; ++n) {
    double acc[3 * OLY_MAX_GRF_PAIRS] = {0};
    int over = 0;
    for (int w = 0; w < W; ++w) {
      const size_t row = (size_t)w * N + n;
      const int nc_raw = ncon[row];
      int nc = nc_raw;
      if (nc < 0) { nc = 0; over = 1; }
      if (nc > C) nc = C;
      for (int k = 0; k < n_pairs; ++k) {
        double f[3] = {0.0, 0.0, 0.0};
        int found = 0;
        for (int i = 0; i < nc; ++i) {
          const int g1 = geom1[row * C + i], g2 = geom2[row * C + i];
          if (g1 < 0 || g1 >= ngeom || g2 < 0 || g2 >= ngeom) continue;
          const int ga = geom_group[g1], gb = geom_group[g2];
          if (ga < 0 || gb < 0) continue;
          if ((ga == pair_a[k] && gb == pair_b[k]) || (ga == pair_b[k] && gb == pair_a[k])) {
            for (int c = 0; c < 3; ++c) f[c] = force6[(row * C + i) * 6 + c];
            found = 1;
            break;
          }
        }
        if (!found && nc_raw > C) over = 1;
        for (int c = 0; c < 3; ++c) {
          if (grf_step) grf_step[row * ncomp + 3 * k + c] = f[c];
          acc[3 * k + c] += f[c];
        }
      }
    }
    for (int j = 0; j < ncomp; ++j) grf_mean[(size_t)n * ncomp + j] = acc[j] / (double)W;
    if (overflow) overflow[n] = (uint8_t)over;
  }
  return OLY_OK;
}

/* PPO.sample's episode-cut bookkeeping, rl/algos/ppo.py:169-196, for N envs in lock step. */
int oly_rollout_cuts_cpu(int N, int max_traj_len, int last_step, const uint8_t* done, int32_t* traj_len,
                         uint8_t* flags, int32_t* n_cut) {
  int c = 0;
  for (int n = 0; n < N; ++n) {
    const int len = traj_len[n] + 1;
    const int d = done[n] != 0;
    const int cut = d || len >= max_traj_len || last_step != 0;
    flags[n] = (uint8_t)((cut ? OLY_FLAG_LAST : 0) | (d ? OLY_FLAG_ABSORBING : 0));
    traj_len[n] = cut ? 0 : len;
    c += cut;
  }
  *n_cut = c;
  return OLY_OK;
}

/* ============================================================================== K14 */

/* One network's share of oly_ppo_update_grads (csrc/k14_ppo_update.hip): the gradients of one PPO minibatch
 * update, PPO.update_policy + the backward() calls of PPO.train (rl/algos/ppo.py:232-282,396-410), as the f32
 * matrix cores evaluate them:
 *   forward     as oly_mlp_forward_cpu (k-ascending fmaf chains, the output layer as eight partial chains);
 *   loss        oly_ppo_loss_cpu's per-row arithmetic with exp -> oly_exp32_cpu and log(std) given; the mirror
 *               loss of oly_mirror_loss_cpu (ppo.py:261-268) with d mirror / d det joined to d mu; with it an actor
 *               tile is 8 rows of the minibatch (tile rows 0-7) + their mirrored observations (tile rows 8-15);
 *   dH          fmaf chain over the layer's output index ascending (padded to 16 for the output layer);
 *   dW, per part (tiles part, part + parts, ... of 16 rows): ONE fmaf chain over the part's rows in tile order,
 *               inside a tile rows 0,4,8,12,1,5,9,13,...; bias gradients as four running f32 sums per column
 *               (rows 4 g .. 4 g + 3 of every tile into sum g), combined as (s0 + s1) + (s2 + s3);
 *   the parts are added in fp64 in four contiguous groups of ceil(parts / 4), each in order, the group sums as
 *   ((S0 + S1) + S2) + S3, rounded once.
 * critic != 0: out_dim = 1, loss = vf_coeff * mse.  stats[6]: sums of surrogate, kl, clipped, mirror, critic, rows. */
typedef struct {
  const float *w1, *b1, *w2, *b2, *w3, *b3, *mean, *std;
} oly_upd_net;

static void upd_forward(const oly_upd_net* nw, int in_dim, int out_dim, const float* xrow, float* xin, float* h1,
                        float* h2, float* out16) {
  enum { H = 256 };
  for (int k = 0; k < in_dim; ++k) {
    float v = xrow ? xrow[k] : 0.0f;
    if (xrow && nw->mean && nw->std) v = (v - nw->mean[k]) / nw->std[k];
    xin[k] = v;
  }
  for (int j = 0; j < H; ++j) {
    float acc = 0.0f;
    for (int k = 0; k < in_dim; ++k) acc = fmaf(xin[k], nw->w1[(size_t)j * in_dim + k], acc);
    const float v = acc + nw->b1[j];
    h1[j] = (v > 0.0f || v != v) ? v : 0.0f;
  }
  for (int j = 0; j < H; ++j) {
    float acc = 0.0f;
    for (int k = 0; k < H; ++k) acc = fmaf(h1[k], nw->w2[(size_t)j * H + k], acc);
    const float v = acc + nw->b2[j];
    h2[j] = (v > 0.0f || v != v) ? v : 0.0f;
  }
  for (int j = 0; j < 16; ++j) {
    float s = 0.0f;
    if (j < out_dim) {
      for (int w = 0; w < 8; ++w) {
        float acc = 0.0f;
        for (int k = 32 * w; k < 32 * w + 32; ++k) acc = fmaf(h2[k], nw->w3[(size_t)j * H + k], acc);
        s = (w == 0) ? acc : s + acc;
      }
      s += nw->b3[j];
    }
    out16[j] = s;
  }
}

static int upd_network(int critic, int B, int in_dim, int out_dim, int parts, const float* obs, const float* mir_obs,
                       const float* action, const float* adv, const float* ret, const float* old_mu,
                       const int32_t* idx, const oly_upd_net* nw, const float* sd, const float* log_sd,
                       const float* old_sd, const float* old_log_sd, const int32_t* act_src, const float* act_sign,
                       float clip, float vf_coeff, float mirror_coeff, float* grad, double* stats) {
  enum { H = 256, R = 16 };
  const float c = 0.9189385332046727f, lo = 1.0f - clip, hi = 1.0f + clip;
  const float inv_b = 1.0f / (float)B;
  const float gscale = (float)(2.0 / ((double)B * out_dim));
  const int mirror = !critic && mir_obs != NULL;
  /* with the mirror loss a tile holds 8 rows of the minibatch (tile rows 0-7) and their mirrored observations (8-15) */
  const int rpt = mirror ? 8 : R;
  const int ntiles = (B + rpt - 1) / rpt;
  const size_t oW1 = 0, ob1 = (size_t)H * in_dim, oW2 = ob1 + H, ob2 = oW2 + (size_t)H * H, oW3 = ob2 + H,
               ob3 = oW3 + (size_t)out_dim * H, gf = ob3 + out_dim;
  double* total = (double*)calloc(4 * gf, sizeof(double));      /* four group sums per element */
  float* G = (float*)malloc(gf * sizeof(float));
  float* sb2 = (float*)malloc(sizeof(float) * H * 4), *sb1 = (float*)malloc(sizeof(float) * H * 4);
  float* xin = (float*)malloc(sizeof(float) * R * 64), *h1 = (float*)malloc(sizeof(float) * R * H),
        *h2 = (float*)malloc(sizeof(float) * R * H), *dz2 = (float*)malloc(sizeof(float) * R * H),
        *dz1 = (float*)malloc(sizeof(float) * R * H);
  if (!total || !G || !sb2 || !sb1 || !xin || !h1 || !h2 || !dz2 || !dz1) return OLY_EINVAL;
  for (int part = 0; part < parts; ++part) {
    memset(G, 0, gf * sizeof(float));
    memset(sb2, 0, sizeof(float) * H * 4);
    memset(sb1, 0, sizeof(float) * H * 4);
    float sb3[16][4];
    memset(sb3, 0, sizeof(sb3));
    for (int tile = part; tile < ntiles; tile += parts) {
      {
        float out[R][16], dz3[R][16];
        memset(dz3, 0, sizeof(dz3));
        for (int m = 0; m < R; ++m) {
          const int r = mirror ? tile * 8 + (m & 7) : tile * R + m;
          const float* xrow = NULL;
          if (r < B) {
            const size_t s = idx ? (size_t)idx[r] : (size_t)r;
            xrow = ((mirror && m >= 8) ? mir_obs : obs) + s * in_dim;
          }
          upd_forward(nw, in_dim, out_dim, xrow, xin + m * 64, h1 + m * H, h2 + m * H, out[m]);
        }
        {
          for (int m = 0; m < rpt; ++m) {
            const int r = tile * rpt + m;
            if (r >= B) continue;
            const size_t s = idx ? (size_t)idx[r] : (size_t)r;
            if (critic) {
              const float v = out[m][0], dv = ret[s] - v;
              stats[4] += (double)(dv * dv);
              dz3[m][0] = vf_coeff * 2.0f * (v - ret[s]) * inv_b;
              continue;
            }
            float t[16], lp = 0.0f, olp = 0.0f;
            for (int j = 0; j < out_dim; ++j) {
              const float a = action[s * out_dim + j];
              t[j] = a - out[m][j];
              const float ot = a - old_mu[s * out_dim + j];
              lp += -(t[j] * t[j]) / (2.0f * (sd[j] * sd[j])) - log_sd[j] - c;
              olp += -(ot * ot) / (2.0f * (old_sd[j] * old_sd[j])) - old_log_sd[j] - c;
            }
            const float lr = lp - olp, ratio = oly_exp32_cpu(lr);
            const float cpi = ratio * adv[s];
            const float rc = fminf(fmaxf(ratio, lo), hi);
            const float cl = rc * adv[s];
            stats[0] += (double)fminf(cpi, cl);
            stats[1] += (double)((ratio - 1.0f) - lr);
            stats[2] += fabsf(ratio - 1.0f) > clip ? 1.0 : 0.0;
            stats[5] += 1.0;
            const float inr = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
            const float w = cpi < cl ? 1.0f : (cpi == cl ? 0.5f + 0.5f * inr : inr);
            const float g_lp = -inv_b * adv[s] * w * ratio;
            for (int j = 0; j < out_dim; ++j) {
              float gm = g_lp * t[j] / (sd[j] * sd[j]);
              if (mirror) {
                const int i = act_src[j];
                const float sg = act_sign[j];
                const float d = out[m][j] - sg * out[m + 8][i];     /* the mirrored row is tile row m + 8 */
                stats[3] += (double)(d * d);
                const float gg = gscale * d;
                gm = gm + mirror_coeff * gg;
                dz3[m + 8][i] = mirror_coeff * (-sg * gg);
              }
              dz3[m][j] = gm;
            }
          }
        }
        /* ---- backward of this tile */
        for (int m = 0; m < R; ++m)
          for (int k = 0; k < H; ++k) {
            float acc = 0.0f;
            for (int n = 0; n < 16; ++n) acc = fmaf(dz3[m][n], n < out_dim ? nw->w3[(size_t)n * H + k] : 0.0f, acc);
            dz2[m * H + k] = h2[m * H + k] > 0.0f ? acc : 0.0f;
          }
        for (int m = 0; m < R; ++m)
          for (int k = 0; k < H; ++k) {
            float acc = 0.0f;
            for (int n = 0; n < H; ++n) acc = fmaf(dz2[m * H + n], nw->w2[(size_t)n * H + k], acc);
            dz1[m * H + k] = h1[m * H + k] > 0.0f ? acc : 0.0f;
          }
        for (int q = 0; q < 4; ++q)
          for (int jj = 0; jj < 4; ++jj) {
            const int m = 4 * jj + q;
            for (int n = 0; n < out_dim; ++n)
              for (int k = 0; k < H; ++k) G[oW3 + (size_t)n * H + k] = fmaf(dz3[m][n], h2[m * H + k], G[oW3 + (size_t)n * H + k]);
            for (int n = 0; n < H; ++n) {
              const float z2 = dz2[m * H + n], z1 = dz1[m * H + n];
              float* g2 = G + oW2 + (size_t)n * H;
              for (int k = 0; k < H; ++k) g2[k] = fmaf(z2, h1[m * H + k], g2[k]);
              float* g1 = G + oW1 + (size_t)n * in_dim;
              for (int k = 0; k < in_dim; ++k) g1[k] = fmaf(z1, xin[m * 64 + k], g1[k]);
            }
          }
        for (int g = 0; g < 4; ++g)
          for (int i = 0; i < 4; ++i) {
            const int m = 4 * g + i;
            for (int n = 0; n < H; ++n) {
              sb2[n * 4 + g] += dz2[m * H + n];
              sb1[n * 4 + g] += dz1[m * H + n];
            }
            for (int n = 0; n < out_dim; ++n) sb3[n][g] += dz3[m][n];
          }
      }
    }
    for (int n = 0; n < H; ++n) {
      G[ob2 + n] = (sb2[n * 4] + sb2[n * 4 + 1]) + (sb2[n * 4 + 2] + sb2[n * 4 + 3]);
      G[ob1 + n] = (sb1[n * 4] + sb1[n * 4 + 1]) + (sb1[n * 4 + 2] + sb1[n * 4 + 3]);
    }
    for (int n = 0; n < out_dim; ++n) G[ob3 + n] = (sb3[n][0] + sb3[n][1]) + (sb3[n][2] + sb3[n][3]);
    /* the finishing launch adds the parts in four contiguous groups of ceil(parts / 4), each in order: ((S0 + S1) + S2) + S3 */
    {
      const int chunk = (parts + 3) / 4, grp = part / chunk;
      double* S = total + (size_t)grp * gf;
      for (size_t e = 0; e < gf; ++e) S[e] += (double)G[e];
    }
  }
  for (size_t e = 0; e < gf; ++e)
    grad[e] = (float)(((total[e] + total[gf + e]) + total[2 * gf + e]) + total[3 * gf + e]);
  free(total); free(G); free(sb2); free(sb1); free(xin); free(h1); free(h2); free(dz2); free(dz1);
  return OLY_OK;
}

int oly_ppo_update_cpu(int B, int in_dim, int act_dim, int parts_actor, int parts_critic, const float* obs,
                       const float* mir_obs, const float* action, const float* adv, const float* ret,
                       const float* old_mu, const int32_t* idx, const float* const* actor_wb,
                       const float* a_mean, const float* a_std, const float* const* critic_wb,
                       const float* c_mean, const float* c_std, const float* sd, const float* log_sd,
                       const float* old_sd, const float* old_log_sd, const int32_t* act_src,
                       const float* act_sign, float clip, float vf_coeff, float mirror_coeff, float* grad_actor,
                       float* grad_critic, double* scal_out) {
  if (B <= 0 || in_dim <= 0 || in_dim > 64 || act_dim <= 0 || act_dim > 16 || parts_actor <= 0 || parts_critic <= 0)
    return OLY_ERANGE;
  const oly_upd_net a = {actor_wb[0], actor_wb[1], actor_wb[2], actor_wb[3], actor_wb[4], actor_wb[5], a_mean, a_std};
  const oly_upd_net cn = {critic_wb[0], critic_wb[1], critic_wb[2], critic_wb[3], critic_wb[4], critic_wb[5], c_mean, c_std};
  double st[6] = {0, 0, 0, 0, 0, 0};
  int rc = upd_network(0, B, in_dim, act_dim, parts_actor, obs, mir_obs, action, adv, ret, old_mu, idx, &a, sd, log_sd,
                       old_sd, old_log_sd, act_src, act_sign, clip, vf_coeff, mirror_coeff, grad_actor, st);
  if (rc != OLY_OK) return rc;
  rc = upd_network(1, B, in_dim, 1, parts_critic, obs, NULL, action, adv, ret, old_mu, idx, &cn, sd, log_sd, old_sd,
                   old_log_sd, NULL, NULL, clip, vf_coeff, 0.0f, grad_critic, st);
  if (rc != OLY_OK) return rc;
  float ent = 0.0f;
  for (int j = 0; j < act_dim; ++j) ent += (0.5f + 0.9189385332046727f) + log_sd[j];
  scal_out[0] = -st[0] / B;
  scal_out[1] = -(st[5] * (double)ent) / ((double)B * act_dim);
  scal_out[2] = (double)vf_coeff * st[4] / B;
  scal_out[3] = st[1] / B;
  scal_out[4] = mir_obs ? st[3] / ((double)B * act_dim) : 0.0;
  scal_out[5] = st[2] / B;
  return OLY_OK;
}

/* oly_ppo_adam_step's twin for ONE network: torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step
 * (rl/algos/ppo.py:399-410; amsgrad off, no weight decay) on flat buffers, float32 per element in the kernel's
 * order; the squared norm as the kernels sum it: blocks of 256 elements, lane l squares elements 4 l .. 4 l + 3 in
 * order, a 64-lane tree per block; the block partials b = l, l + 64, ... added in order into 64 sums, those by the
 * same tree (fp64).                                                                     */
int oly_ppo_adam_step_cpu(int n, int step, float lr, float beta1, float beta2, float eps, float max_norm,
                          float* param, const float* grad, float* exp_avg, float* exp_avg_sq) {
  if (n <= 0 || step <= 0) return OLY_EINVAL;
  const int blocks = (n + 255) / 256;
  double part[64];                 /* lane l of the stepping kernel adds block partials l, l + 64, ... in order */
  for (int l = 0; l < 64; ++l) part[l] = 0.0;
  for (int b = 0; b < blocks; ++b) {
    double v[64];                  /* a block: 256 elements, lane l squares elements 4 l .. 4 l + 3 in order */
    for (int l = 0; l < 64; ++l) {
      double sq = 0.0;
      for (int i = 0; i < 4; ++i) {
        const long e = 256L * b + 4 * l + i;
        if (e < n) {
          const double g = (double)grad[e];
          sq += g * g;
        }
      }
      v[l] = sq;
    }
    for (int off = 32; off > 0; off >>= 1)
      for (int l = 0; l < off; ++l) v[l] += v[l + off];
    part[b & 63] += v[0];
  }
  for (int off = 32; off > 0; off >>= 1)          /* then the 64-lane tree */
    for (int l = 0; l < off; ++l) part[l] += part[l + off];
  const double ss = part[0];
  const float norm = (float)sqrt(ss);
  const float coef = fminf(max_norm / (norm + 1e-6f), 1.0f);
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float w1 = (float)(1.0 - (double)beta1), w2 = (float)(1.0 - (double)beta2);
  const float neg_step = (float)(-((double)lr / bc1)), bc2_sqrt = (float)sqrt(bc2);
  for (int i = 0; i < n; ++i) {
    const float g = grad[i] * coef;
    float m = exp_avg[i], v = exp_avg_sq[i];
    m = m + (g - m) * w1;
    v = v * beta2 + (w2 * g) * g;
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    param[i] = param[i] + neg_step * (m / denom);
  }
  return OLY_OK;
}
