/*
 * oly_oracle.h - CPU restatement ("oracle") of the reference's hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, load or call this library; the product
 * (libolympic_hip.so and the olympic_hip host package) never does and fails loudly
 * when its HIP library is missing.
 *
 * Every function is the plain-C, one-row-at-a-time restatement of the reference Python
 * named in its comment (file:line under /root/reference) and takes the same arguments
 * as its oly_* twin in include/olympic_hip.h, with HOST pointers, an explicit model
 * struct instead of a ctx, and no stream.
 *
 * Pinning (see DESIGN.md "Oracle"): checked in tests/test_oracle_golden.py against the
 * golden vectors of tests/golden/, which were produced by executing the reference's own
 * functions (tests/golden/gen_golden.py).  Not pinned by any reference output, because
 * the arithmetic lives in packages absent from /root/reference and from this image:
 *   - mushroom-rl (>=1.10, unpinned in requirements.txt): ObservationHelper._build_obs
 *     (the qpos/qvel gather order), compute_gae, RunningAveragedWindow;
 *   - transforms3d (unpinned): quat2euler / euler2quat / quat2mat / mat2euler / compose
 *     used by StickFigureA3.get_obs and WalkingTask.update_goal_steps;
 *   - mujoco 2.3.6: everything upstream of the arrays these functions receive.
 * Those pieces are restated from the packages' published definitions: PARITY UNPINNED.
 */
#ifndef OLY_ORACLE_H
#define OLY_ORACLE_H

#include "../include/olympic_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

int oly_il_step_cpu(const oly_il_model* m, int T, int N, const double* qpos, const double* qvel,
                    const float* action, const double* grf_mean, double* prev_inout, void* obs,
                    float* reward, uint8_t* absorbing, uint8_t* fall_code, void* ctrl,
                    int out_flags, double* reward_f64 /* [T,N] or NULL: un-narrowed reward */);

int oly_traj_reset_cpu(int n_keys, int n_traj, int len, const double* table, int N,
                       const int32_t* traj_no, const int32_t* step, int32_t* cur_traj,
                       int32_t* cur_step, double* origin, double* sample);
int oly_traj_next_cpu(int n_keys, int n_traj, int len, const double* table, int N,
                      const uint8_t* active, const int32_t* cur_traj, int32_t* cur_step,
                      const double* origin, double* sample, uint8_t* at_end);
int oly_traj_euler_cpu(int n_keys, int N, int n_qpos, double dt, const double* curr_qpos,
                       double* sample);

int oly_contact_reduce_cpu(int ngeom, const int32_t* geom_bodyid, int floor_body, int rfoot_body,
                           int lfoot_body, int N, int C, const int32_t* ncon,
                           const int32_t* geom1, const int32_t* geom2, const double* force6,
                           const double* pos_z, int32_t* n_r, int32_t* n_l, int32_t* idx_r,
                           int32_t* idx_l, double* grf_r, double* grf_l, double* min_z,
                           uint8_t* bad);

int oly_a3_step_cpu(const oly_a3_model* m, int N, const oly_a3_inputs* in, const oly_a3_state* st,
                    void* obs, float* rew6, float* reward, uint8_t* done, int out_flags,
                    double* rew6_f64 /* [N,6] or NULL */, double* reward_f64 /* [N] or NULL */);
int oly_a3_pd_target_cpu(const oly_a3_model* m, int N, const float* action, double* target);
int oly_a3_pd_torque_cpu(const oly_a3_model* m, int N, const double* kp, const double* kd,
                         const double* target, const double* act_len, const double* act_vel,
                         double* tau);

int oly_a3_vec_step_cpu(const oly_a3_model* m, int ngeom, const int32_t* geom_bodyid, int floor_body,
                        int rfoot_body, int lfoot_body, int N, const oly_a3_blocks* b,
                        const oly_a3_state* st, const oly_a3_rollout* ro, int flags);

int oly_mlp_forward_cpu(int N, int in_dim, int out_dim, const float* x, const float* w1, const float* b1,
                        const float* w2, const float* b2, const float* w3, const float* b3,
                        const float* in_mean, const float* in_std, float* y);

int oly_return_scan_cpu(int mode, int T, int N, double gamma, double lam, const float* rew,
                        const float* val, const float* next_val, const uint8_t* flags, float* ret,
                        float* adv);

int oly_return_scan_r64_cpu(int T, int N, double gamma, const double* rew, const float* val,
                            const float* next_val, const uint8_t* flags, float* ret, float* adv);

int oly_adv_stats_cpu(int64_t n, const float* x, double* stats3_out);
int oly_adv_normalize_parts_cpu(int64_t n, float* x, const double* parts3, int parts, int ddof, double eps);
int oly_adv_normalize_cpu(int64_t n, float* x, const double* stats3, int ddof, double eps);
int oly_col_stats_cpu(int B, int D, const float* x, double* colstats, int accumulate);

int oly_disc_standardize_cpu(int B, int Dx, int D, const float* x, const int32_t* mask,
                             const double* mean, const double* std, float* out);
int oly_disc_reparam_cpu(int64_t n, const float* mu, const float* logvar, const float* eps,
                         float* z);
int oly_disc_reward_cpu(int64_t B, const float* logits, float* reward);

float oly_exp32_cpu(float x);
int oly_disc_forward_cpu(int64_t B, int Dx, int D, const float* x, const int32_t* mask, const double* mean,
                         const double* sd, const double* colstats, const float* enc_w0, const float* enc_b0, const float* enc_w1,
                         const float* enc_b1, const float* mu_w, const float* mu_b, const float* lv_w,
                         const float* lv_b, const float* dec_w, const float* dec_b, const float* eps,
                         float* reward, float* logits, float* mu_out, float* logvar_out);

int oly_il_ground_forces_cpu(int ngeom, const int32_t* geom_group, int n_pairs, const int32_t* pair_a,
                             const int32_t* pair_b, int W, int N, int C, const int32_t* ncon,
                             const int32_t* geom1, const int32_t* geom2, const double* force6,
                             double* grf_step, double* grf_mean, uint8_t* overflow);
int oly_rollout_cuts_cpu(int N, int max_traj_len, int last_step, const uint8_t* done, int32_t* traj_len,
                         uint8_t* flags, int32_t* n_cut);
int oly_obs_filter_cpu(int B, int D, const float* x, const double* mean, const double* var,
                       double eps, double clip, float* out);
int oly_signed_perm_cpu(int B, int D, const float* x, const int32_t* src, const float* sign,
                        float* out);
int oly_mirror_loss_cpu(int B, int A, const float* det, const float* mir, const int32_t* src,
                        const float* sign, double* loss_out, float* grad_det, float* grad_mir);
int oly_ppo_loss_cpu(int B, int A, const float* mu, const float* std, int std_mode,
                     const float* old_mu, const float* old_std, int old_std_mode,
                     const float* action, const float* adv, const float* ret, const float* value,
                     float clip, float vf_coeff, double* scal_out, float* grad_mu, float* grad_std,
                     float* grad_value);

/* K14 twin: gradients of one PPO minibatch update (rl/algos/ppo.py:232-282,396-410); *_wb = {w1,b1,w2,b2,w3,b3};
 * grad_* flat in parameter order; scal_out[6] = actor, entropy_penalty, critic, approx_kl, mirror, clip_fraction. */
int oly_ppo_update_cpu(int B, int in_dim, int act_dim, int parts_actor, int parts_critic, const float* obs,
                       const float* mir_obs, const float* action, const float* adv, const float* ret,
                       const float* old_mu, const int32_t* idx, const float* const* actor_wb,
                       const float* a_mean, const float* a_std, const float* const* critic_wb,
                       const float* c_mean, const float* c_std, const float* sd, const float* log_sd,
                       const float* old_sd, const float* old_log_sd, const int32_t* act_src,
                       const float* act_sign, float clip, float vf_coeff, float mirror_coeff, float* grad_actor,
                       float* grad_critic, double* scal_out);

int oly_ppo_adam_step_cpu(int n, int step, float lr, float beta1, float beta2, float eps, float max_norm,
                          float* param, const float* grad, float* exp_avg, float* exp_avg_sq);

/* OpenMP-parallel variant used only by bench.py's cpu_baseline leg (threads <= 0: all). */
int oly_il_step_cpu_mt(const oly_il_model* m, int T, int N, const double* qpos,
                       const double* qvel, const float* action, double* prev_inout, void* obs,
                       float* reward, uint8_t* absorbing, void* ctrl, int out_flags, int threads);
int oly_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
