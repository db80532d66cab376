"""ctypes mirror of include/olympic_hip.h: constants, structs and the argument lists of
every exported entry point.  Declarations only - no library is loaded here (that is
_ffi.py for the HIP library).  tests/test_abi.py checks this table against the header.
"""
import ctypes as C

OLY_OK = 0
OLY_EINVAL, OLY_ENOTCONF, OLY_EHIP, OLY_ENOMEM, OLY_ERANGE, OLY_ENODEV = -1, -2, -3, -4, -5, -6
OLY_MAX_OBS, OLY_MAX_ACT, OLY_MAX_FALL, OLY_MAX_SEQ, OLY_MAX_PERIOD = 128, 64, 16, 20, 256

REWARD_NONE, REWARD_TARGET_VELOCITY, REWARD_X_POS = 0, 1, 2
OUT_OBS_F64, OUT_CTRL_F64 = 1, 2
MODE_STANDING, MODE_FORWARD, MODE_BACKWARD, MODE_LATERAL = 1, 2, 3, 4
SCAN_RETURN, SCAN_GAE = 0, 1
SCAN_REW_F64 = 0x100
OLY_MAX_STAT_PARTS = 64
FLAG_ABSORBING, FLAG_LAST = 1, 2

i32p = C.POINTER(C.c_int32)
f64p = C.POINTER(C.c_double)
vp = C.c_void_p
PHYSICS_FN = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), vp)


class IlModel(C.Structure):
    _fields_ = [
        ("nq", C.c_int32), ("nv", C.c_int32), ("n_pos", C.c_int32), ("n_vel", C.c_int32),
        ("n_drop", C.c_int32), ("n_grf", C.c_int32), ("n_act", C.c_int32), ("nu", C.c_int32),
        ("qpos_adr", i32p), ("qvel_adr", i32p), ("act_to_ctrl", i32p),
        ("act_mean", f64p), ("act_delta", f64p), ("ctrl_lo", f64p), ("ctrl_hi", f64p),
        ("n_fall", C.c_int32), ("fall_idx", i32p), ("fall_lo", f64p), ("fall_hi", f64p),
        ("use_absorbing_states", C.c_int32), ("reward_type", C.c_int32),
        ("reward_idx", C.c_int32), ("target_velocity", C.c_double),
    ]


class A3Model(C.Structure):
    _fields_ = [
        ("nq", C.c_int32), ("nv", C.c_int32), ("nu", C.c_int32), ("period", C.c_int32),
        ("delay_frames", C.c_int32), ("target_radius", C.c_double), ("mass", C.c_double),
        ("goal_height_ref", C.c_double), ("goal_speed_ref", C.c_double),
        ("clock_lut", f64p), ("motor_offset", f64p), ("gear", f64p),
    ]


A3_INPUT_FIELDS = ["qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel", "rf_vel",
                   "root_pos", "root_quat", "head_pos", "grf_l", "grf_r", "min_z", "n_r", "n_l", "bad"]
A3_STATE_FIELDS = ["phase", "t1", "t2", "reached_frames", "target_reached", "mode", "seq_len",
                   "sequence", "goal"]


class A3Inputs(C.Structure):
    _fields_ = [(n, vp) for n in A3_INPUT_FIELDS]


class A3State(C.Structure):
    _fields_ = [(n, vp) for n in A3_STATE_FIELDS]


A3_BLOCK_F64 = ["qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel", "rf_vel", "root_pos", "root_quat",
                "head_pos"]
A3_BLOCK_TAIL = ["ncon", "geom1", "geom2", "force6", "cpos_z"]


class A3Blocks(C.Structure):
    """oly_a3_blocks: [K,N,...] device blocks of K consecutive physics readbacks."""
    _fields_ = [("K", C.c_int32), ("C", C.c_int32)] + [(n, vp) for n in A3_BLOCK_F64 + A3_BLOCK_TAIL]


class ContactRecord(C.Structure):
    """oly_contact_record (64 bytes): one used contact slot of the compact (CSR) contact form."""
    _fields_ = [("geom1", C.c_int32), ("geom2", C.c_int32), ("force6", C.c_double * 6), ("pos_z", C.c_double)]


class A3ResetRecord(C.Structure):
    """oly_a3_reset_record: one pre-drawn WalkingTask.reset (656 bytes)."""
    _fields_ = [("mode", C.c_int32), ("phase", C.c_int32), ("seq_len", C.c_int32), ("pad", C.c_int32),
                ("seq", (C.c_double * 4) * 20)]


class A3Rollout(C.Structure):
    """oly_a3_rollout: policy outputs, rollout-buffer rows, side list, reset pool, device counters."""
    _fields_ = [("T", C.c_int32), ("max_traj_len", C.c_int32), ("deterministic", C.c_int32), ("pad0", C.c_int32),
                ("mu", vp), ("value", vp), ("scale", vp), ("eps", vp), ("state", vp), ("pd_target", vp),
                ("buf_states", vp), ("buf_actions", vp), ("buf_rewards", vp), ("buf_values", vp), ("buf_flags", vp),
                ("buf_rew6", vp), ("traj_len", vp),
                ("side_slots", C.c_int32), ("pad1", C.c_int32), ("side_obs", vp), ("side_t", vp), ("side_count", vp),
                ("pool_depth", C.c_int32), ("pad2", C.c_int32), ("pool", vp), ("pool_count", vp),
                ("ctr", vp), ("buf_mu", vp)]


VSTEP_RESET_ALL = 1

# oly_a3_readback: per-env slots of the pinned staging (11 double arrays, ncon/geom1/geom2 int32, force6, cpos_z)
A3_READBACK_FIELDS = (("qpos", C.c_double), ("qvel", C.c_double), ("act_len", C.c_double), ("act_vel", C.c_double),
                      ("lf_pos", C.c_double), ("rf_pos", C.c_double), ("lf_vel", C.c_double), ("rf_vel", C.c_double),
                      ("root_pos", C.c_double), ("root_quat", C.c_double), ("head_pos", C.c_double),
                      ("ncon", C.c_int32), ("geom1", C.c_int32), ("geom2", C.c_int32), ("force6", C.c_double),
                      ("cpos_z", C.c_double))


class A3Readback(C.Structure):
    _fields_ = [(n, C.POINTER(t)) for n, t in A3_READBACK_FIELDS]


class IlContacts(C.Structure):
    _fields_ = [("W", C.c_int), ("C", C.c_int), ("ncon", C.POINTER(C.c_int32)), ("ncon_stride", C.c_long),
                ("geom1", C.POINTER(C.c_int32)), ("geom2", C.POINTER(C.c_int32)), ("geom_stride", C.c_long),
                ("force6", C.POINTER(C.c_double)), ("force_stride", C.c_long)]


PHYSICS_CONTACTS_FN = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(IlContacts), vp)
A3_PHYSICS_FN = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_double), C.POINTER(A3Readback), vp)


# name -> (restype, argtypes); device/host pointers are void*.
STD_SCALAR, STD_PER_DIM, STD_FULL = 0, 1, 2
ABI_VERSION = 7          # OLY_ABI_VERSION of include/olympic_hip.h this table mirrors


class AdamNet(C.Structure):
    """oly_adam_net: one network's flat optimiser buffers."""
    _fields_ = [("param", vp), ("grad", vp), ("exp_avg", vp), ("exp_avg_sq", vp), ("packed", vp), ("in_mean", vp),
                ("in_std", vp), ("out_dim", C.c_int32), ("pad", C.c_int32)]


class PPOAdam(C.Structure):
    """oly_ppo_adam (K14's optimiser half): clip_grad_norm_ + Adam.step + re-pack for actor and critic."""
    _fields_ = [("in_dim", C.c_int32), ("step", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("max_grad_norm", C.c_float), ("norm_ready", C.c_int32), ("net", AdamNet * 2), ("ws", vp)]


class PPOUpdate(C.Structure):
    """oly_ppo_update (K14): one PPO minibatch update's gradient call."""
    _fields_ = [("B", C.c_int32), ("in_dim", C.c_int32), ("act_dim", C.c_int32),
                ("parts_actor", C.c_int32), ("parts_critic", C.c_int32),
                ("normalize_actor", C.c_int32), ("normalize_critic", C.c_int32), ("pad0", C.c_int32),
                ("obs", vp), ("mir_obs", vp), ("action", vp), ("adv", vp), ("ret", vp), ("old_mu", vp), ("idx", vp),
                ("packed_actor", vp), ("packed_critic", vp),
                ("sd", vp), ("log_sd", vp), ("old_sd", vp), ("old_log_sd", vp),
                ("act_src", vp), ("act_sign", vp),
                ("clip", C.c_float), ("vf_coeff", C.c_float), ("mirror_coeff", C.c_float), ("pad1", C.c_int32),
                ("grad_actor", vp), ("grad_critic", vp), ("scal_out", vp), ("ws", vp), ("ws_floats", C.c_int64), ("gnorm_ws", vp)]

SIGNATURES = {
    "oly_strerror": (C.c_char_p, [C.c_int]),
    "oly_last_error": (C.c_char_p, [vp]),
    "oly_version": (C.c_char_p, []),
    "oly_abi_version": (C.c_int, []),
    "oly_create": (C.c_int, [C.POINTER(vp), C.c_int]),
    "oly_destroy": (None, [vp]),
    "oly_il_configure": (C.c_int, [vp, C.POINTER(IlModel)]),
    "oly_il_obs_dim": (C.c_int, [vp]),
    "oly_il_step": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                              C.c_int, vp]),
    "oly_il_ctrl": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, vp]),
    "oly_batcher_create": (C.c_int, [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_double, vp, vp]),
    "oly_batcher_destroy": (None, [vp]),
    "oly_batcher_qpos": (C.POINTER(C.c_double), [vp]),
    "oly_batcher_qvel": (C.POINTER(C.c_double), [vp]),
    "oly_batcher_prev": (vp, [vp]),
    "oly_batcher_set_prev": (C.c_int, [vp, vp, vp]),
    "oly_batcher_step": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, vp]),
    "oly_batcher_last_timing": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "oly_batcher_set_mapped": (C.c_int, [vp, C.c_int]),
    "oly_batcher_enable_contacts": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "oly_batcher_enable_contacts_packed": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "oly_il_grf_window": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "oly_a3_batcher_create": (C.c_int, [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "oly_a3_batcher_destroy": (None, [vp]),
    "oly_a3_batcher_slots": (C.c_int, [vp, C.c_int, C.POINTER(A3Readback)]),
    "oly_a3_batcher_upload": (C.c_int, [vp, vp]),
    "oly_a3_batcher_step": (C.c_int, [vp, vp, C.POINTER(A3State), vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "oly_a3_batcher_last_timing": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "oly_a3_batcher_set_mapped": (C.c_int, [vp, C.c_int]),
    "oly_a3_batcher_set_compact": (C.c_int, [vp, C.c_int]),
    "oly_traj_upload": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
    "oly_traj_reset": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "oly_traj_next": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "oly_traj_euler": (C.c_int, [vp, C.c_int, C.c_int, C.c_double, vp, vp, vp]),
    "oly_expert_rows": (C.c_int64, [vp]),
    "oly_expert_gather": (C.c_int, [vp, C.c_int64, vp, C.c_int, vp, vp, vp, vp]),
    "oly_expert_dataset": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp]),
    "oly_contact_configure": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]),
    "oly_contact_reduce": (C.c_int, [vp, C.c_int, C.c_int] + [vp] * 13 + [vp]),
    "oly_contact_reduce_csr": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int64] + [vp] * 7),
    "oly_a3_configure": (C.c_int, [vp, C.POINTER(A3Model)]),
    "oly_a3_step": (C.c_int, [vp, C.c_int, C.POINTER(A3Inputs), C.POINTER(A3State), vp, vp, vp, vp,
                              C.c_int, vp]),
    "oly_a3_vec_ctr_len": (C.c_int, [C.c_int]),
    "oly_a3_vec_step": (C.c_int, [vp, C.c_int, C.POINTER(A3Blocks), C.POINTER(A3State), C.POINTER(A3Rollout), C.c_int, vp]),
    "oly_a3_pd_target": (C.c_int, [vp, C.c_int, vp, vp, vp]),
    "oly_a3_pd_torque": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "oly_mlp_packed_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "oly_mlp_pack": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "oly_a3_rollout_persistent": (C.c_int, [vp, C.c_int, C.POINTER(A3Blocks), C.POINTER(A3State), C.POINTER(A3Rollout), C.c_int,
                                            vp, C.c_int, vp, C.c_int, vp, vp, vp]),
    "oly_mlp_forward2": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp]),
    "oly_return_scan": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                  vp, vp, vp, vp, vp, vp, vp]),
    "oly_return_scan_stats": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                        vp, vp, vp, vp, vp, vp, vp, vp]),
    "oly_adv_stats": (C.c_int, [vp, C.c_int64, vp, vp, vp]),
    "oly_adv_normalize_parts": (C.c_int, [vp, C.c_int64, vp, vp, C.c_int, C.c_int, C.c_double, vp]),
    "oly_adv_normalize": (C.c_int, [vp, C.c_int64, vp, vp, C.c_int, C.c_double, vp]),
    "oly_col_stats": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    "oly_disc_standardize": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "oly_disc_reparam": (C.c_int, [vp, C.c_int64, vp, vp, vp, vp, vp]),
    "oly_disc_reward": (C.c_int, [vp, C.c_int64, vp, vp, vp]),
    "oly_disc_packed_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "oly_disc_pack": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int] + [vp] * 12),
    "oly_disc_forward": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int] + [vp] * 12),
    "oly_disc_reward_step": (C.c_int, [vp, C.c_int64, C.c_int, vp, vp, C.c_int] + [vp] * 8),
    "oly_grf_configure": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp]),
    "oly_il_ground_forces": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "oly_rollout_cuts": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "oly_obs_filter": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, C.c_double, C.c_double, vp, vp]),
    "oly_signed_perm": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "oly_mirror_loss": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "oly_ppo_loss": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp,
                               C.c_float, C.c_float, vp, vp, vp, vp, vp]),
    "oly_ppo_update_grad_floats": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "oly_ppo_update_ws_floats": (C.c_int64, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "oly_ppo_update_grads": (C.c_int, [vp, C.POINTER(PPOUpdate), vp]),
    "oly_ppo_adam_step": (C.c_int, [vp, C.POINTER(PPOAdam), vp]),
    "oly_ppo_update_epoch": (C.c_int, [vp, C.POINTER(PPOUpdate), C.POINTER(PPOAdam), vp, C.c_int, vp, vp]),
    "oly_event_create": (C.c_int, [C.POINTER(vp)]),
    "oly_event_destroy": (C.c_int, [vp]),
    "oly_event_record": (C.c_int, [vp, vp]),
    "oly_event_sync": (C.c_int, [vp]),
    "oly_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    "oly_stream_sync": (C.c_int, [vp]),
}
