"""Shared helpers for the parity tests (input reshaping, ulp distance, synthetic batches)."""
import numpy as np

from olympic_hip.synthetic import h1_rows_from_full, h1_synthetic_block  # noqa: F401


def ulp_diff(a, b):
    """Distance in units of the spacing of b (works for float32 or float64 arrays)."""
    a, b = np.asarray(a), np.asarray(b)
    sp = np.spacing(np.maximum(np.abs(b), np.finfo(b.dtype).tiny).astype(b.dtype))
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / sp.astype(np.float64)


def a3_fixture_arrays(g, k):
    """(state at reset, inputs of step k) of tests/golden/a3_task.npz as oracle/kernel dicts."""
    E = g["phase"].shape[0]
    st = dict(phase=g["phase0"].astype(np.int32).copy(), t1=g["t1_0"].astype(np.int32).copy(),
              t2=g["t2_0"].astype(np.int32).copy(), reached_frames=np.zeros(E, np.int32),
              target_reached=np.zeros(E, np.uint8), mode=g["mode"].astype(np.int32).copy(),
              seq_len=g["seq_len"].astype(np.int32).copy(),
              sequence=np.ascontiguousarray(g["sequence"], dtype=np.float64),
              goal=np.zeros((E, 8)))
    inp = {n: np.ascontiguousarray(g[n][:, k]) for n in
           ("qpos", "qvel", "act_len", "act_vel", "lf_pos", "rf_pos", "lf_vel", "rf_vel", "root_pos",
            "root_quat", "head_pos")}
    return st, inp




def ppo_update_arrays(g):
    """numpy forward of the fixture's actor / old actor / critic (relu MLPs) on its obs, and the
    mirrored forward: everything oly_ppo_loss / oly_mirror_loss take as inputs."""
    from olympic_hip.wrappers import _signed_perm

    def mlp(x, tag, layers, head):
        for i in range(2):
            x = np.maximum(x @ g[f"{tag}.{layers}.{i}.weight"].T + g[f"{tag}.{layers}.{i}.bias"], 0)
        return (x @ g[f"{tag}.{head}.weight"].T + g[f"{tag}.{head}.bias"]).astype(np.float32)
    obs = g["obs"].astype(np.float32)
    o_src, o_sgn = _signed_perm(g["mirrored_obs"].tolist())
    a_src, a_sgn = _signed_perm(g["mirrored_acts"].tolist())
    mobs = obs[:, o_src] * o_sgn
    for i in (31, 32):                                     # mirror_clock_observation: sin(arcsin(x) + pi)
        mobs[:, i] = np.sin(np.arcsin(mobs[:, i]) + np.float32(np.pi))
    return dict(mu=mlp(obs, "pi", "actor_layers", "means"), old_mu=mlp(obs, "old", "actor_layers", "means"),
                value=mlp(obs, "vf", "critic_layers", "network_out").reshape(-1),
                mir=mlp(mobs.astype(np.float32), "pi", "actor_layers", "means"),
                act_src=a_src.astype(np.int32), act_sign=a_sgn.astype(np.float32),
                obs_src=o_src.astype(np.int32), obs_sign=o_sgn.astype(np.float32),
                std=np.float32(g["fixed_std"]), action=g["act"].astype(np.float32),
                adv=g["adv"].reshape(-1).astype(np.float32), ret=g["ret"].reshape(-1).astype(np.float32),
                clip=float(g["clip"]))
