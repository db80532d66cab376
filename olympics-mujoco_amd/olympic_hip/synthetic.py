"""Seeded synthetic physics batches: stand-ins for what the host
MuJoCo batcher would hand over.  Used by bench.py, the tests and the synthetic physics
backend; physics itself (mj_step) stays on the host and is out of scope."""
import numpy as np


def h1_rows_from_full(spec, full):
    """Spec-ordered rows [M, n_pos+n_vel] -> (qpos [M,nq], qvel [M,nv]) in MuJoCo address order."""
    full = np.asarray(full, dtype=np.float64)
    M = len(full)
    qpos = np.zeros((M, spec.nq))
    qvel = np.zeros((M, spec.nv))
    qpos[:, spec.qpos_adr] = full[:, :spec.n_pos]
    qvel[:, spec.qvel_adr] = full[:, spec.n_pos:spec.n_pos + spec.n_vel]
    return qpos, qvel


def h1_synthetic_block(spec, T, N, seed=1234, fall_frac="bench"):
    """Config 2: qpos,qvel [T,N,nq] f64, action [T,N,n_act] f32 ~ U(-1,1) (seed+1).
    Joint angles U(joint range), dq ~ N(0,1.5), x-velocity ~ N(1.25,0.5); the pelvis pose
    is spread so that a few percent of the rows are fallen ("bench") or most are ("wide")."""
    rng = np.random.default_rng(seed)
    R = T * N
    lo, hi = spec.joint_lo, spec.joint_hi
    n_pos = spec.n_pos
    full = np.empty((R, n_pos + spec.n_vel))
    full[:, 0:2] = rng.uniform(-5, 5, (R, 2))
    if fall_frac == "bench":
        full[:, 2] = rng.normal(-0.1, 0.085, R)          # with the eulers below: ~3 % fallen rows
        full[:, 3:6] = rng.normal(0, 0.105, (R, 3))
    else:
        full[:, 2] = rng.uniform(-0.4, 0.2, R)
        full[:, 3:6] = rng.uniform(-0.6, 0.6, (R, 3))
    full[:, 6:n_pos] = rng.uniform(lo[6:], hi[6:], (R, n_pos - 6))
    full[:, n_pos:] = rng.normal(0.0, 1.5, (R, spec.n_vel))
    full[:, n_pos] = rng.normal(1.25, 0.5, R)
    qpos, qvel = h1_rows_from_full(spec, full)
    action = np.random.default_rng(seed + 1).uniform(-1, 1, (R, spec.n_act)).astype(np.float32)
    return qpos.reshape(T, N, -1), qvel.reshape(T, N, -1), action.reshape(T, N, -1)


A3_GEOM_BODYID = np.array([0, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10], np.int32)   # geoms 7,8 right foot; 11,12 left foot
A3_FLOOR_BODY, A3_RFOOT_BODY, A3_LFOOT_BODY = 0, 7, 10


def a3_synthetic_blocks(N, K, seed=1, C=16, p_bad=1.0 / 300, p_low=0.0):
    """Config 3: K consecutive synthetic physics readbacks for N StickFigureA3 environments
    as numpy arrays [K,N,...] (oly_a3_blocks fields): qpos / qvel ~ N(0,1), feet around a footstep
    grid, root 0.8 m above them, unit root quaternions, contact lists with ncon ~ Poisson(4) clipped to
    C whose slots are all (floor, foot) pairs except that with probability p_bad per (step, env) one
    slot is a non-foot geom (check_bad_collisions -> done); p_low: probability of a root height that
    trips the done height test."""
    rng = np.random.default_rng(seed)
    rnd = lambda *s: rng.normal(0, 1, s)
    base = np.zeros((N, 3)) + np.array([0.3, 0.15, 0.0])
    b = dict(qpos=rnd(K, N, 25), qvel=rnd(K, N, 24), act_len=rnd(K, N, 12), act_vel=rnd(K, N, 12),
             lf_pos=base + 0.1 * rnd(K, N, 3), rf_pos=base + 0.3 * rnd(K, N, 3), lf_vel=0.2 * rnd(K, N, 3),
             rf_vel=0.2 * rnd(K, N, 3), root_quat=rnd(K, N, 4), force6=100 * rnd(K, N, C, 6),
             cpos_z=0.01 * rnd(K, N, C))
    b["lf_pos"][..., 2] = np.abs(b["lf_pos"][..., 2]) * 0.3
    b["rf_pos"][..., 2] = np.abs(b["rf_pos"][..., 2]) * 0.1
    foot = np.minimum(b["lf_pos"][..., 2], b["rf_pos"][..., 2])
    low = rng.uniform(size=(K, N)) < p_low
    b["root_pos"] = base + 0.05 * rnd(K, N, 3)
    b["root_pos"][..., 2] = foot + np.where(low, 0.5, 0.8) + 0.02 * np.abs(rnd(K, N))
    b["head_pos"] = b["root_pos"] + np.array([0, 0, 0.4]) + 0.05 * rnd(K, N, 3)
    b["root_quat"] /= np.linalg.norm(b["root_quat"], axis=-1, keepdims=True)
    b["ncon"] = np.minimum(rng.poisson(4, (K, N)), C).astype(np.int32)
    b["geom1"] = np.zeros((K, N, C), np.int32)
    b["geom2"] = rng.choice([7, 8, 11, 12], (K, N, C)).astype(np.int32)
    hit = (rng.uniform(size=(K, N)) < p_bad) & (b["ncon"] > 0)
    slot = rng.integers(0, 1 << 30, (K, N)) % np.maximum(b["ncon"], 1)
    kk, nn = np.nonzero(hit)
    b["geom2"][kk, nn, slot[kk, nn]] = 9
    return {k: np.ascontiguousarray(v) for k, v in b.items()}
