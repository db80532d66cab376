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
