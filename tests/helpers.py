"""Shared helpers for the parity tests (input reshaping, ulp distance, synthetic batches)."""
import numpy as np


def ulp_diff(a, b):
    """Distance in units of the spacing of b (works for float32 or float64 arrays)."""
    a, b = np.asarray(a), np.asarray(b)
    sp = np.spacing(np.maximum(np.abs(b), np.finfo(b.dtype).tiny).astype(b.dtype))
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / sp.astype(np.float64)


def h1_rows_from_full(spec, full):
    """Spec-ordered rows [M, n_pos+n_vel] -> (qpos [M,nq], qvel [M,nv]) in MuJoCo address
    order, i.e. the inverse of ObservationHelper's gather."""
    full = np.asarray(full, dtype=np.float64)
    M = len(full)
    qpos = np.zeros((M, spec.nq))
    qvel = np.zeros((M, spec.nv))
    qpos[:, spec.qpos_adr] = full[:, :spec.n_pos]
    qvel[:, spec.qvel_adr] = full[:, spec.n_pos:spec.n_pos + spec.n_vel]
    return qpos, qvel


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


def h1_synthetic_block(spec, T, N, seed=1234, fall_frac="bench"):
    """SURVEY 8(d) config-2 generator: qpos,qvel [T,N,17] f64, action [T,N,11] f32 U(-1,1).
    Joint angles U(range), pelvis pose spread so that a few percent of rows are fallen."""
    rng = np.random.default_rng(seed)
    R = T * N
    lo, hi = spec.joint_lo, spec.joint_hi
    full = np.empty((R, spec.n_pos + spec.n_vel))
    full[:, 0:2] = rng.uniform(-5, 5, (R, 2))
    if fall_frac == "bench":
        full[:, 2] = rng.normal(-0.1, 0.06, R)
        full[:, 3:6] = rng.normal(0, 0.08, (R, 3))
    else:
        full[:, 2] = rng.uniform(-0.4, 0.2, R)
        full[:, 3:6] = rng.uniform(-0.6, 0.6, (R, 3))
    full[:, 6:spec.n_pos] = rng.uniform(lo[6:], hi[6:], (R, spec.n_pos - 6))
    full[:, spec.n_pos:] = rng.normal(0.0, 1.5, (R, spec.n_vel))
    full[:, spec.n_pos] = rng.normal(1.25, 0.5, R)
    qpos, qvel = h1_rows_from_full(spec, full)
    action = np.random.default_rng(seed + 1).uniform(-1, 1, (R, spec.n_act)).astype(np.float32)
    return (qpos.reshape(T, N, -1), qvel.reshape(T, N, -1), action.reshape(T, N, -1))
