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


