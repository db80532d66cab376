"""Typed torch front-end over the C ABI: one method per entry point of
include/olympic_hip.h.  Tensors are validated (device, dtype, contiguity, shape) on the
host before any kernel is launched - a wrong shape must never reach the GPU.  PyTorch is
only the allocator/stream provider here; every computation is a HIP kernel of
libolympic_hip.so running on torch's current stream.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._ffi import Context, OlyError, ptr


def _req(t, name, shape, dtype, device, optional=False):
    if t is None:
        if optional:
            return None
        raise OlyError(f"{name}: required tensor is None")
    if not isinstance(t, torch.Tensor):
        raise OlyError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if t.device != device:
        raise OlyError(f"{name}: tensor on {t.device}, engine on {device}")
    if t.dtype != dtype:
        raise OlyError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if tuple(t.shape) != tuple(shape):
        raise OlyError(f"{name}: shape {tuple(t.shape)}, expected {tuple(shape)}")
    if not t.is_contiguous():
        raise OlyError(f"{name}: tensor must be contiguous")
    return t


class Engine:
    """HIP hot-path engine bound to one device."""

    def __init__(self, device=None):
        self.ctx = Context(device)
        self.device = self.ctx.device
        self.il_spec = None
        self.a3_spec = None
        self.traj_shape = None
        self.contact_ok = False

    # -------------------------------------------------------------- helpers
    def _new(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _s(self):
        return self.ctx.stream()

    def _out(self, out, key, shape, dtype):
        """Caller-supplied output buffer `out[key]` (validated) or a fresh one; nothing is allocated
        when the caller passes its own buffers."""
        t = out.get(key) if out else None
        if t is None:
            t = self._new(shape, dtype)
        return _req(t, key, shape, dtype, self.device)

    # -------------------------------------------------------------- K1 / K5
    def il_configure(self, spec):
        self.ctx.call("oly_il_configure", C.byref(spec.to_c()))
        self.il_spec = spec
        return self

    def il_step(self, qpos, qvel, action, prev_in, prev_out=None, grf_mean=None, out=None,
                obs_f64=False, ctrl_f64=False, want_fall_code=True, want_ctrl=True):
        """qpos [T,N,nq] f64, qvel [T,N,nv] f64, action [T,N,n_act] f32 (or None).
        Returns dict(obs, reward, absorbing, fall_code, ctrl, prev)."""
        sp = self.il_spec
        if sp is None:
            raise OlyError("il_step before il_configure")
        if qpos.dim() != 3:
            raise OlyError(f"qpos: expected [T,N,nq], got {tuple(qpos.shape)}")
        T, N = int(qpos.shape[0]), int(qpos.shape[1])
        dv = self.device
        _req(qpos, "qpos", (T, N, sp.nq), torch.float64, dv)
        _req(qvel, "qvel", (T, N, sp.nv), torch.float64, dv)
        want_ctrl = want_ctrl and action is not None
        _req(action, "action", (T, N, sp.n_act), torch.float32, dv, optional=True)
        _req(grf_mean, "grf_mean", (T, N, sp.n_grf), torch.float64, dv, optional=sp.n_grf == 0)
        _req(prev_in, "prev_in", (N,), torch.float64, dv)
        if prev_out is None:
            prev_out = prev_in if T == 1 else self._new((N,), torch.float64)
        _req(prev_out, "prev_out", (N,), torch.float64, dv)
        if T > 1 and prev_out.data_ptr() == prev_in.data_ptr():
            raise OlyError("prev_out must not alias prev_in when T > 1")
        out = out or {}
        od = torch.float64 if obs_f64 else torch.float32
        cd = torch.float64 if ctrl_f64 else torch.float32
        obs = self._out(out, "obs", (T, N, sp.n_obs), od)
        reward = self._out(out, "reward", (T, N), torch.float32)
        absorbing = self._out(out, "absorbing", (T, N), torch.uint8)
        code = self._out(out, "fall_code", (T, N), torch.uint8) if want_fall_code else None
        ctrl = self._out(out, "ctrl", (T, N, sp.nu), cd) if want_ctrl else None
        flags = (_abi.OUT_OBS_F64 if obs_f64 else 0) | (_abi.OUT_CTRL_F64 if ctrl_f64 else 0)
        self.ctx.call("oly_il_step", T, N, ptr(qpos), ptr(qvel), ptr(action), ptr(grf_mean), ptr(prev_in),
                      ptr(prev_out), ptr(obs), ptr(reward), ptr(absorbing), ptr(code), ptr(ctrl), flags,
                      self._s())
        return dict(obs=obs, reward=reward, absorbing=absorbing, fall_code=code, ctrl=ctrl, prev=prev_out)

    def il_step_prepare(self, qpos, qvel, action, prev_in, prev_out=None, **kw):
        """Validate once, then return a zero-argument callable that re-launches oly_il_step on
        the SAME buffers (the per-step regime of a vec env: ~2 us of host time per call instead
        of the full validation path).  Returns (call, outputs)."""
        rec = {}
        orig = self.ctx.call

        def capture(name, *args):
            rec["name"], rec["args"] = name, args
        self.ctx.call = capture
        try:
            outs = self.il_step(qpos, qvel, action, prev_in, prev_out, **kw)
        finally:
            self.ctx.call = orig
        from ._ffi import check, lib
        fn = getattr(lib(), rec["name"])
        h, args = self.ctx.handle, rec["args"][:-1]
        keep = (qpos, qvel, action, prev_in, outs)     # keep the tensors alive with the closure

        def call(stream=None):
            rc = fn(h, *args, stream if stream is not None else self._s())
            if rc:
                check(h, rc, rec["name"])
            return keep[-1]
        return call, outs

    # -------------------------------------------------------------- K4
    def traj_upload(self, table):
        table = np.ascontiguousarray(table, dtype=np.float64)
        if table.ndim != 3:
            raise OlyError(f"trajectory table must be [n_keys,n_traj,len], got {table.shape}")
        K, J, L = table.shape
        self.ctx.call("oly_traj_upload", K, J, L, ptr(table))
        self.traj_shape = (K, J, L)
        return self

    def traj_reset(self, traj_no, step, cur_traj=None, cur_step=None, origin=None, sample=None):
        if self.traj_shape is None:
            raise OlyError("traj_reset before traj_upload")
        K = self.traj_shape[0]
        N = int(traj_no.shape[0])
        dv = self.device
        _req(traj_no, "traj_no", (N,), torch.int32, dv)
        _req(step, "step", (N,), torch.int32, dv)
        cur_traj = _req(cur_traj if cur_traj is not None else self._new((N,), torch.int32), "cur_traj", (N,), torch.int32, dv)
        cur_step = _req(cur_step if cur_step is not None else self._new((N,), torch.int32), "cur_step", (N,), torch.int32, dv)
        origin = _req(origin if origin is not None else self._new((N, 2), torch.float64), "origin", (N, 2), torch.float64, dv)
        sample = _req(sample if sample is not None else self._new((N, K), torch.float64), "sample", (N, K), torch.float64, dv)
        self.ctx.call("oly_traj_reset", N, ptr(traj_no), ptr(step), ptr(cur_traj), ptr(cur_step), ptr(origin),
                      ptr(sample), self._s())
        return cur_traj, cur_step, origin, sample

    def traj_next(self, cur_traj, cur_step, origin, sample, active=None, at_end=None):
        if self.traj_shape is None:
            raise OlyError("traj_next before traj_upload")
        K = self.traj_shape[0]
        N = int(cur_traj.shape[0])
        dv = self.device
        _req(cur_traj, "cur_traj", (N,), torch.int32, dv)
        _req(cur_step, "cur_step", (N,), torch.int32, dv)
        _req(origin, "origin", (N, 2), torch.float64, dv)
        _req(sample, "sample", (N, K), torch.float64, dv)
        _req(active, "active", (N,), torch.uint8, dv, optional=True)
        at_end = _req(at_end if at_end is not None else self._new((N,), torch.uint8), "at_end", (N,), torch.uint8, dv)
        self.ctx.call("oly_traj_next", N, ptr(active), ptr(cur_traj), ptr(cur_step), ptr(origin), ptr(sample),
                      ptr(at_end), self._s())
        return at_end

    def traj_euler(self, n_qpos, dt, curr_qpos, sample):
        if self.traj_shape is None:
            raise OlyError("traj_euler before traj_upload")
        K = self.traj_shape[0]
        N = int(sample.shape[0])
        _req(curr_qpos, "curr_qpos", (N, n_qpos), torch.float64, self.device)
        _req(sample, "sample", (N, K), torch.float64, self.device)
        self.ctx.call("oly_traj_euler", N, int(n_qpos), C.c_double(dt), ptr(curr_qpos), ptr(sample), self._s())
        return sample

    # -------------------------------------------------------------- expert data (K4 table as the dataset)
    def expert_rows(self):
        from ._ffi import lib
        n = int(lib().oly_expert_rows(self.ctx.handle))
        if n < 0:
            raise OlyError("expert data before traj_upload")
        return n

    def expert_gather(self, idx, cols, want_next=False, check=True):
        """Rows `idx` ([B] int64 device tensor, caller-drawn) of create_dataset's states (and next_states)
        restricted to the table keys `cols` ([n_cols] int32 device tensor), as float32."""
        B, n_cols = int(idx.shape[0]), int(cols.shape[0])
        _req(idx, "idx", (B,), torch.int64, self.device)
        _req(cols, "cols", (n_cols,), torch.int32, self.device)
        if check and B:
            lo, hi = int(idx.min()), int(idx.max())
            if lo < 0 or hi >= self.expert_rows():
                raise OlyError(f"expert_gather: row index out of range [{lo}, {hi}] for {self.expert_rows()} rows")
        st = self._new((B, n_cols), torch.float32)
        nx = self._new((B, n_cols), torch.float32) if want_next else None
        self.ctx.call("oly_expert_gather", C.c_int64(B), ptr(idx), n_cols, ptr(cols), ptr(st), ptr(nx), self._s())
        return (st, nx) if want_next else st

    def expert_dataset(self, cols):
        """dict(states, next_states [rows, n_cols] f64, absorbing [rows], last [rows + 1]) on the device."""
        n_cols, rows = int(cols.shape[0]), self.expert_rows()
        _req(cols, "cols", (n_cols,), torch.int32, self.device)
        d = dict(states=self._new((rows, n_cols), torch.float64), next_states=self._new((rows, n_cols), torch.float64),
                 absorbing=self._new((rows,), torch.float64), last=self._new((rows + 1,), torch.float64))
        self.ctx.call("oly_expert_dataset", n_cols, ptr(cols), ptr(d["states"]), ptr(d["next_states"]),
                      ptr(d["absorbing"]), ptr(d["last"]), self._s())
        return d

    # -------------------------------------------------------------- K3
    def contact_configure(self, geom_bodyid, floor_body, rfoot_body, lfoot_body):
        gb = np.ascontiguousarray(geom_bodyid, dtype=np.int32)
        self.ctx.call("oly_contact_configure", len(gb), ptr(gb), int(floor_body), int(rfoot_body), int(lfoot_body))
        self.contact_ok = True
        return self

    def contact_reduce(self, ncon, geom1, geom2, force6, pos_z, want_idx=True):
        if not self.contact_ok:
            raise OlyError("contact_reduce before contact_configure")
        dv = self.device
        if geom1.dim() != 2:
            raise OlyError(f"geom1: expected [N,C], got {tuple(geom1.shape)}")
        N, Cc = int(geom1.shape[0]), int(geom1.shape[1])
        _req(ncon, "ncon", (N,), torch.int32, dv)
        _req(geom1, "geom1", (N, Cc), torch.int32, dv)
        _req(geom2, "geom2", (N, Cc), torch.int32, dv)
        _req(force6, "force6", (N, Cc, 6), torch.float64, dv)
        _req(pos_z, "pos_z", (N, Cc), torch.float64, dv)
        o = dict(n_r=self._new((N,), torch.int32), n_l=self._new((N,), torch.int32),
                 idx_r=self._new((N, Cc), torch.int32) if want_idx else None,
                 idx_l=self._new((N, Cc), torch.int32) if want_idx else None,
                 grf_r=self._new((N,), torch.float64), grf_l=self._new((N,), torch.float64),
                 min_z=self._new((N,), torch.float64), bad=self._new((N,), torch.uint8))
        self.ctx.call("oly_contact_reduce", N, Cc, ptr(ncon), ptr(geom1), ptr(geom2), ptr(force6), ptr(pos_z),
                      ptr(o["n_r"]), ptr(o["n_l"]), ptr(o["idx_r"]), ptr(o["idx_l"]), ptr(o["grf_r"]),
                      ptr(o["grf_l"]), ptr(o["min_z"]), ptr(o["bad"]), self._s())
        return o

    def contact_reduce_csr(self, ncon, coff, records, max_contacts):
        """oly_contact_reduce over compact storage: env n's contacts are records[coff[n] : coff[n] + min(ncon[n], C)];
        records: uint8 device tensor holding oly_contact_record's (64 bytes each)."""
        if not self.contact_ok:
            raise OlyError("contact_reduce_csr before contact_configure")
        dv = self.device
        N = int(ncon.shape[0])
        _req(ncon, "ncon", (N,), torch.int32, dv)
        _req(coff, "coff", (N,), torch.int32, dv)
        _req(records, "records", records.shape, torch.uint8, dv)
        if records.dim() != 1 or records.numel() % C.sizeof(_abi.ContactRecord):
            raise OlyError("records: expected a flat uint8 tensor of whole 64-byte records")
        o = dict(n_r=self._new((N,), torch.int32), n_l=self._new((N,), torch.int32),
                 grf_r=self._new((N,), torch.float64), grf_l=self._new((N,), torch.float64),
                 min_z=self._new((N,), torch.float64), bad=self._new((N,), torch.uint8))
        self.ctx.call("oly_contact_reduce_csr", N, int(max_contacts), ptr(ncon), ptr(coff), ptr(records),
                      records.numel() // C.sizeof(_abi.ContactRecord), ptr(o["n_r"]),
                      ptr(o["n_l"]), ptr(o["grf_r"]), ptr(o["grf_l"]), ptr(o["min_z"]), ptr(o["bad"]), self._s())
        return o

    # -------------------------------------------------------------- K2
    _A3_IN = dict(qpos=("nq", torch.float64), qvel=("nv", torch.float64), act_len=("nu", torch.float64),
                  act_vel=("nu", torch.float64), lf_pos=(3, torch.float64), rf_pos=(3, torch.float64),
                  lf_vel=(3, torch.float64), rf_vel=(3, torch.float64), root_pos=(3, torch.float64),
                  root_quat=(4, torch.float64), head_pos=(3, torch.float64), grf_l=(None, torch.float64),
                  grf_r=(None, torch.float64), min_z=(None, torch.float64), n_r=(None, torch.int32),
                  n_l=(None, torch.int32), bad=(None, torch.uint8))
    _A3_ST = dict(phase=(None, torch.int32), t1=(None, torch.int32), t2=(None, torch.int32),
                  reached_frames=(None, torch.int32), target_reached=(None, torch.uint8),
                  mode=(None, torch.int32), seq_len=(None, torch.int32),
                  sequence=((_abi.OLY_MAX_SEQ, 4), torch.float64), goal=(8, torch.float64))

    def a3_configure(self, spec, clock_lut):
        lut = np.ascontiguousarray(clock_lut, dtype=np.float64)
        self.ctx.call("oly_a3_configure", C.byref(spec.to_c(lut)))
        self.a3_spec = spec
        return self

    def _a3_struct(self, cls, table, tensors, N):
        sp = self.a3_spec
        st = cls()
        for name, (w, dt) in table.items():
            if isinstance(w, str):
                w = getattr(sp, w)
            shape = (N,) if w is None else ((N,) + tuple(w) if isinstance(w, tuple) else (N, w))
            t = _req(tensors.get(name), name, shape, dt, self.device)
            setattr(st, name, t.data_ptr())
        return st

    def a3_state_struct(self, state, N):
        """oly_a3_state filled with the (validated) device tensors of `state`."""
        if self.a3_spec is None:
            raise OlyError("a3 state before a3_configure")
        return self._a3_struct(_abi.A3State, self._A3_ST, state, N)

    def a3_step(self, inputs, state, obs_f64=False, out=None):
        sp = self.a3_spec
        if sp is None:
            raise OlyError("a3_step before a3_configure")
        N = int(state["phase"].shape[0])
        cin = self._a3_struct(_abi.A3Inputs, self._A3_IN, inputs, N)
        cst = self._a3_struct(_abi.A3State, self._A3_ST, state, N)
        out = out or {}
        od = torch.float64 if obs_f64 else torch.float32
        obs = self._out(out, "obs", (N, sp.n_obs), od)
        rew6 = self._out(out, "rew6", (N, 6), torch.float32)
        reward = self._out(out, "reward", (N,), torch.float32)
        done = self._out(out, "done", (N,), torch.uint8)
        self.ctx.call("oly_a3_step", N, C.byref(cin), C.byref(cst), ptr(obs), ptr(rew6), ptr(reward), ptr(done),
                      _abi.OUT_OBS_F64 if obs_f64 else 0, self._s())
        return dict(obs=obs, rew6=rew6, reward=reward, done=done)

    # -------------------------------------------------------------- K10 (one launch per vec step)
    def a3_vec_ctr_len(self, N):
        from ._ffi import lib
        return int(lib().oly_a3_vec_ctr_len(int(N)))

    def a3_vec_prepare(self, blocks, state, ro):
        """Validate every tensor of a device-resident rollout ONCE and return `launch(flags=0, mu=None,
        value=None)`, which enqueues oly_a3_vec_step on the current stream (mu / value may be swapped
        per call for an eager policy forward; everything else is fixed, so the launch is replayable
        from a HIP graph).
          blocks: dict of [K,N,...] readback tensors (oly_a3_blocks fields)
          state:  the VecA3Env task-state dict (oly_a3_state fields; mode / seq_len / sequence are
                  rewritten by device-side resets)
          ro:     dict with T, max_traj_len, deterministic, side_slots, pool_depth (ints) and the
                  tensors mu, value, scale, eps, state, pd_target, buf_states, buf_actions,
                  buf_rewards, buf_values, buf_flags, buf_rew6 (or None), traj_len, side_obs, side_t,
                  side_count, pool (uint8 [N*pool_depth*656]), pool_count, ctr, buf_mu (or None)."""
        sp = self.a3_spec
        if sp is None or not self.contact_ok:
            raise OlyError("a3_vec_prepare before a3_configure / contact_configure")
        dv, f32, f64, i32, u8 = self.device, torch.float32, torch.float64, torch.int32, torch.uint8
        K, N = (int(v) for v in blocks["qpos"].shape[:2])
        Cc = int(blocks["geom1"].shape[2])
        cb = _abi.A3Blocks()
        cb.K, cb.C = K, Cc
        shapes = dict(qpos=(sp.nq,), qvel=(sp.nv,), act_len=(sp.nu,), act_vel=(sp.nu,), lf_pos=(3,), rf_pos=(3,),
                      lf_vel=(3,), rf_vel=(3,), root_pos=(3,), root_quat=(4,), head_pos=(3,), ncon=(), geom1=(Cc,),
                      geom2=(Cc,), force6=(Cc, 6), cpos_z=(Cc,))
        for name, tail in shapes.items():
            dt = i32 if name in ("ncon", "geom1", "geom2") else f64
            setattr(cb, name, _req(blocks.get(name), name, (K, N) + tail, dt, dv).data_ptr())
        cst = self._a3_struct(_abi.A3State, self._A3_ST, state, N)
        T, nobs, nu = int(ro["T"]), sp.n_obs, sp.nu
        slots, depth = int(ro["side_slots"]), int(ro["pool_depth"])
        det = bool(ro["deterministic"])
        cr = _abi.A3Rollout()
        cr.T, cr.max_traj_len, cr.deterministic = T, int(ro["max_traj_len"]), int(det)
        cr.side_slots, cr.pool_depth = slots, depth
        spec = dict(mu=((N, nu), f32), value=((N,), f32), scale=((nu,), f32), eps=((T, N, nu), f32),
                    state=((N, nobs), f32), pd_target=((N, nu), f64), buf_states=((T, N, nobs), f32),
                    buf_actions=((T, N, nu), f32), buf_rewards=((T, N), f64), buf_values=((T, N), f32),
                    buf_flags=((T, N), u8), buf_rew6=((T, N, 6), f32), traj_len=((N,), i32),
                    side_obs=((N * slots, nobs), f32), side_t=((N * slots,), i32), side_count=((N,), i32),
                    pool=((N * depth * C.sizeof(_abi.A3ResetRecord),), u8), pool_count=((N,), i32),
                    ctr=((self.a3_vec_ctr_len(N),), i32), buf_mu=((T, N, nu), f32))
        optional = {"buf_rew6", "buf_mu"} | ({"scale", "eps"} if det else set())
        for name, (shape, dt) in spec.items():
            tns = _req(ro.get(name), name, shape, dt, dv, optional=name in optional)
            setattr(cr, name, None if tns is None else tns.data_ptr())
        keep = (blocks, state, ro)          # the structs hold raw addresses: keep the tensors alive
        from ._ffi import check, lib
        fn, h = lib().oly_a3_vec_step, self.ctx.handle

        def launch(flags=0, mu=None, value=None):
            if mu is not None:
                cr.mu = _req(mu, "mu", (N, nu), f32, dv).data_ptr()
            if value is not None:
                cr.value = _req(value, "value", (N,), f32, dv).data_ptr()
            rc = fn(h, N, C.byref(cb), C.byref(cst), C.byref(cr), int(flags), self._s())
            if rc:
                check(h, rc, "oly_a3_vec_step")
            return keep
        pfn = lib().oly_a3_rollout_persistent

        def persistent(packed_actor, norm_actor, packed_critic, norm_critic, mu_out=None, value_out=None):
            """The remaining steps of the rollout (device counter t .. T - 1) in ONE launch (K13): per step the
            actor + critic forward on the packed weights, then this same vec step; bit-identical buffers."""
            for name, w, od in (("packed_actor", packed_actor, nu), ("packed_critic", packed_critic, 1)):
                _req(w, name, (self._mlp_floats(nobs, od),), f32, dv)
            if mu_out is not None:
                _req(mu_out, "mu_out", (N, nu), f32, dv)
            if value_out is not None:
                _req(value_out, "value_out", (N,), f32, dv)
            rc = pfn(h, N, C.byref(cb), C.byref(cst), C.byref(cr), nobs, ptr(packed_actor), int(bool(norm_actor)),
                     ptr(packed_critic), int(bool(norm_critic)), ptr(mu_out), ptr(value_out), self._s())
            if rc:
                check(h, rc, "oly_a3_rollout_persistent")
            return keep
        launch.persistent = persistent
        launch.structs = (cb, cst, cr)       # the C structs of this launch (tests call the C entry points with them)
        return launch

    def a3_pd_target(self, action):
        sp = self.a3_spec
        N = int(action.shape[0])
        _req(action, "action", (N, sp.nu), torch.float32, self.device)
        target = self._new((N, sp.nu), torch.float64)
        self.ctx.call("oly_a3_pd_target", N, ptr(action), ptr(target), self._s())
        return target

    def a3_pd_torque(self, kp, kd, target, act_len, act_vel):
        sp = self.a3_spec
        N = int(target.shape[0])
        for t, nme in ((target, "target"), (act_len, "act_len"), (act_vel, "act_vel")):
            _req(t, nme, (N, sp.nu), torch.float64, self.device)
        _req(kp, "kp", (sp.nu,), torch.float64, self.device)
        _req(kd, "kd", (sp.nu,), torch.float64, self.device)
        tau = self._new((N, sp.nu), torch.float64)
        self.ctx.call("oly_a3_pd_torque", N, ptr(kp), ptr(kd), ptr(target), ptr(act_len), ptr(act_vel), ptr(tau),
                      self._s())
        return tau

    # -------------------------------------------------------------- K11 (fused MLP forward)
    def mlp_pack(self, w1, b1, w2, b2, w3, b3, in_mean=None, in_std=None, packed=None):
        """torch Linear parameters of a relu MLP in -> 256 -> 256 -> out -> packed operand stream."""
        from ._ffi import lib
        f32, dv = torch.float32, self.device
        H, in_dim = (int(v) for v in w1.shape)
        out_dim = int(w3.shape[0])
        n = int(lib().oly_mlp_packed_floats(in_dim, H, out_dim))
        if n < 0 or tuple(w2.shape) != (H, H) or int(w3.shape[1]) != H:
            raise OlyError(f"mlp_pack: unsupported MLP shape {in_dim} -> {tuple(w2.shape)} -> {out_dim}")
        for t, name, shape in ((w1, "w1", (H, in_dim)), (b1, "b1", (H,)), (w2, "w2", (H, H)), (b2, "b2", (H,)),
                               (w3, "w3", (out_dim, H)), (b3, "b3", (out_dim,))):
            _req(t, name, shape, f32, dv)
        _req(in_mean, "in_mean", (in_dim,), f32, dv, optional=True)
        _req(in_std, "in_std", (in_dim,), f32, dv, optional=True)
        packed = _req(packed if packed is not None else self._new((n,), f32), "packed", (n,), f32, dv)
        self.ctx.call("oly_mlp_pack", in_dim, H, out_dim, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3),
                      ptr(in_mean), ptr(in_std), ptr(packed), self._s())
        return packed

    def _mlp_floats(self, in_dim, out_dim):
        key = (int(in_dim), int(out_dim))
        cache = self.__dict__.setdefault("_mlp_sizes", {})
        if key not in cache:
            from ._ffi import lib
            cache[key] = int(lib().oly_mlp_packed_floats(key[0], 256, key[1]))
        return cache[key]

    def mlp_forward2(self, x, packed_a, out_a, y_a, packed_b=None, out_b=0, y_b=None, normalize_a=False,
                     normalize_b=False):
        """y_a = net_a(x) and (optionally) y_b = net_b(x) in one launch; x [N,in] f32."""
        N, in_dim = (int(v) for v in x.shape)
        f32, dv = torch.float32, self.device
        _req(x, "x", (N, in_dim), f32, dv)
        # the raw pointers carry no length: a buffer that is not a whole packed stream is refused here
        _req(packed_a, "packed_a", (self._mlp_floats(in_dim, out_a),), f32, dv)
        _req(y_a, "y_a", (N, int(out_a)), f32, dv)
        if packed_b is not None:
            _req(packed_b, "packed_b", (self._mlp_floats(in_dim, out_b),), f32, dv)
            _req(y_b, "y_b", (N, int(out_b)), f32, dv)
        self.ctx.call("oly_mlp_forward2", N, in_dim, ptr(x), ptr(packed_a), int(out_a), int(bool(normalize_a)), ptr(y_a),
                      ptr(packed_b), int(out_b), int(bool(normalize_b)), ptr(y_b), self._s())
        return y_a, y_b

    # -------------------------------------------------------------- K14 (fused PPO update gradients)
    def ppo_update_plan(self, B, in_dim, act_dim, mirror=False):
        """(workspace floats, parts_actor, parts_critic) for minibatches of B rows on this device."""
        from ._ffi import lib
        pa, pc = C.c_int32(0), C.c_int32(0)
        n = int(lib().oly_ppo_update_ws_floats(self.ctx.handle, int(B), int(in_dim), int(act_dim), int(bool(mirror)),
                                               C.byref(pa), C.byref(pc)))
        if n < 0:
            raise OlyError(f"ppo_update: unsupported shape B={B} in={in_dim} act={act_dim}")
        return n, int(pa.value), int(pc.value)

    def ppo_update_grads(self, obs, action, adv, ret, old_mu, packed_actor, packed_critic, sd, log_sd, old_sd, old_log_sd,
                         grad_actor, grad_critic, scal_out, ws, idx=None, mir_obs=None, act_src=None, act_sign=None,
                         normalize_actor=True, normalize_critic=False, clip=0.2, vf_coeff=0.5, mirror_coeff=0.0,
                         parts=(0, 0), gnorm_ws=None, prepare=False):
        """oly_ppo_update_grads: gradients of one PPO minibatch update (rl/algos/ppo.py:232-282,396-410) into the flat
        buffers grad_actor / grad_critic (parameter order), the six scalars of update_policy into scal_out [6] f64.
        gnorm_ws [1024] f64: the finishing launch also leaves the squared-norm block partials ppo_adam_step(norm_ready=True)
        reads.  prepare=True: validate now, launch nothing, return `launch(idx, scal_out)` for minibatches of the same size
        on the same buffers (no per-call checks: at minibatch 64 they cost more host time than the kernels run)."""
        from ._ffi import lib
        f32, dv = torch.float32, self.device
        if obs.dim() != 2 or action.dim() != 2:
            raise OlyError("ppo_update_grads: obs [n, in] and action [n, act] expected")
        n, in_dim = (int(v) for v in obs.shape)
        act_dim = int(action.shape[1])
        _req(obs, "obs", (n, in_dim), f32, dv)
        _req(action, "action", (n, act_dim), f32, dv)
        _req(adv, "adv", (n,), f32, dv)
        _req(ret, "ret", (n,), f32, dv)
        _req(old_mu, "old_mu", (n, act_dim), f32, dv)
        _req(mir_obs, "mir_obs", (n, in_dim), f32, dv, optional=True)
        if idx is not None:
            if idx.dim() != 1:
                raise OlyError("idx: expected [B]")
            _req(idx, "idx", (int(idx.shape[0]),), torch.int32, dv)
        B = int(idx.shape[0]) if idx is not None else n
        for t, name in ((sd, "sd"), (log_sd, "log_sd"), (old_sd, "old_sd"), (old_log_sd, "old_log_sd")):
            _req(t, name, (act_dim,), f32, dv)
        if mir_obs is not None:
            _req(act_src, "act_src", (act_dim,), torch.int32, dv)
            _req(act_sign, "act_sign", (act_dim,), f32, dv)
        _req(packed_actor, "packed_actor", (self._mlp_floats(in_dim, act_dim),), f32, dv)
        _req(packed_critic, "packed_critic", (self._mlp_floats(in_dim, 1),), f32, dv)
        ga = int(lib().oly_ppo_update_grad_floats(in_dim, 256, act_dim))
        gc = int(lib().oly_ppo_update_grad_floats(in_dim, 256, 1))
        if ga < 0 or gc < 0:
            raise OlyError(f"ppo_update_grads: unsupported shape in={in_dim} act={act_dim}")
        _req(grad_actor, "grad_actor", (ga,), f32, dv)
        _req(grad_critic, "grad_critic", (gc,), f32, dv)
        _req(scal_out, "scal_out", (6,), torch.float64, dv)
        if ws.dim() != 1:
            raise OlyError("ws: expected a flat float32 workspace")
        _req(ws, "ws", (int(ws.shape[0]),), f32, dv)
        u = _abi.PPOUpdate()
        u.B, u.in_dim, u.act_dim = B, in_dim, act_dim
        u.parts_actor, u.parts_critic = int(parts[0]), int(parts[1])
        u.normalize_actor, u.normalize_critic = int(bool(normalize_actor)), int(bool(normalize_critic))
        u.obs, u.mir_obs, u.action, u.adv, u.ret, u.old_mu, u.idx = (ptr(t) for t in (obs, mir_obs, action, adv, ret, old_mu, idx))
        u.packed_actor, u.packed_critic = ptr(packed_actor), ptr(packed_critic)
        u.sd, u.log_sd, u.old_sd, u.old_log_sd = ptr(sd), ptr(log_sd), ptr(old_sd), ptr(old_log_sd)
        u.act_src, u.act_sign = ptr(act_src if mir_obs is not None else None), ptr(act_sign if mir_obs is not None else None)
        u.clip, u.vf_coeff, u.mirror_coeff = float(clip), float(vf_coeff), float(mirror_coeff)
        u.grad_actor, u.grad_critic, u.scal_out = ptr(grad_actor), ptr(grad_critic), ptr(scal_out)
        u.ws, u.ws_floats = ptr(ws), int(ws.shape[0])
        _req(gnorm_ws, "gnorm_ws", (1024,), torch.float64, dv, optional=True)
        u.gnorm_ws = ptr(gnorm_ws)
        if prepare:
            from ._ffi import check
            fn, h = lib().oly_ppo_update_grads, self.ctx.handle
            keep = (obs, action, adv, ret, old_mu, mir_obs, packed_actor, packed_critic, sd, log_sd, old_sd, old_log_sd, act_src,
                    act_sign, grad_actor, grad_critic, ws, gnorm_ws)
            ref = C.byref(u)

            def launch(idx_, scal_):
                if idx_.dtype != torch.int32 or idx_.shape[0] != B or not idx_.is_contiguous():
                    raise OlyError("prepared ppo_update_grads: idx must be a contiguous int32 [B] of the prepared size")
                u.idx, u.scal_out = idx_.data_ptr(), scal_.data_ptr()
                rc = fn(h, ref, self._s())
                if rc:
                    check(h, rc, "oly_ppo_update_grads")
                return keep
            launch.struct, launch.B, launch.keep = u, B, keep
            return launch
        self.ctx.call("oly_ppo_update_grads", C.byref(u), self._s())
        return grad_actor, grad_critic, scal_out

    def ppo_adam_step(self, in_dim, step, lr, eps, max_grad_norm, nets, ws, beta1=0.9, beta2=0.999, norm_ready=False,
                      prepare=False):
        """oly_ppo_adam_step: clip_grad_norm_ + Adam.step + weight re-pack for (actor, critic).  nets: two dicts with
        flat f32 tensors param / grad / exp_avg / exp_avg_sq, out_dim, and optionally packed / in_mean / in_std.
        norm_ready: `ws` already holds the squared-norm partials of these gradients (gnorm_ws of ppo_update_grads).
        prepare=True: validate now, launch nothing, return `launch(step, norm_ready)`."""
        from ._ffi import lib
        f32, dv = torch.float32, self.device
        a = _abi.PPOAdam()
        a.in_dim, a.step = int(in_dim), int(step)
        a.lr, a.beta1, a.beta2, a.eps, a.max_grad_norm = float(lr), float(beta1), float(beta2), float(eps), float(max_grad_norm)
        if len(nets) != 2:
            raise OlyError("ppo_adam_step: nets = (actor, critic)")
        for i, nt in enumerate(nets):
            out_dim = int(nt["out_dim"])
            gf = int(lib().oly_ppo_update_grad_floats(int(in_dim), 256, out_dim))
            if gf < 0:
                raise OlyError(f"ppo_adam_step: unsupported shape in={in_dim} out={out_dim}")
            for k in ("param", "grad", "exp_avg", "exp_avg_sq"):
                _req(nt[k], k, (gf,), f32, dv)
            packed = nt.get("packed")
            if packed is not None:
                _req(packed, "packed", (self._mlp_floats(in_dim, out_dim),), f32, dv)
            _req(nt.get("in_mean"), "in_mean", (int(in_dim),), f32, dv, optional=True)
            _req(nt.get("in_std"), "in_std", (int(in_dim),), f32, dv, optional=True)
            s = a.net[i]
            s.param, s.grad, s.exp_avg, s.exp_avg_sq = ptr(nt["param"]), ptr(nt["grad"]), ptr(nt["exp_avg"]), ptr(nt["exp_avg_sq"])
            s.packed, s.in_mean, s.in_std, s.out_dim = ptr(packed), ptr(nt.get("in_mean")), ptr(nt.get("in_std")), out_dim
        _req(ws, "ws", (1024,), torch.float64, dv)
        a.ws = ptr(ws)
        a.norm_ready = int(bool(norm_ready))
        if prepare:
            from ._ffi import check
            fn, h, ref, keep = lib().oly_ppo_adam_step, self.ctx.handle, C.byref(a), (nets, ws)

            def launch(step_, ready_):
                a.step, a.norm_ready = int(step_), 1 if ready_ else 0
                rc = fn(h, ref, self._s())
                if rc:
                    check(h, rc, "oly_ppo_adam_step")
                return keep
            launch.struct, launch.keep = a, keep
            return launch
        self.ctx.call("oly_ppo_adam_step", C.byref(a), self._s())

    def ppo_update_epoch(self, grads_launch, adam_launch, first_step, perm, n_batches, scal_out):
        """oly_ppo_update_epoch: the minibatch loop of one epoch in one call, from the two prepared launches
        (ppo_update_grads / ppo_adam_step with prepare=True): minibatch b takes rows perm[b B : (b + 1) B], leaves its six
        scalars in scal_out[b] and is optimiser step first_step + b."""
        u, a, B = grads_launch.struct, adam_launch.struct, grads_launch.B
        n_batches = int(n_batches)
        if perm.dim() != 1 or int(perm.shape[0]) < n_batches * B:
            raise OlyError(f"ppo_update_epoch: perm holds {tuple(perm.shape)} indices, {n_batches} minibatches of {B} need {n_batches * B}")
        _req(perm, "perm", (int(perm.shape[0]),), torch.int32, self.device)
        _req(scal_out, "scal_out", (n_batches, 6), torch.float64, self.device)
        a.step = int(first_step)
        self.ctx.call("oly_ppo_update_epoch", C.byref(u), C.byref(a), ptr(perm), n_batches, ptr(scal_out), self._s())

    # -------------------------------------------------------------- K6
    def return_scan(self, mode, gamma, lam, rew, val, next_val, flags, ret=None, adv=None, stats3=None):
        """rew [T,N] float32, or float64 (RETURN mode: the un-narrowed reward of env.step).  With
        `stats3` ([3] f64 device tensor) the same pass also leaves (T*N, sum adv, sum adv^2) there."""
        if rew.dim() != 2:
            raise OlyError(f"rew: expected [T,N], got {tuple(rew.shape)}")
        T, N = int(rew.shape[0]), int(rew.shape[1])
        dv = self.device
        r64 = rew.dtype == torch.float64
        if r64 and int(mode) != _abi.SCAN_RETURN:
            raise OlyError("float64 rewards are defined for SCAN_RETURN only")
        _req(rew, "rew", (T, N), torch.float64 if r64 else torch.float32, dv)
        _req(val, "val", (T, N), torch.float32, dv)
        _req(next_val, "next_val", (T, N), torch.float32, dv)
        _req(flags, "flags", (T, N), torch.uint8, dv)
        _req(stats3, "stats3", (3,), torch.float64, dv, optional=True)
        ret = _req(ret if ret is not None else self._new((T, N), torch.float32), "ret", (T, N), torch.float32, dv)
        adv = _req(adv if adv is not None else self._new((T, N), torch.float32), "adv", (T, N), torch.float32, dv)
        self.ctx.call("oly_return_scan_stats", int(mode) | (_abi.SCAN_REW_F64 if r64 else 0), T, N, C.c_double(gamma),
                      C.c_double(lam), ptr(rew), ptr(val), ptr(next_val), ptr(flags), ptr(ret), ptr(adv), ptr(stats3),
                      self._s())
        return ret, adv

    def rollout_cuts(self, done, traj_len, flags, n_cut, max_traj_len, last_step):
        """PPO.sample's per-step episode-cut bookkeeping (traj_len / flags updated in place; n_cut [1] i32)."""
        N = int(done.shape[0])
        _req(done, "done", (N,), torch.uint8, self.device)
        _req(traj_len, "traj_len", (N,), torch.int32, self.device)
        _req(flags, "flags", (N,), torch.uint8, self.device)
        _req(n_cut, "n_cut", (1,), torch.int32, self.device)
        self.ctx.call("oly_rollout_cuts", N, int(max_traj_len), int(bool(last_step)), ptr(done), ptr(traj_len), ptr(flags),
                      ptr(n_cut), self._s())
        return flags

    # -------------------------------------------------------------- K7
    def adv_stats(self, x, stats3=None):
        _req(x, "x", x.shape, torch.float32, self.device)
        stats3 = _req(stats3 if stats3 is not None else self._new((3,), torch.float64), "stats3", (3,), torch.float64, self.device)
        self.ctx.call("oly_adv_stats", C.c_int64(x.numel()), ptr(x), ptr(stats3), self._s())
        return stats3

    def adv_normalize(self, x, stats3, ddof, eps):
        """stats3: [3] (one triple) or [parts,3] (one triple per rank, as all-gathered)."""
        _req(x, "x", x.shape, torch.float32, self.device)
        parts = 1 if stats3.dim() == 1 else int(stats3.shape[0])
        if not 1 <= parts <= _abi.OLY_MAX_STAT_PARTS:
            raise OlyError(f"stats3: {parts} parts, at most {_abi.OLY_MAX_STAT_PARTS}")
        _req(stats3, "stats3", (3,) if stats3.dim() == 1 else (parts, 3), torch.float64, self.device)
        self.ctx.call("oly_adv_normalize_parts", C.c_int64(x.numel()), ptr(x), ptr(stats3), parts, int(ddof),
                      C.c_double(eps), self._s())
        return x

    def col_stats(self, x, colstats=None):
        if x.dim() != 2:
            raise OlyError(f"x: expected [B,D], got {tuple(x.shape)}")
        B, D = int(x.shape[0]), int(x.shape[1])
        _req(x, "x", (B, D), torch.float32, self.device)
        acc = colstats is not None
        colstats = _req(colstats if acc else self._new((3, D), torch.float64), "colstats", (3, D), torch.float64, self.device)
        self.ctx.call("oly_col_stats", B, D, ptr(x), ptr(colstats), int(acc), self._s())
        return colstats

    # -------------------------------------------------------------- K3 (IL ground forces)
    def grf_configure(self, geom_group, pairs):
        """geom_group [ngeom] (collision-group index per geom, -1 none); pairs [(group_a, group_b), ...]."""
        gg = np.ascontiguousarray(geom_group, dtype=np.int32)
        pa = np.ascontiguousarray([a for a, _ in pairs], dtype=np.int32)
        pb = np.ascontiguousarray([b for _, b in pairs], dtype=np.int32)
        self.ctx.call("oly_grf_configure", len(gg), ptr(gg), len(pairs), ptr(pa), ptr(pb))
        self.n_grf_pairs = len(pairs)
        return self

    def il_ground_forces(self, ncon, geom1, geom2, force6, want_steps=False, check=False):
        """[W,N,...] substep contact slots -> dict(mean [N,3P], steps [W,N,3P] or None, overflow [N] u8).
        `ncon` is the raw data.ncon.  An environment whose count exceeded the staged slots in a substep where a
        sensor pair found no contact among them cannot be reproduced (the reference scans every contact,
        UnitreeH1.py:113-123): the kernel flags it in `overflow` and the caller owns those bytes (VecLocoEnv keeps a
        sticky copy and reads it in its next host round trip: reset() / raise_if_contact_overflow()).  check=True reads
        them back HERE: a blocking device-to-host round trip per call, which also breaks graph capture of the caller."""
        if not getattr(self, "n_grf_pairs", 0):
            raise OlyError("il_ground_forces before grf_configure")
        W, N, Cc = (int(v) for v in geom1.shape)
        _req(ncon, "ncon", (W, N), torch.int32, self.device)
        _req(geom1, "geom1", (W, N, Cc), torch.int32, self.device)
        _req(geom2, "geom2", (W, N, Cc), torch.int32, self.device)
        _req(force6, "force6", (W, N, Cc, 6), torch.float64, self.device)
        k = 3 * self.n_grf_pairs
        mean = self._new((N, k), torch.float64)
        steps = self._new((W, N, k), torch.float64) if want_steps else None
        over = self._new((N,), torch.uint8)
        self.ctx.call("oly_il_ground_forces", W, N, Cc, ptr(ncon), ptr(geom1), ptr(geom2), ptr(force6), ptr(steps),
                      ptr(mean), ptr(over), self._s())
        if check and N and bool(over.any().item()):
            bad = torch.nonzero(over).flatten()[:8].tolist()
            raise OlyError(f"il_ground_forces: more contacts than the {Cc} staged slots and a sensor pair without a "
                           f"contact among them (or a negative count) in environments {bad}: stage more slots")
        return dict(mean=mean, steps=steps, overflow=over)

    def il_grf_window(self, grf_step, mean=None):
        """[W,N,K] per-substep ground-force rows -> [N,K] window mean (sum in substep order / W)."""
        W, N, K = (int(v) for v in grf_step.shape)
        _req(grf_step, "grf_step", (W, N, K), torch.float64, self.device)
        mean = _req(mean if mean is not None else self._new((N, K), torch.float64), "mean", (N, K), torch.float64,
                    self.device)
        self.ctx.call("oly_il_grf_window", W, N, K, ptr(grf_step), ptr(mean), self._s())
        return mean

    # -------------------------------------------------------------- K8
    def disc_standardize(self, x, mask, mean, std, out=None):
        B, Dx = int(x.shape[0]), int(x.shape[1])
        _req(x, "x", (B, Dx), torch.float32, self.device)
        D = Dx if mask is None else int(mask.shape[0])
        _req(mask, "mask", (D,), torch.int32, self.device, optional=True)
        _req(mean, "mean", (D,), torch.float64, self.device)
        _req(std, "std", (D,), torch.float64, self.device)
        out = _req(out if out is not None else self._new((B, D), torch.float32), "out", (B, D), torch.float32, self.device)
        self.ctx.call("oly_disc_standardize", B, Dx, D, ptr(x), ptr(mask), ptr(mean), ptr(std), ptr(out), self._s())
        return out

    def obs_filter(self, x, mean, var, eps=1e-8, clip=10.0, out=None):
        """Normalize._obfilt on a [B,D] float32 batch (out may be x itself)."""
        B, D = int(x.shape[0]), int(x.shape[1])
        _req(x, "x", (B, D), torch.float32, self.device)
        _req(mean, "mean", (D,), torch.float64, self.device)
        _req(var, "var", (D,), torch.float64, self.device)
        out = _req(out if out is not None else self._new((B, D), torch.float32), "out", (B, D), torch.float32, self.device)
        self.ctx.call("oly_obs_filter", B, D, ptr(x), ptr(mean), ptr(var), float(eps), float(clip or 0.0), ptr(out),
                      self._s())
        return out

    def disc_reparam(self, mu, logvar, eps, z=None):
        for t, nme in ((mu, "mu"), (logvar, "logvar"), (eps, "eps")):
            _req(t, nme, mu.shape, torch.float32, self.device)
        z = _req(z if z is not None else torch.empty_like(mu), "z", mu.shape, torch.float32, self.device)
        self.ctx.call("oly_disc_reparam", C.c_int64(mu.numel()), ptr(mu), ptr(logvar), ptr(eps), ptr(z), self._s())
        return z

    def disc_reward(self, logits, reward=None):
        _req(logits, "logits", logits.shape, torch.float32, self.device)
        reward = _req(reward if reward is not None else self._new((logits.numel(),), torch.float32), "reward",
                      (logits.numel(),), torch.float32, self.device)
        self.ctx.call("oly_disc_reward", C.c_int64(logits.numel()), ptr(logits), ptr(reward), self._s())
        return reward

    # -------------------------------------------------------------- K12 (fused discriminator forward)
    def disc_pack(self, enc_w0, enc_b0, enc_w1, enc_b1, mu_w, mu_b, lv_w, lv_b, dec_w, dec_b, packed=None):
        """torch parameters of the variational discriminator in -> 256 -> 128 -> (mu, logvar) 128 -> 1
        -> packed operand stream of oly_disc_forward."""
        from ._ffi import lib
        f32, dv = torch.float32, self.device
        H, in_dim = (int(v) for v in enc_w0.shape)
        E, Z = int(enc_w1.shape[0]), int(mu_w.shape[0])
        n = int(lib().oly_disc_packed_floats(in_dim, H, E, Z))
        if n < 0:
            raise OlyError(f"disc_pack: unsupported discriminator shape {in_dim} -> {H} -> {E} -> {Z} -> 1")
        for t, name, shape in ((enc_w0, "enc_w0", (H, in_dim)), (enc_b0, "enc_b0", (H,)), (enc_w1, "enc_w1", (E, H)),
                               (enc_b1, "enc_b1", (E,)), (mu_w, "mu_w", (Z, E)), (mu_b, "mu_b", (Z,)),
                               (lv_w, "lv_w", (Z, E)), (lv_b, "lv_b", (Z,)), (dec_w, "dec_w", (1, Z)),
                               (dec_b, "dec_b", (1,))):
            _req(t, name, shape, f32, dv)
        packed = _req(packed if packed is not None else self._new((n,), f32), "packed", (n,), f32, dv)
        self.ctx.call("oly_disc_pack", in_dim, H, E, Z, ptr(enc_w0), ptr(enc_b0), ptr(enc_w1), ptr(enc_b1), ptr(mu_w),
                      ptr(mu_b), ptr(lv_w), ptr(lv_b), ptr(dec_w), ptr(dec_b), ptr(packed), self._s())
        return packed

    def disc_forward(self, x, packed, mask=None, mean=None, std=None, colstats=None, eps=None, want=("reward",),
                     out=None):
        """make_discrim_reward for x [B,Dx] f32 in one launch.  want: any of reward / logits / mu / logvar.
        Standardisation from mean / std [D] f64, or from the running colstats [3,D] of col_stats, or none."""
        from ._ffi import lib
        f32, dv = torch.float32, self.device
        B, Dx = (int(v) for v in x.shape)
        D = Dx if mask is None else int(mask.shape[0])
        _req(x, "x", (B, Dx), f32, dv)
        _req(mask, "mask", (D,), torch.int32, dv, optional=True)
        if (mean is None) != (std is None) or (mean is not None and colstats is not None):
            raise OlyError("disc_forward: give mean and std together, or colstats, or neither")
        _req(colstats, "colstats", (3, D), torch.float64, dv, optional=True)
        _req(mean, "mean", (D,), torch.float64, dv, optional=True)
        _req(std, "std", (D,), torch.float64, dv, optional=True)
        cache = self.__dict__.setdefault("_disc_sizes", {})
        if D not in cache:
            cache[D] = int(lib().oly_disc_packed_floats(D, 256, 128, 128))
        _req(packed, "packed", (cache[D],), f32, dv)        # the raw pointer carries no length
        _req(eps, "eps", (B, 128), f32, dv, optional=True)
        shapes = dict(reward=(B,), logits=(B,), mu=(B, 128), logvar=(B, 128))
        out = dict(out or {})
        for k in want:
            if k not in shapes:
                raise OlyError(f"disc_forward: unknown output {k!r}")
            out[k] = _req(out.get(k) if out.get(k) is not None else self._new(shapes[k], f32), k, shapes[k], f32, dv)
        if not want:
            raise OlyError("disc_forward: no output requested")
        g = lambda k: ptr(out[k]) if k in want else None
        self.ctx.call("oly_disc_forward", C.c_int64(B), Dx, D, ptr(x), ptr(mask), ptr(mean), ptr(std), ptr(colstats),
                      ptr(packed), ptr(eps), g("reward"), g("logits"), g("mu"), g("logvar"), self._s())
        return out

    def disc_reward_step(self, x, packed, colstats, accumulate, eps=None, want=("reward",), out=None, weights=None):
        """oly_disc_reward_step: (re-pack of `weights`, the ten parameter tensors in oly_disc_pack's order, when given,)
        the Standardizer's running update with x (into colstats [3,D] f64, accumulate: added to the running sums) and
        the fused forward on the updated statistics, one C call.  x [B,D] f32, whole rows.  accumulate=None: validate
        now and return `launch(accumulate)` for repeated calls on the same tensors."""
        from ._ffi import lib
        f32, dv = torch.float32, self.device
        B, D = (int(v) for v in x.shape)
        _req(x, "x", (B, D), f32, dv)
        _req(colstats, "colstats", (3, D), torch.float64, dv)
        cache = self.__dict__.setdefault("_disc_sizes", {})
        if D not in cache:
            cache[D] = int(lib().oly_disc_packed_floats(D, 256, 128, 128))
        _req(packed, "packed", (cache[D],), f32, dv)
        _req(eps, "eps", (B, 128), f32, dv, optional=True)
        wp = None
        if weights is not None:
            shapes_w = ((256, D), (256,), (128, 256), (128,), (128, 128), (128,), (128, 128), (128,), (1, 128), (1,))
            if len(weights) != 10:
                raise OlyError("disc_reward_step: weights = the ten tensors of disc_pack")
            for i, (t, sh) in enumerate(zip(weights, shapes_w)):
                _req(t, f"weights[{i}]", sh, f32, dv)
            wp = (C.c_void_p * 10)(*[t.data_ptr() for t in weights])
        shapes = dict(reward=(B,), logits=(B,), mu=(B, 128), logvar=(B, 128))
        out = dict(out or {})
        if not want:
            raise OlyError("disc_reward_step: no output requested")
        for k in want:
            if k not in shapes:
                raise OlyError(f"disc_reward_step: unknown output {k!r}")
            out[k] = _req(out.get(k) if out.get(k) is not None else self._new(shapes[k], f32), k, shapes[k], f32, dv)
        g = lambda k: ptr(out[k]) if k in want else None
        if accumulate is None:
            # validated once: `launch(accumulate)` re-issues the same call on the current stream with no per-call checks
            # (at B = 4096 the kernels take ~20 us; the checks above cost the host more than that)
            from ._ffi import check
            fn, h = lib().oly_disc_reward_step, self.ctx.handle
            args = (C.c_int64(B), D, ptr(x), ptr(colstats))
            tail = (wp, ptr(packed), ptr(eps), g("reward"), g("logits"), g("mu"), g("logvar"))
            keep = (x, colstats, packed, eps, weights, out)

            def launch(acc):
                rc = fn(h, *args, 1 if acc else 0, *tail, self._s())
                if rc:
                    check(h, rc, "oly_disc_reward_step")
                return keep[-1]
            return launch
        self.ctx.call("oly_disc_reward_step", C.c_int64(B), D, ptr(x), ptr(colstats), int(bool(accumulate)), wp, ptr(packed),
                      ptr(eps), g("reward"), g("logits"), g("mu"), g("logvar"), self._s())
        return out

    # -------------------------------------------------------------- K9
    def signed_perm(self, x, src, sign, out=None):
        B, D = int(x.shape[0]), int(x.shape[1])
        _req(x, "x", (B, D), torch.float32, self.device)
        _req(src, "src", (D,), torch.int32, self.device)
        _req(sign, "sign", (D,), torch.float32, self.device)
        out = _req(out if out is not None else self._new((B, D), torch.float32), "out", (B, D), torch.float32, self.device)
        self.ctx.call("oly_signed_perm", B, D, ptr(x), ptr(src), ptr(sign), ptr(out), self._s())
        return out

    def mirror_loss(self, det, mir, src, sign, want_grad=True):
        """-> (loss [1] f64, grad_det, grad_mir)."""
        B, A = int(det.shape[0]), int(det.shape[1])
        _req(det, "det", (B, A), torch.float32, self.device)
        _req(mir, "mir", (B, A), torch.float32, self.device)
        _req(src, "src", (A,), torch.int32, self.device)
        _req(sign, "sign", (A,), torch.float32, self.device)
        loss = self._new((1,), torch.float64)
        gd = self._new((B, A), torch.float32) if want_grad else None
        gm = self._new((B, A), torch.float32) if want_grad else None
        self.ctx.call("oly_mirror_loss", B, A, ptr(det), ptr(mir), ptr(src), ptr(sign), ptr(loss), ptr(gd), ptr(gm),
                      self._s())
        return loss, gd, gm

    def _std_arg(self, std, B, A, name):
        if std.numel() == 1:
            mode, shape = _abi.STD_SCALAR, (1,)
        elif std.numel() == A:
            mode, shape = _abi.STD_PER_DIM, (A,)
        else:
            mode, shape = _abi.STD_FULL, (B, A)
        return _req(std.reshape(shape), name, shape, torch.float32, self.device), mode

    def ppo_loss(self, mu, std, old_mu, old_std, action, adv, ret, value, clip, vf_coeff=0.5, want_grad=True,
                 want_grad_std=False):
        """-> dict(scal [5] f64: actor, entropy_penalty, critic, approx_kl, clip_fraction;
        grad_mu [B,A], grad_std [B,A] or None, grad_value [B])."""
        B, A = int(mu.shape[0]), int(mu.shape[1])
        for t, nme in ((mu, "mu"), (old_mu, "old_mu"), (action, "action")):
            _req(t, nme, (B, A), torch.float32, self.device)
        std, sm = self._std_arg(std, B, A, "std")
        old_std, om = self._std_arg(old_std, B, A, "old_std")
        adv, ret, value = (_req(t.reshape(B), nme, (B,), torch.float32, self.device)
                           for t, nme in ((adv, "adv"), (ret, "ret"), (value, "value")))
        scal = self._new((5,), torch.float64)
        gmu = self._new((B, A), torch.float32) if want_grad else None
        gsd = self._new((B, A), torch.float32) if want_grad_std else None
        gv = self._new((B,), torch.float32) if want_grad else None
        self.ctx.call("oly_ppo_loss", B, A, ptr(mu), ptr(std), sm, ptr(old_mu), ptr(old_std), om, ptr(action),
                      ptr(adv), ptr(ret), ptr(value), float(clip), float(vf_coeff), ptr(scal), ptr(gmu), ptr(gsd),
                      ptr(gv), self._s())
        return dict(scal=scal, grad_mu=gmu, grad_std=gsd, grad_value=gv)
