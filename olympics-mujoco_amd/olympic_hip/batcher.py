"""Python handle on the C++ host physics batcher (csrc/host_batcher.hip).

    b = HostBatcher(engine, N, n_threads=0, dt=0.01)      # built-in kinematic physics
    b.qpos[:] = ...; b.qvel[:] = ...                       # numpy views of the pinned staging
    obs, reward, absorbing = b.step(action)                # device tensors

`physics` may be a Python callable (env, ctrl, qpos, qvel) for tests; a MuJoCo build passes a
C function pointer that wraps mj_step instead."""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._ffi import OlyError, check, lib, ptr


class HostBatcher:
    def __init__(self, engine, num_envs, n_threads=0, dt=0.01, physics=None, obs_f64=False):
        sp = engine.il_spec
        if sp is None:
            raise OlyError("HostBatcher needs an engine with il_configure() done")
        self.eng, self.N, self.spec, self.obs_f64 = engine, int(num_envs), sp, obs_f64
        self._cb = None
        fn = None
        if physics is not None:
            nq, nv, nu = sp.nq, sp.nv, sp.nu

            def tramp(env, ctrl, qpos, qvel, user):
                physics(env, np.ctypeslib.as_array(ctrl, (nu,)), np.ctypeslib.as_array(qpos, (nq,)),
                        np.ctypeslib.as_array(qvel, (nv,)))
            self._cb = _abi.PHYSICS_FN(tramp)
            fn = C.cast(self._cb, C.c_void_p)
        self._h = C.c_void_p()
        L = lib()
        check(engine.ctx.handle, L.oly_batcher_create(C.byref(self._h), engine.ctx.handle, self.N, int(n_threads),
                                                      C.c_double(dt), fn, None), "oly_batcher_create")
        self.qpos = np.ctypeslib.as_array(L.oly_batcher_qpos(self._h), (self.N, sp.nq))
        self.qvel = np.ctypeslib.as_array(L.oly_batcher_qvel(self._h), (self.N, sp.nv))
        dev = engine.device
        od = torch.float64 if obs_f64 else torch.float32
        self.obs = torch.empty((self.N, sp.n_obs), dtype=od, device=dev)
        self.reward = torch.empty(self.N, dtype=torch.float32, device=dev)
        self.absorbing = torch.empty(self.N, dtype=torch.uint8, device=dev)
        self.fall_code = torch.empty(self.N, dtype=torch.uint8, device=dev)

    def set_mapped(self, on=True):
        """Let the kernels address the pinned staging directly instead of issuing copy commands
        (oly_batcher_set_mapped).  True: controls always, state rows when a step moves less than 4 MB of
        them (beyond that the copy engine is faster than the kernel's own PCIe reads); False: copies; an int
        is passed through as the bit mask (1 controls, 2 state rows)."""
        if on is True:
            on = 1 | (2 if self.N * (self.spec.nq + self.spec.nv) * 8 <= (4 << 20) else 0)
        check(self.eng.ctx.handle, lib().oly_batcher_set_mapped(self._h, int(on)), "oly_batcher_set_mapped")
        return self

    def enable_contacts(self, n_intermediate, max_contacts, physics, packed=False):
        """packed=True: the worker threads reduce each environment's contact slots to the first-contact
        force of every sensor pair on the host (oly_batcher_enable_contacts_packed): same observations,
        W * 3 * n_pairs doubles per environment over PCIe instead of every slot.
        use_foot_forces: `physics(env, ctrl, qpos, qvel, con)` performs the control step's W
        intermediate steps and writes contact snapshot w into con["ncon"][w], con["geom1"][w, i],
        con["geom2"][w, i], con["force6"][w, i, :] (numpy views of the pinned staging)."""
        sp = self.spec
        nq, nv, nu = sp.nq, sp.nv, sp.nu
        W, Cc = int(n_intermediate), int(max_contacts)
        ast = np.lib.stride_tricks.as_strided

        def tramp(env, ctrl, qpos, qvel, oc, user):
            o = oc.contents
            n = np.ctypeslib.as_array(o.ncon, ((W - 1) * o.ncon_stride + 1,))
            g1 = np.ctypeslib.as_array(o.geom1, ((W - 1) * o.geom_stride + Cc,))
            g2 = np.ctypeslib.as_array(o.geom2, ((W - 1) * o.geom_stride + Cc,))
            f6 = np.ctypeslib.as_array(o.force6, ((W - 1) * o.force_stride + 6 * Cc,))
            con = dict(ncon=ast(n, (W,), (4 * o.ncon_stride,)), geom1=ast(g1, (W, Cc), (4 * o.geom_stride, 4)),
                       geom2=ast(g2, (W, Cc), (4 * o.geom_stride, 4)),
                       force6=ast(f6, (W, Cc, 6), (8 * o.force_stride, 48, 8)))
            physics(env, np.ctypeslib.as_array(ctrl, (nu,)), np.ctypeslib.as_array(qpos, (nq,)),
                    np.ctypeslib.as_array(qvel, (nv,)), con)
        self._ccb = _abi.PHYSICS_CONTACTS_FN(tramp)
        fn = lib().oly_batcher_enable_contacts_packed if packed else lib().oly_batcher_enable_contacts
        rc = fn(self._h, W, Cc, C.cast(self._ccb, C.c_void_p), None)
        check(self.eng.ctx.handle, rc, "oly_batcher_enable_contacts")
        return self

    def set_prev(self, prev):
        """Write the carried reward state (the reset observation's value) for every env."""
        t = torch.as_tensor(prev, dtype=torch.float64, device=self.eng.device).contiguous()
        if t.shape != (self.N,):
            raise OlyError(f"prev: shape {tuple(t.shape)}, expected ({self.N},)")
        rc = lib().oly_batcher_set_prev(self._h, ptr(t), self.eng._s())
        check(self.eng.ctx.handle, rc, "oly_batcher_set_prev")
        torch.cuda.current_stream(self.eng.device).synchronize()   # `t` may be a temporary

    def step(self, action):
        sp = self.spec
        if not isinstance(action, torch.Tensor) or action.device != self.eng.device or action.dtype != torch.float32 \
                or tuple(action.shape) != (self.N, sp.n_act) or not action.is_contiguous():
            raise OlyError("action must be a contiguous float32 device tensor of shape [N, n_act]")
        rc = lib().oly_batcher_step(self._h, ptr(action), ptr(self.obs), ptr(self.reward), ptr(self.absorbing),
                                    ptr(self.fall_code), _abi.OUT_OBS_F64 if self.obs_f64 else 0, self.eng._s())
        check(self.eng.ctx.handle, rc, "oly_batcher_step")
        return self.obs, self.reward, self.absorbing

    def last_timing(self):
        t = (C.c_double * 3)()
        lib().oly_batcher_last_timing(self._h, t)
        return dict(ctrl_d2h_s=t[0], physics_s=t[1], h2d_enqueue_s=t[2])

    def close(self):
        if self._h:
            lib().oly_batcher_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class A3HostBatcher:
    """C++ batcher of the RL robot (oly_a3_batcher_*): N StickFigureA3 environments whose physics
    (frame_skip PD substeps + readback) runs in a host thread pool and whose post-physics path
    (oly_contact_reduce -> oly_a3_step) runs on the device, one H2D copy per step.

    `physics(env, target [nu], slots)` fills `slots` - a dict of numpy views of that env's pinned
    staging rows (qpos, qvel, act_len, act_vel, lf_pos, rf_pos, lf_vel, rf_vel, root_pos, root_quat,
    head_pos, ncon [1], geom1 [C], geom2 [C], force6 [C,6], cpos_z [C]).  A MuJoCo build passes a C
    function pointer instead."""

    def __init__(self, engine, num_envs, max_contacts, physics, n_threads=0, obs_f64=False):
        if engine.a3_spec is None:
            raise OlyError("A3HostBatcher needs an engine with a3_configure() and contact_configure() done")
        self.eng, self.N, self.C, self.obs_f64 = engine, int(num_envs), int(max_contacts), obs_f64
        sp = engine.a3_spec
        self.spec = sp
        widths = dict(qpos=sp.nq, qvel=sp.nv, act_len=sp.nu, act_vel=sp.nu, lf_pos=3, rf_pos=3, lf_vel=3, rf_vel=3,
                      root_pos=3, root_quat=4, head_pos=3, ncon=1, geom1=self.C, geom2=self.C, force6=6 * self.C,
                      cpos_z=self.C)
        self._widths = widths

        def views(rb):
            out = {n: np.ctypeslib.as_array(getattr(rb, n), (widths[n],)) for n, _ in _abi.A3_READBACK_FIELDS}
            out["force6"] = out["force6"].reshape(self.C, 6)
            return out

        def tramp(env, target, rb, user):
            physics(env, np.ctypeslib.as_array(target, (sp.nu,)), views(rb.contents))
        self._views = views
        self._cb = _abi.A3_PHYSICS_FN(tramp) if physics is not None else None
        self._h = C.c_void_p()
        check(engine.ctx.handle, lib().oly_a3_batcher_create(C.byref(self._h), engine.ctx.handle, self.N, self.C,
                                                             int(n_threads),
                                                             C.cast(self._cb, C.c_void_p) if self._cb else None, None),
              "oly_a3_batcher_create")
        dev = engine.device
        self.obs = torch.empty((self.N, sp.n_obs), dtype=torch.float64 if obs_f64 else torch.float32, device=dev)
        self.rew6 = torch.empty((self.N, 6), dtype=torch.float32, device=dev)
        self.reward = torch.empty(self.N, dtype=torch.float32, device=dev)
        self.done = torch.empty(self.N, dtype=torch.uint8, device=dev)

    def set_mapped(self, on=True):
        """PD targets are stored straight into the pinned host rows (no D2H copy command)."""
        check(self.eng.ctx.handle, lib().oly_a3_batcher_set_mapped(self._h, int(on)), "oly_a3_batcher_set_mapped")
        return self

    def set_compact(self, on=True):
        """Only what the kernels read crosses PCIe (oly_a3_batcher_set_compact): the seven base numbers of
        qpos / qvel, the actuator / site rows and the USED contact slots; same results."""
        check(self.eng.ctx.handle, lib().oly_a3_batcher_set_compact(self._h, int(bool(on))), "oly_a3_batcher_set_compact")
        return self

    def slots(self, env):
        """numpy views of env's pinned staging rows (write a reset state here, then upload())."""
        rb = _abi.A3Readback()
        check(self.eng.ctx.handle, lib().oly_a3_batcher_slots(self._h, int(env), C.byref(rb)), "oly_a3_batcher_slots")
        return self._views(rb)

    def upload(self):
        check(self.eng.ctx.handle, lib().oly_a3_batcher_upload(self._h, self.eng._s()), "oly_a3_batcher_upload")

    def step(self, action, state, with_physics=True):
        """state: dict of device tensors named as oly_a3_state (updated in place).
        -> (obs, reward, done, rew6) device tensors owned by the batcher."""
        sp = self.spec
        if with_physics and (not isinstance(action, torch.Tensor) or action.device != self.eng.device
                             or action.dtype != torch.float32 or tuple(action.shape) != (self.N, sp.nu)
                             or not action.is_contiguous()):
            raise OlyError("action must be a contiguous float32 device tensor of shape [N, nu]")
        st = self.eng.a3_state_struct(state, self.N)
        rc = lib().oly_a3_batcher_step(self._h, ptr(action) if with_physics else None, C.byref(st), ptr(self.obs),
                                       ptr(self.rew6), ptr(self.reward), ptr(self.done),
                                       _abi.OUT_OBS_F64 if self.obs_f64 else 0, int(bool(with_physics)), self.eng._s())
        check(self.eng.ctx.handle, rc, "oly_a3_batcher_step")
        return self.obs, self.reward, self.done, self.rew6

    def last_timing(self):
        t = (C.c_double * 3)()
        lib().oly_a3_batcher_last_timing(self._h, t)
        return dict(target_d2h_s=t[0], physics_s=t[1], h2d_kernels_enqueue_s=t[2])

    def close(self):
        if self._h:
            lib().oly_a3_batcher_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
