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
