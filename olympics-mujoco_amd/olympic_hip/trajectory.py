"""Host-side trajectory preprocessing + the device-resident lookup cursor.

Mirrors olympic_mujoco/utils/trajectory.py of the reference (file:line citations below):
  loading / key handling          :56-96
  joint-range clipping            :325-366
  split by split_points           :195-228
  cubic resampling to control_dt  :230-287
  create_dataset                  :129-193
  reset / next-sample cursor      :289-323, :381-401   -> HIP kernels (K4) through `DeviceCursor`

The one-off preprocessing stays on the host in numpy/scipy (the reference uses the same
scipy call, so the resampled table is bit-identical); the per-step lookup for N environments
runs on the GPU from a table uploaded once.
"""
import warnings
from copy import deepcopy

import numpy as np
from scipy import interpolate


class Trajectory:
    """Equal-length reference trajectories, resampled to the control frequency."""

    def __init__(self, keys, low=None, high=None, joint_pos_idx=None, traj_path=None, traj_files=None,
                 traj_dt=0.002, control_dt=0.01, ignore_keys=None, clip_trajectory_to_joint_ranges=False,
                 traj_info=None, warn=True):
        if (traj_path is not None) == (traj_files is not None):
            raise AssertionError("Please specify either traj_path or traj_files, but not both.")
        if traj_path is not None:
            # the reference passes allow_pickle=True (:62); plain float arrays need no pickle and
            # nothing from the file is executed here
            with np.load(traj_path, allow_pickle=False) as f:
                files = {k: f[k] for k in f.files}
        else:
            files = {k: np.asarray(v) for k, v in traj_files.items()}
        keys = list(keys)
        if warn or clip_trajectory_to_joint_ranges:
            self._check_range(files, keys, low, high, joint_pos_idx, warn, clip_trajectory_to_joint_ranges)
        keys += [k for k in files if k.startswith("goal") and k not in keys]          # :74-78
        if ignore_keys is not None:
            for ik in ignore_keys:
                keys.remove(ik)
        self.keys = keys
        if "split_points" in files:
            self.split_points = np.asarray(files["split_points"])
        else:
            self.split_points = np.array([0, len(next(iter(files.values())))])
        cols = [np.asarray(files[k]) for k in keys]
        n = len(cols[0])
        if any(len(c) != n for c in cols):
            raise AssertionError("Some observations have different lengths than others. Trajectory is corrupted. ")
        parts = [np.split(c, self.split_points[1:-1]) for c in cols]
        L = len(parts[0][0])
        if any(len(p) != L for col in parts for p in col):
            raise AssertionError("Only trajectories of equal length are currently supported.")
        self.trajectories = [np.array(col) for col in parts]        # per key [n_traj, L]
        if traj_info is not None and len(traj_info) != self.number_of_trajectories:
            raise AssertionError("The number of trajectory infos/labels need to be equal to the number of trajectories.")
        self._traj_info = traj_info
        self.traj_dt, self.control_dt = traj_dt, control_dt
        if traj_dt != control_dt:
            self._resample()
        self.subtraj_step_no = 0
        self.traj_no = 0
        self.subtraj = self._get_subtraj(0)

    # ------------------------------------------------------------------ preprocessing
    @staticmethod
    def _check_range(files, keys, low, high, j_idx, warn, clip):
        """check_if_trajectory_is_in_range (:325-366): entries of the FILE dict, in file order,
        whose position is a joint-position index beyond x,y are tested against low/high."""
        j_idx = list(j_idx[2:])
        for i, (k, d) in enumerate(list(files.items())):
            if i in j_idx:
                hi_i, lo_i = high[i - 2], low[i - 2]
                if warn:
                    msg = "Clipping the trajectory into range!" if clip else ""
                    if np.max(d) > hi_i:
                        warnings.warn("Trajectory violates joint range in %s. Maximum in trajectory is %f "
                                      "and maximum range is %f. %s" % (keys[i], np.max(d), hi_i, msg), RuntimeWarning)
                    elif np.min(d) < lo_i:
                        warnings.warn("Trajectory violates joint range in %s. Minimum in trajectory is %f "
                                      "and minimum range is %f. %s" % (keys[i], np.min(d), lo_i, msg), RuntimeWarning)
                if clip:
                    files[k] = np.clip(files[k], lo_i, hi_i)

    def _resample(self):
        """Cubic resampling of every trajectory from traj_dt to control_dt (:230-287)."""
        L = self.trajectory_length
        x = np.arange(L)
        x_new = np.linspace(0, L - 1, round(L * (self.traj_dt / self.control_dt)), endpoint=True)
        new = []
        for j in range(self.number_of_trajectories):
            block = np.array([col[j] for col in self.trajectories])                 # [n_keys, L]
            new.append(interpolate.interp1d(x, block, kind="cubic", axis=1)(x_new))
        self.trajectories = [np.array([t[i] for t in new]) for i in range(len(self.trajectories))]
        sp = [0]
        for j in range(self.number_of_trajectories):
            sp.append(sp[-1] + len(self.trajectories[0][j]))
        self.split_points = np.array(sp)

    # ------------------------------------------------------------------ views
    @property
    def table(self):
        """[n_keys, n_traj, len] float64: what oly_traj_upload takes."""
        return np.ascontiguousarray(np.array(self.trajectories, dtype=np.float64))

    @property
    def number_obs_trajectory(self):
        return len(self.trajectories)

    @property
    def trajectory_length(self):
        return self.trajectories[0].shape[1]

    @property
    def number_of_trajectories(self):
        return self.trajectories[0].shape[0]

    def flattened_trajectories(self):
        out = []
        for obs in self.trajectories:
            if obs.ndim == 2:
                out.append(obs.reshape((-1, 1)))
            elif obs.ndim == 3:
                out.append(obs.reshape((-1, obs.shape[2])))
            else:
                raise ValueError("Unsupported shape of observation %s." % (obs.shape,))
        return out

    def create_dataset(self, ignore_keys=None, state_callback=None, state_callback_params=None):
        """states / next_states / absorbing / last for imitation learning (:129-193)."""
        data = dict(zip(self.keys, deepcopy(self.flattened_trajectories())))
        for k in ignore_keys or []:
            del data[k]
        states = np.concatenate(list(data.values()), axis=1)
        if state_callback is not None:
            states = np.array([state_callback(s, **state_callback_params) for s in states])
        last = np.zeros(len(states))
        last[self.split_points[1:] - 1] = 1.0
        out = dict(states=states[:-1], next_states=states[1:], absorbing=np.zeros(len(states) - 1), last=last)
        if self._traj_info is not None:
            out["info"] = np.array([[lab] * self.trajectory_length for lab in self._traj_info]).reshape(-1)
        return out

    # ------------------------------------------------------------------ single-env cursor (host)
    def _get_subtraj(self, i):
        return [obs[i].copy() for obs in self.trajectories]

    def _pick(self, given, upper):
        """Caller's index (the reference only bounds it by <=, trajectory.py:307,314) or one draw from
        numpy's global stream."""
        if given is None:
            return np.random.randint(0, upper)
        if not 0 <= given <= upper:
            raise AssertionError(f"index {given} outside [0, {upper}]")
        return given

    def reset_trajectory(self, substep_no=None, traj_no=None):
        """Select (trajectory, step), drawing whichever is None (trajectory first, then step: the
        reference's draw order), and re-zero x / y at that step on a copy of the sub-trajectory
        (:289-323).  Returns the sample at the cursor."""
        self.traj_no = self._pick(traj_no, self.number_of_trajectories)
        self.subtraj_step_no = self._pick(substep_no, self.trajectory_length)
        self.subtraj = self._get_subtraj(self.traj_no)
        for planar in (0, 1):                                  # keys 0 and 1 are the pelvis x and y
            self.subtraj[planar] -= self.subtraj[planar][self.subtraj_step_no]
        return self._row(self.subtraj_step_no, copy=False)

    def _row(self, k, copy=True):
        if copy:
            return [np.array(col[k].copy()).flatten() for col in self.subtraj]
        return [col[k] for col in self.subtraj]

    def get_current_sample(self):
        return self._row(self.subtraj_step_no)

    def get_next_sample(self):
        """Advance the cursor; None once it has walked off the end (:389-401)."""
        self.subtraj_step_no += 1
        return None if self.subtraj_step_no == self.trajectory_length else self._row(self.subtraj_step_no)

    def get_idx(self, key):
        return self.keys.index(key)

    def get_from_sample(self, sample, key):
        assert len(sample) == len(self.keys)
        return sample[self.get_idx(key)]


def synthetic_h1_trajectory_files(spec, n_traj=2, length=1000, seed=0, traj_dt=0.002):
    """SURVEY config 1 stand-in for the (absent) mocap datasets: smooth sinusoids inside the
    joint ranges with analytic derivatives, forward speed 1.25 m/s, keyed like the reference
    .npz wire format (q_*/dq_* + split_points)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_traj * length) * traj_dt
    t = t - np.repeat(np.arange(n_traj) * length * traj_dt, length)
    out = {}
    names = [k[2:] for k in spec.obs_keys[:spec.n_pos]]
    for i, name in enumerate(names):
        w = rng.uniform(2.0, 8.0)
        ph = rng.uniform(0, 2 * np.pi)
        lo, hi = spec.joint_lo[i], spec.joint_hi[i]
        if name == "pelvis_tx":
            q, dq = 1.25 * t + 0.01 * np.sin(w * t), 1.25 + 0.01 * w * np.cos(w * t)
        elif name == "pelvis_tz":
            q, dq = 0.02 * np.sin(w * t + ph), 0.02 * w * np.cos(w * t + ph)
        elif name == "pelvis_ty":
            q, dq = -0.05 + 0.02 * np.sin(w * t + ph), 0.02 * w * np.cos(w * t + ph)
        elif name.startswith("pelvis_"):
            q, dq = 0.05 * np.sin(w * t + ph), 0.05 * w * np.cos(w * t + ph)
        else:
            mid, amp = 0.5 * (lo + hi), 0.35 * (hi - lo)
            q, dq = mid + amp * np.sin(w * t + ph), amp * w * np.cos(w * t + ph)
        out["q_" + name], out["dq_" + name] = q, dq
    out["split_points"] = np.arange(n_traj + 1) * length
    return out
