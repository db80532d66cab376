"""CPU tests of the host side: spec tables, trajectory preprocessing, wrappers, registry,
the C ABI surface, the no-fallback rule and the multi-process statistics exchange."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from olympic_hip import _abi, specs
from olympic_hip.trajectory import Trajectory

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_H1_XML = "/root/reference/olympic_mujoco/environments/data/unitree_h1/h1.xml"


# ------------------------------------------------------------------------------ specs
def test_h1_spec_matches_reference_tables(golden):
    g = golden("h1_tables.npz")
    sp = specs.unitree_h1("walk")
    assert sp.obs_keys == list(g["spec_keys"])
    assert sp.joint_names == list(g["joint_names"])
    assert (sp.nq, sp.nv, sp.n_pos) == (int(g["nq"]), int(g["nv"]), int(g["n_pos"]))
    assert np.array_equal(sp.qpos_adr, g["qpos_perm"]) and np.array_equal(sp.qvel_adr, g["qvel_perm"])
    assert sp.action_names == list(g["action_names"])
    assert np.array_equal(sp.act_to_ctrl, g["act_to_ctrl"])
    assert np.array_equal(sp.joint_lo, g["joint_lo"]) and np.array_equal(sp.joint_hi, g["joint_hi"])
    assert np.array_equal(sp.ctrl_lo, g["ctrl_lo"]) and np.array_equal(sp.ctrl_hi, g["ctrl_hi"])
    assert sp.reward_idx == int(g["x_vel_idx"]) == 15
    assert sp.n_obs == 32 and sp.n_act == 11
    assert np.array_equal(sp.act_delta, np.full(11, 0.95)) and not sp.act_mean.any()
    # SURVEY 8a tables
    assert sp.qpos_adr.tolist() == [0, 1, 2, 3, 4, 5, 16, 13, 12, 11, 14, 15, 8, 7, 6, 9, 10]
    assert sp.act_to_ctrl.tolist() == [10, 7, 6, 5, 8, 9, 2, 1, 0, 3, 4]


def test_h1_fall_thresholds_are_the_reference_doubles():
    sp = specs.unitree_h1("walk")
    pi = np.pi
    assert [t[1:] for t in sp.fall_tests] == [(-0.3, 0.1), (-pi / 4.5, pi / 12), (-pi / 12, pi / 8),
                                             (-pi / 8, pi / 8)]
    assert [sp.obs_idx(t[0]) for t in sp.fall_tests] == [0, 1, 2, 3]
    assert specs.unitree_h1("run").target_velocity == 2.5
    with pytest.raises(ValueError):
        specs.unitree_h1("fly")
    with pytest.raises(NotImplementedError):
        specs.unitree_h1("walk", reward_type="bogus")


@pytest.mark.skipif(not os.path.exists(REF_H1_XML), reason="reference MJCF not present (GPU box)")
def test_tables_rederived_from_mjcf():
    from olympic_hip.mjcf_tables import tables_from_mjcf
    arms = specs._H1_ARM_JOINTS
    t = tables_from_mjcf(REF_H1_XML, removed_joints=arms, removed_motors=[a + "_actuator" for a in arms])
    sp = specs.unitree_h1("walk")
    assert [j[0] for j in t["joints"]] == sp.joint_names and t["nq"] == sp.nq
    adr = {j[0]: j[1] for j in t["joints"]}
    assert [adr[k[2:]] for k in sp.obs_keys[:sp.n_pos]] == sp.qpos_adr.tolist()
    mid = {m[0]: i for i, m in enumerate(t["motors"])}
    assert [mid[a] for a in sp.action_names] == sp.act_to_ctrl.tolist()
    assert all(m[3:] == (-0.95, 0.95) for m in t["motors"])
    full = tables_from_mjcf(REF_H1_XML)
    assert [(j[0], j[3], j[4]) for j in full["joints"]] == specs._H1_JOINTS
    assert [(m[1], m[2]) for m in full["motors"]] == specs._H1_MOTORS


def test_a3_spec_constants(golden):
    g = golden("a3_task.npz")
    sp = specs.A3Spec()
    assert sp.period == int(g["period"]) == 88 and sp.delay_frames == int(g["delay_frames"]) == 30
    assert np.array_equal(sp.motor_offset, g["motor_offset"])
    s = golden("symmetry.npz")
    assert list(sp.mirrored_obs) == s["mirrored_obs"].tolist()
    assert list(sp.mirrored_acts) == s["mirrored_acts"].tolist()
    assert list(sp.clock_inds) == s["clock_inds"].tolist()


# ------------------------------------------------------------------------------ trajectory
def _traj_from_golden(g):
    keys = list(g["keys"])
    files = {k: g["raw"][i] for i, k in enumerate(keys)}
    files["split_points"] = g["raw_split_points"]
    return Trajectory(keys=keys, low=g["low"], high=g["high"], joint_pos_idx=np.arange(17), traj_files=files,
                      traj_dt=float(g["traj_dt"]), control_dt=float(g["control_dt"]),
                      clip_trajectory_to_joint_ranges=True, warn=False)


def test_trajectory_preprocessing_matches_reference(golden):
    g = golden("trajectory.npz")
    tr = _traj_from_golden(g)
    assert np.array_equal(tr.table, g["table"])              # clip + split + cubic resample: bit-exact
    assert np.array_equal(tr.split_points, g["split_points"])
    assert tr.trajectory_length == 50 and tr.number_of_trajectories == 2
    for (sub, tno), exp in zip(g["resets"], g["reset_samples"]):
        s = tr.reset_trajectory(int(sub), int(tno))
        assert np.array_equal(np.array(s, dtype=np.float64).ravel(), exp)
    sub, tno = g["rnd_reset"]
    s = tr.reset_trajectory(int(sub), int(tno))
    walk = [np.array(s, dtype=np.float64).ravel()]
    while True:
        s = tr.get_next_sample()
        if s is None:
            break
        walk.append(np.concatenate(s))
    assert np.array_equal(np.array(walk), g["walk"])
    ds = tr.create_dataset(ignore_keys=["q_pelvis_tx", "q_pelvis_tz"])
    for k in ("states", "next_states", "absorbing", "last"):
        assert np.array_equal(ds[k], g["ds_" + k]), k


def test_trajectory_argument_errors(golden):
    g = golden("trajectory.npz")
    keys = list(g["keys"])
    with pytest.raises(AssertionError):
        Trajectory(keys=keys, traj_path=None, traj_files=None)
    files = {k: g["raw"][i] for i, k in enumerate(keys)}
    files["split_points"] = np.array([0, 100, 500])          # unequal lengths
    with pytest.raises(AssertionError):
        Trajectory(keys=keys, traj_files=files, warn=False)
    files["split_points"] = g["raw_split_points"]
    files[keys[3]] = files[keys[3]][:-1]
    with pytest.raises(AssertionError):
        Trajectory(keys=keys, traj_files=files, warn=False)


def test_synthetic_trajectory_is_in_range_and_not_fallen():
    from olympic_hip.trajectory import synthetic_h1_trajectory_files
    sp = specs.unitree_h1("walk")
    files = synthetic_h1_trajectory_files(sp, n_traj=2, length=1000, seed=0)
    for i, k in enumerate(sp.obs_keys[:sp.n_pos]):
        if i >= 6:
            assert files[k].min() >= sp.joint_lo[i] and files[k].max() <= sp.joint_hi[i]
    for key, lo, hi in sp.fall_tests:
        assert files[key].min() > lo and files[key].max() < hi


# ------------------------------------------------------------------------------ wrappers
def test_symmetry_wrappers(golden):
    from olympic_hip.wrappers import SymmetricEnv, WrapEnv, _get_symmetry_matrix
    g = golden("symmetry.npz")
    assert np.array_equal(_get_symmetry_matrix(g["mirrored_obs"].tolist()), g["obs_matrix"])
    assert np.array_equal(_get_symmetry_matrix(g["mirrored_acts"].tolist()), g["act_matrix"])

    class Dummy:
        base_obs_len = 41
    env = SymmetricEnv(Dummy, mirrored_obs=g["mirrored_obs"].tolist(), mirrored_act=g["mirrored_acts"].tolist(),
                       clock_inds=g["clock_inds"].tolist())
    obs, act = torch.tensor(g["obs"]), torch.tensor(g["act"])
    assert np.array_equal(env.mirror_observation(obs).numpy(), g["obs_mirror"])     # signed permutation: exact
    assert np.array_equal(env.mirror_action(act).numpy(), g["act_mirror"])
    np.testing.assert_allclose(env.mirror_clock_observation(obs).numpy(), g["obs_mirror_clock"], rtol=0, atol=6e-7)  # f32 sin(arcsin(x)+pi), numpy vs torch
    with pytest.raises(AssertionError):
        SymmetricEnv(Dummy, mirrored_obs=[0.1, 1], mirrored_act=None)

    class One:
        def step(self, a):
            return np.zeros(3), 1.5, False, {}

        def reset(self):
            return np.ones(3)
    w = WrapEnv(One)
    s, r, d, i = w.step(np.zeros((1, 2)))
    assert s.shape == (1, 3) and r.shape == (1,) and d.shape == (1,) and w.reset().shape == (1, 3)


# ------------------------------------------------------------------------------ registry
def test_registry_and_name_validation():
    from olympic_hip.envs import LocoEnvBase, UnitreeH1, ValidTaskConf, check_validity_task_mode_dataset
    names = LocoEnvBase.get_all_task_names()
    assert "UnitreeH1.walk.real" in names and "UnitreeH1.carry.perfect" not in names
    with pytest.raises(ValueError, match="does not exit"):
        UnitreeH1.generate("jump", "real")
    with pytest.raises(ValueError, match="Dataset type"):
        UnitreeH1.generate("walk", "imaginary")
    with pytest.raises(ValueError, match="not combineable"):
        check_validity_task_mode_dataset("UnitreeH1", "carry", None, "perfect", *UnitreeH1.valid_task_confs.get_all())
    with pytest.raises(KeyError):
        LocoEnvBase.make("NoSuchRobot.walk")
    c = ValidTaskConf(tasks=["a", "b"], data_types=["x"])
    assert c.get_all_combinations() == [{"task": "a", "data_type": "x"}, {"task": "b", "data_type": "x"}]


# ------------------------------------------------------------------------------ C ABI
def _header_functions():
    txt = open(os.path.join(ROOT, "include", "olympic_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(oly_[a-z0-9_]+)\s*\(", txt))


def test_abi_table_matches_header():
    assert _header_functions() == set(_abi.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from olympic_hip import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        pytest.fail(f"{_ffi.LIB_PATH} missing: run python __graft_entry__.py build")
    L = _ffi.lib()                      # binds every symbol; AttributeError on a missing one
    for name in _abi.SIGNATURES:
        assert hasattr(L, name)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _ffi.LIB_PATH], text=True)
    exported = set(re.findall(r" T (oly_[a-z0-9_]+)", out))
    assert set(_abi.SIGNATURES) <= exported
    assert L.oly_strerror(_abi.OLY_EINVAL) == b"invalid argument"
    assert b"gfx950" in L.oly_version()
    # the ABI version of the header, the library and the binding table agree (a mismatch makes lib() refuse to load)
    hdr = int(re.search(r"#define OLY_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "olympic_hip.h")).read()).group(1))
    assert hdr == _abi.ABI_VERSION == int(L.oly_abi_version())


def test_struct_layouts_match_the_header():
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    import tempfile
    fields = {"oly_il_model": _abi.IlModel, "oly_a3_model": _abi.A3Model, "oly_a3_inputs": _abi.A3Inputs,
              "oly_a3_state": _abi.A3State, "oly_a3_readback": _abi.A3Readback, "oly_il_contacts": _abi.IlContacts,
              "oly_a3_blocks": _abi.A3Blocks, "oly_a3_reset_record": _abi.A3ResetRecord,
              "oly_a3_rollout": _abi.A3Rollout, "oly_contact_record": _abi.ContactRecord,
              "oly_ppo_update": _abi.PPOUpdate, "oly_adam_net": _abi.AdamNet,
              "oly_ppo_adam": _abi.PPOAdam}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/olympic_hip.h"', "int main(){"]
    for cname, cls in fields.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f, _ in cls._fields_:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(src, "w").write("\n".join(lines))
        subprocess.check_call(["gcc", "-o", exe, src])
        out = dict(l.split() for l in subprocess.check_output([exe], text=True).splitlines())
    for cname, cls in fields.items():
        assert int(out[cname]) == ctypes.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(out[f"{cname}.{f}"]) == getattr(cls, f).offset, (cname, f)


def test_no_cpu_fallback_and_no_oracle_in_product():
    from olympic_hip._ffi import Context, OlyError
    if not torch.cuda.is_available():
        with pytest.raises(OlyError, match="no CPU fallback"):
            Context(0)
    pkg = os.path.join(ROOT, "olympics-mujoco_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oly_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


# ------------------------------------------------------------------------------ multi-process
def test_shard_range():
    from olympic_hip.dist import shard_range
    parts = [shard_range(32768, r, 8) for r in range(8)]
    assert parts[0] == (0, 4096) and parts[-1] == (28672, 32768)
    parts = [shard_range(10, r, 4) for r in range(4)]
    assert [b - a for a, b in parts] == [3, 3, 2, 2] and parts[-1][1] == 10


@pytest.mark.parametrize("world", [2, 8])
def test_adv_stats_allgather_two_ranks_gloo(tmp_path, world):
    """world_size 2 and 8 (the driver's 8-GPU shape) over gloo: each rank reduces its env shard to (count, sum, sumsq),
    ONE all-gather of one triple per rank, every rank normalises with the same global statistics = the single-process
    result (PPO ddof 1 / 1e-5 and GAIL ddof 0 / 1e-8); the eight near-equal shards of 1001 environments tile them."""
    script = os.path.join(ROOT, "tests", "_dist_worker.py")
    import socket
    with socket.socket() as sk:                      # a port that is free right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), script, str(tmp_path)],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stdout + out.stderr
    rs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    r0 = rs[0]
    for r in rs[1:]:
        assert np.array_equal(r0["total"], r["total"]) and np.array_equal(r0["parts"], r["parts"])   # identical on every rank
    assert r0["parts"].shape == (world, 3) and r0["parts"][:, 0].sum() == r0["total"][0]              # one triple per rank
    adv = np.load(tmp_path / "full.npy")
    for ddof, eps, key in ((1, 1e-5, "ppo"), (0, 1e-8, "gail")):
        ref = (adv - adv.mean()) / (adv.std(ddof=ddof) + eps)
        got = np.concatenate([r[key] for r in rs], axis=1)
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6)
    assert r0["total"][0] == adv.size
    # replicated learner: identical weights after the broadcast, gradients = mean over ranks
    for r in rs[1:]:
        assert np.array_equal(r0["w_after_broadcast"], r["w_after_broadcast"])
    torch.manual_seed(100)
    net0 = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 2))
    assert np.array_equal(r0["w_after_broadcast"], torch.cat([p.detach().reshape(-1) for p in net0.parameters()]).numpy())
    mean_rank = (world + 1) / 2.0
    want = np.concatenate([np.full(n, mean_rank * (i + 1), np.float32) for i, n in enumerate((35, 7, 14, 2))])
    for r in rs:
        np.testing.assert_allclose(r["g_after_allreduce"], want, rtol=1e-6)
    # ranks with DIFFERENT minibatch sizes (3, 5, 7, ... rows): the row-weighted mean is the concatenated batch's gradient
    for r in rs:
        assert np.array_equal(r["g_weighted"], r0["g_weighted"]) and np.array_equal(r["g_weighted_flat"], r0["g_weighted"])
        np.testing.assert_allclose(r["g_weighted"], r["g_concatenated"], rtol=2e-5, atol=1e-7)


# ------------------------------------------------------------------------------ A3 host side
def test_clock_lut_matches_create_phase_reward(golden):
    from olympic_hip.a3 import clock_lut
    g = golden("a3_task.npz")
    lut = clock_lut(0.75, 0.35, 0.1, "grounded", 40.0)
    assert lut.shape == (4, 88)
    assert np.array_equal(lut, g["clock_lut"])           # same knots, same scipy PCHIP: bit-exact


def test_walking_task_reset_matches_reference(golden):
    """Same numpy global-stream draws as WalkingTask.reset (walking_task.py:321-397)."""
    from olympic_hip.a3 import WalkingTaskReset, yaw_of_quat
    g = golden("a3_task.npz")
    reset = WalkingTaskReset(specs.A3Spec())
    for e in range(len(g["mode"])):
        np.random.seed(1000 + e)                          # the seed gen_golden.py used per env
        r = reset(g["reset_lfoot"][e], g["reset_rfoot"][e], yaw_of_quat(g["reset_root_quat"][e]),
                  int(g["iter_count"][e]))
        assert r["mode"] == g["mode"][e] and r["phase"] == g["phase0"][e]
        assert r["seq_len"] == g["seq_len"][e] and (r["t1"], r["t2"]) == (g["t1_0"][e], g["t2_0"][e])
        np.testing.assert_allclose(r["sequence"], g["sequence"][e, :r["seq_len"]], rtol=1e-13, atol=1e-14)


# ------------------------------------------------------------------------------ PPO losses
def test_ppo_update_policy_matches_reference(golden, tmp_path):
    """PPO.update_policy (rl/algos/ppo.py:232-282) on the reference's own actor/critic weights."""
    from olympic_hip.ppo import PPO, MLPCritic, MLPGaussianActor
    from olympic_hip.wrappers import SymmetricEnv
    g = golden("ppo_update.npz")

    def load(mod, tag):
        sd = {k[len(tag) + 1:]: torch.tensor(g[k]) for k in g.files if k.startswith(tag + ".")}
        mod.load_state_dict(sd)
        return mod
    std = torch.tensor(float(g["fixed_std"]))
    pi = load(MLPGaussianActor(41, 12, fixed_std=std), "pi")
    old = load(MLPGaussianActor(41, 12, fixed_std=std), "old")
    vf = load(MLPCritic(41), "vf")
    args = dict(gamma=0.99, lam=0.95, lr=1e-4, eps=1e-5, entropy_coeff=0.0, clip=float(g["clip"]), minibatch_size=64,
                epochs=3, max_traj_len=400, use_gae=False, num_procs=4, max_grad_norm=0.05, mirror_coeff=0.4,
                eval_freq=100)
    ppo = PPO(args, str(tmp_path))
    assert open(ppo.train_fn).read() == "ep_returns,ep_lens\n" and ppo.batch_size == 1600
    ppo.policy, ppo.critic, ppo.old_policy = pi, vf, old

    class Dummy:
        base_obs_len = 41
    sym = SymmetricEnv(Dummy, mirrored_obs=g["mirrored_obs"].tolist(), mirrored_act=g["mirrored_acts"].tolist(),
                       clock_inds=[31, 32])
    t = lambda k: torch.tensor(g[k])
    out = ppo.update_policy(t("obs"), t("act"), t("ret"), t("adv"), 1, sym.mirror_clock_observation, sym.mirror_action)
    names = ("actor_loss", "entropy_penalty", "critic_loss", "approx_kl_div", "mirror_loss", "clip_fraction")
    for n, v in zip(names, out):
        v = v.detach() if torch.is_tensor(v) else v
        np.testing.assert_allclose(float(v), float(g[n]), rtol=2e-5, atol=2e-7, err_msg=n)
    out2 = ppo.update_policy(t("obs"), t("act"), t("ret"), t("adv"), 1)
    assert float(out2[4]) == float(g["mirror_loss_none"]) == 0.0


# ------------------------------------------------------------------------------ Atlas / Talos
@pytest.mark.parametrize("robot", ["atlas", "talos"])
def test_atlas_talos_tables_and_has_fallen(golden, oracle, robot):
    """The IL kernel is table-driven: Atlas and Talos are data.  Tables and _has_fallen (first
    violated condition of the elif chain) against the reference classes."""
    g = golden(f"{robot}_tables.npz")
    fn = getattr(specs, robot)
    for tag, kw in (("default", {}), ("all_joints", dict(disable_arms=False, disable_back_joint=False))):
        sp = fn("walk", **kw)
        assert sp.obs_keys == list(g[f"{tag}.keys"])
        assert (sp.nq, sp.n_pos) == (int(g[f"{tag}.nq"]), int(g[f"{tag}.n_pos"]))
        assert np.array_equal(sp.qpos_adr, g[f"{tag}.qpos_perm"])
        assert np.array_equal(sp.act_to_ctrl, g[f"{tag}.act_to_ctrl"])
        assert sp.fall_names == list(g[f"{tag}.cond_names"])[:len(sp.fall_names)]
        obs = g[f"{tag}.obs"]
        full = np.concatenate([np.zeros((len(obs), 2)), obs], axis=1)
        from olympic_hip.synthetic import h1_rows_from_full
        qpos, qvel = h1_rows_from_full(sp, full)
        o = oracle.il_step(sp, qpos[None], qvel[None], None, np.zeros(len(obs)), obs_f64=True)
        assert np.array_equal(o["obs"][0], obs)
        assert np.array_equal(o["absorbing"][0].astype(bool), g[f"{tag}.fallen"])
        assert np.array_equal(o["fall_code"][0], g[f"{tag}.code"])


def test_atlas_talos_registered():
    from olympic_hip.envs import Atlas, LocoEnvBase, Talos
    names = LocoEnvBase.get_all_task_names()
    assert "Atlas.walk.real" in names and "Talos.walk.perfect" in names
    with pytest.raises(ValueError):
        Atlas.generate("run", "real")
    assert Talos._default_back is False and Atlas._default_back is True


# ------------------------------------------------------------------------------ sanitizers
def test_oracle_golden_suite_under_asan_ubsan():
    """The oracle's golden tests once more with the C code built -fsanitize=address,undefined
    (GPU ASan does not exist on this pool; the CPU restatement shares the indexing logic)."""
    import shutil
    import subprocess
    import sys
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or shutil.which("make") is None:
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, OLY_ORACLE_ASAN="1", LD_PRELOAD=asan, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_oracle_golden.py")],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


# ------------------------------------------------------------------------------ discriminator losses
def test_discriminator_losses_match_reference(golden):
    """GailDiscriminatorLoss / VDBLoss (imitation_lib/utils/math.py) on the reference's own logits."""
    from olympic_hip.gail import VDBLoss, gail_discriminator_loss
    g = golden("vail_disc.npz")
    t = lambda k: torch.tensor(g[k])
    np.testing.assert_allclose(float(gail_discriminator_loss(t("d"), t("target"))), float(g["gail_loss"]), rtol=2e-6)
    vl = VDBLoss(info_constraint=0.1, lr_beta=1e-5)
    np.testing.assert_allclose(float(vl((t("d"), t("mu"), t("logvar")), t("target"))), float(g["vdb_loss"]), rtol=2e-6)
    np.testing.assert_allclose(float(vl._beta), float(g["vdb_beta_after"]), rtol=1e-6)


# ------------------------------------------------------------------------------ reset-record pool
def test_reset_records_follow_walking_task_reset():
    """The vectorised record draw (device-side resets) against the scalar WalkingTaskReset, which
    test_walking_task_reset_matches_reference pins to the reference: same distribution parameters, and
    for identical draws the same local step sequence."""
    from olympic_hip.a3 import WalkingTaskReset
    from olympic_hip.vecstep import draw_reset_records
    spec = specs.A3Spec(mass=41.5)
    rec = draw_reset_records(np.random.RandomState(0), 4000, spec, iter_count=7000)
    assert set(np.unique(rec["mode"])) == {_abi.MODE_STANDING, _abi.MODE_FORWARD}
    assert abs((rec["mode"] == _abi.MODE_STANDING).mean() - 0.2) < 0.03
    assert set(np.unique(rec["phase"])) == {0, 44} and abs((rec["phase"] == 44).mean() - 0.5) < 0.04
    fwd = rec[rec["mode"] == _abi.MODE_FORWARD]
    assert (fwd["seq_len"] == 20).all() and (rec[rec["mode"] == _abi.MODE_STANDING]["seq_len"] == 1).all()
    # a scalar reset with the same draws: replay them through a stub RandomState
    class Replay:
        def __init__(self, vals):
            self.vals = list(vals)

        def uniform(self, lo, hi):
            return self.vals.pop(0)

        def randint(self, lo, hi):
            return self.vals.pop(0)
    for r in fwd[:50]:
        first = abs(r["seq"][0, 1])
        zs = r["seq"][:, 2]
        step_h = zs[-1] / max(1, np.count_nonzero(np.diff(zs)))
        c = int(np.argmax(zs != 0)) - 1 if zs.any() else 2
        rr = WalkingTaskReset(spec, Replay([first, c]))            # _sequence draws uniform, then randint
        seq = rr._sequence(r["phase"], 88, step_size=0.3, step_gap=0.15, step_height=step_h, num_steps=20, lateral=False)
        np.testing.assert_allclose(np.array(seq)[:, :2], r["seq"][:, :2], rtol=0, atol=0)
        np.testing.assert_allclose(np.array(seq)[:, 2], zs, rtol=1e-12, atol=1e-15)
