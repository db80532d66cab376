"""Multi-rank path on a one-GPU box: two rank processes share cuda:0 (gloo rendezvous), each runs
the real config-5 tail on its env shard.  The driver's 8-GPU run uses the same code over RCCL."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_config5_two_ranks_share_one_gpu(tmp_path):
    """oly_return_scan_stats on each shard -> ONE all-gather of the device triples ->
    oly_adv_normalize_parts: both ranks hold bit-identical [2,3] statistics; returns and raw
    advantages equal the single-process block bit for bit (environments are independent); the
    sharded statistics equal the single-process ones to 1e-12 (different but fixed summation
    trees), the normalised advantages to 1 f32 ulp."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), str(tmp_path)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["parts"], r1["parts"]) and r0["parts"].shape == (2, 3)      # same on every rank
    assert np.array_equal(r0["parts"][0], r0["local"]) and np.array_equal(r0["parts"][1], r1["local"])
    assert (r0["lo"], r0["hi"], r1["lo"], r1["hi"]) == (0, 4096, 4096, 8192)
    ret = np.concatenate([r0["ret"], r1["ret"]], axis=1)
    assert np.array_equal(ret, r0["single_ret"])
    tot = r0["parts"][0] + r0["parts"][1]
    assert tot[0] == r0["single_stats"][0] == 400 * 8192
    np.testing.assert_allclose(tot, r0["single_stats"], rtol=1e-12)
    got = np.concatenate([r0["adv_norm"], r1["adv_norm"]], axis=1)
    want = r0["single_adv_norm"]
    assert np.abs(got - want).max() <= np.spacing(np.abs(want).max().astype(np.float32))
    a = r0["single_adv"].astype(np.float64)
    np.testing.assert_allclose(want, (a - a.mean()) / (a.std(ddof=1) + 1e-5), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("n0,n1", [(64, 64), (64, 48)])
def test_ppo_train_two_ranks_share_one_gpu(tmp_path, n0, n1):
    """PPO.train with world 2: the replicas of the ONE learner stay bit-identical through two iterations
    (parameter broadcast from rank 0, row-weighted gradient all-reduce per minibatch, globally reduced evaluation
    return driving highest_reward / the anneal), also when the shards differ in size (every rank enters the same
    number of collectives: no hang), and only rank 0 writes logs and checkpoints."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "tests", "_dist_train_worker.py"), str(tmp_path), str(n0), str(n1)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert not np.array_equal(r0["first"], r1["first"])                  # they started from different weights ...
    assert np.array_equal(r0["params"], r1["params"])                    # ... and end bit-identical
    assert np.isfinite(r0["params"]).all() and np.isfinite(r0["losses"]).all()
    assert float(r0["highest"]) == float(r1["highest"]) > -1e9
    assert np.array_equal(r0["eval_returns"], r1["eval_returns"]) and np.array_equal(r0["ep_returns"], r1["ep_returns"])
    assert int(r0["total_steps"]) == 2 * 8 * n0 and int(r1["total_steps"]) == 2 * 8 * n1
    # header + ONE line per iteration in each log (not one per rank); checkpoints exist (rank 0 wrote them)
    for f in ("train.txt", "eval.txt"):
        assert len(open(tmp_path / f).read().splitlines()) == 3, f
    assert os.path.exists(tmp_path / "actor.pt") and os.path.exists(tmp_path / "critic_1.pt")


@pytest.mark.parametrize("config", [5, 2])
def test_bench_starts_its_own_ranks(config):
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two rank processes itself and reports
    n_gpus = 2 (here both on cuda:0 over gloo; on an 8-GPU node one per GPU over RCCL)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--share-device", "--config", str(config), "--steps", "5", "--warmup", "2",
                          "--T", "100", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["roofline"]["bound"] == "hbm"
    if config == 5:
        assert set(line["stages_ms"]) >= {"scan_with_fused_stats", "all_gather_host_wall", "normalise"}
        assert line["config"]["ranks_seen"] == 2
    else:
        # the default (config 2) line of a multi-rank run also measures the design's ONE exchange step in the same
        # processes: the process group saw both ranks, the all-gather returned one triple per rank
        tail = line["config5_tail"]
        assert "error" not in tail, tail
        assert tail["ranks_seen"] == 2 and tail["gathered_rows"] == 2 and tail["backend"] == "gloo"
        assert set(tail["stages_ms"]) >= {"scan_with_fused_stats", "all_gather_host_wall", "normalise"}
        assert tail["value"] > 0 and 0 < tail["roofline"]["frac"] < 1
        assert "configs" not in line and "per_step" not in line


@pytest.mark.parametrize("config", [2, 5])
def test_bench_five_ranks_in_the_drivers_command_shape(config):
    """The driver's multi-GPU command shape with as many ranks as one box may put on its GPU next to the test process
    itself (the pool admits six processes on a GPU: this pytest process, whose earlier tests hold a context, + five
    ranks; the 8-rank all-gather / tree sum itself runs on the CPU in
    test_host_cpu.py::test_adv_stats_allgather_two_ranks_gloo[8]): `bench.py --gpus 5` starts five ranks, the process
    group sees five, the ONE all-gather returns five triples, the default command's line carries config5_tail."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "5", "--backend", "gloo",
                          "--share-device", "--config", str(config), "--steps", "3", "--warmup", "1",
                          "--T", "50", "--N", "1024", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 5 and line["scaling"] == "weak" and line["value"] > 0
    tail = line if config == 5 else line["config5_tail"]
    assert "error" not in tail, tail
    if config == 5:
        assert line["config"]["ranks_seen"] == 5
    else:
        assert tail["ranks_seen"] == 5 and tail["gathered_rows"] == 5 and tail["backend"] == "gloo"
    assert set(tail["stages_ms"]) >= {"scan_with_fused_stats", "all_gather_host_wall", "normalise"}


def test_bench_launcher_ends_when_a_rank_dies_before_the_rendezvous():
    """Rank 1 exits before init_process_group; rank 0 is then blocked in the rendezvous.  The launcher polls all its
    children, ends the survivors and fails, instead of waiting on rank 0 forever."""
    import time
    env = dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")},
               OLY_BENCH_TEST_DIE_RANK="1")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--share-device", "--steps", "2", "--warmup", "1", "--T", "50", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode != 0 and "rank 1 failed first" in out.stderr, out.stderr[-2000:]
    assert time.time() - t0 < 120


def test_bench_default_line_carries_every_baseline_config():
    """One GPU, the driver's command shape: the config-2 headline plus `configs` = BASELINE configs 3 / 4 / 5, each
    with its own roofline (and, without --no-cpu-baseline, an oracle cpu_baseline), none of them an error."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["roofline"]["traffic_source"].startswith("profiles/traffic_k1.json")
    c = line["configs"]
    assert set(c) == {"config3_a3_ppo_sampling", "config4_vail_reward", "config5_tail_world1"}
    assert not any("error" in v for v in c.values()), c
    c3 = c["config3_a3_ppo_sampling"]
    assert c3["sampling"]["persistent"]["us_per_vec_step"] < c3["sampling"]["graph_replay"]["us_per_vec_step"]
    for k, bound in (("a3_rollout_kernel(K13)", "mfma"), ("mlp_forward16_kernel(K11)", "mfma"), ("a3_vec_kernel(K10)", "hbm")):
        r = c3["kernels"][k]["roofline"]
        assert r["bound"] == bound and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c4 = c["config4_vail_reward"]
    assert c4["shapes"]["T400xN4096"]["roofline"]["frac"] > 0.3 and c4["shapes"]["B4096"]["disc_forward_us"] < 40
    # round 4: a WHOLE config-3 iteration (sampling + kernel update) and config 4's three-stage pipeline in the line
    it = c3["iteration"]
    assert "error" not in it and it["with_mirror_loss"]["losses_finite"] and it["without_mirror_loss"]["losses_finite"]
    assert it["with_mirror_loss"]["env_steps_per_s"] > 5e6 and it["with_mirror_loss"]["updates"] == 75
    kr = it["kernels"]["without_mirror_loss"]["minibatch_65536"]
    assert kr["roofline"]["bound"] == "mfma" and 0.2 < kr["roofline"]["frac"] < 1 and kr["update_ms"] < 1.5
    assert it["kernels"]["without_mirror_loss"]["minibatch_64"]["update_ms"] < 0.6     # the torch graph replay took 0.60
    pipe = c4["shapes"]["B4096"]["pipeline"]
    assert set(pipe["stages_us"]) >= {"reward_step", "gae_scan_with_statistics", "normalise"} and pipe["us"] < 200
    assert set(c["config5_tail_world1"]["stages_ms"]) >= {"scan_with_fused_stats", "normalise"}
    assert line["per_step"]["config3_a3_ppo_sampling"] == c3["sampling"]


def test_bench_fails_loudly_when_a_rank_fails():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device",
                          "--config", "5", "--steps", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0                        # --share-device without gloo is refused before any GPU work


def _torchrun_one_rank(script_args, timeout=600):
    env = dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")},
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), *script_args],
                          capture_output=True, text=True, timeout=timeout, env=env)


def test_rccl_collectives_with_one_rank():
    """The box has one GPU, so RCCL cannot be run across ranks here; a ONE-rank "nccl" group still sends
    every collective of the multi-GPU path through RCCL on the device (what the driver's N > 1 runs use)."""
    out = _torchrun_one_rank([os.path.join(ROOT, "tests", "_rccl_one_rank_worker.py")])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["backend"] == "nccl" and line["world"] == 1
    assert line["stats"][0][0] == 50 * 1024 and line["norm_err"] < 1e-5
    assert line["grads_unchanged"] and line["max"] == 1.5


@pytest.mark.parametrize("config", [2, 5])
def test_bench_under_torchrun_uses_rccl(config):
    """The driver's launch line with one rank: bench.py forms an "nccl" group (barrier with device_ids,
    MAX over ranks, config 5's all-gather) and prints the contract's JSON line."""
    out = _torchrun_one_rank([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", str(config), "--steps", "5",
                              "--warmup", "2", "--T", "100", "--no-cpu-baseline", "--no-per-step", "--no-configs"])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["parallelism"].startswith("env-sharded x1")
