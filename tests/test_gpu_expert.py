"""SURVEY 8(f)-2 on the device: Trajectory.create_dataset's states / next_states / absorbing / last and
the discriminator's expert minibatches (GAIL._fit_discriminator, gail_TRPO.py:176-202) served from the
device-resident trajectory table."""
import numpy as np
import pytest
import torch

from olympic_hip.trajectory import Trajectory

pytestmark = pytest.mark.gpu


def _traj_from_golden(g):
    keys = list(g["keys"])
    files = {k: g["raw"][i] for i, k in enumerate(keys)}
    files["split_points"] = g["raw_split_points"]
    return Trajectory(keys=keys, low=g["low"], high=g["high"], joint_pos_idx=np.arange(17), traj_files=files,
                      traj_dt=float(g["traj_dt"]), control_dt=float(g["control_dt"]),
                      clip_trajectory_to_joint_ranges=True, warn=False)


def test_expert_dataset_and_minibatches_from_the_device_table(golden):
    """(a) the dataset arrays = the reference's create_dataset output (trajectory.npz ds_*), bit-exact;
    (b) a minibatch drawn like mushroom's minibatch_generator (np.random.shuffle of the row indices, the
    first batch_size entries) and sliced like gail_TRPO.py:176-183 (rows, state mask, .astype(float32));
    (c) the device table the cursors (K4) read is the very same upload."""
    from olympic_hip.engine import Engine
    from olympic_hip.gail import ExpertDataset
    g = golden("trajectory.npz")
    tr = _traj_from_golden(g)
    eng = Engine(0)
    ds = ExpertDataset(eng, tr, ignore_keys=["q_pelvis_tx", "q_pelvis_tz"])
    arr = {k: v.cpu().numpy() for k, v in ds.arrays().items()}
    for k in ("states", "next_states", "absorbing", "last"):
        assert np.array_equal(arr[k], g["ds_" + k]), k                      # bit-exact float64
    assert ds.rows == len(g["ds_states"]) == 99
    # the reference's draw, restated: indexes = arange(size); np.random.shuffle(indexes); first batch
    B = 64
    np.random.seed(3)
    ref_idx = np.arange(ds.rows)
    np.random.shuffle(ref_idx)
    ref_idx = ref_idx[:B]
    idx = ds.shuffled_indices(B, np.random.RandomState(3))                   # same MT19937 stream as the global seed
    assert np.array_equal(idx.cpu().numpy(), ref_idx)
    st, nx = ds.minibatch(idx, want_next=True)
    assert np.array_equal(st.cpu().numpy(), g["ds_states"][ref_idx].astype(np.float32))
    assert np.array_equal(nx.cpu().numpy(), g["ds_next_states"][ref_idx].astype(np.float32))
    # a state mask folded into the column table (prepare_discrim_inputs, gail_TRPO.py:297-313)
    mask = np.array([0, 3, 4, 17, 31, 2])
    dm = ExpertDataset(eng, tr, ignore_keys=["q_pelvis_tx", "q_pelvis_tz"], state_mask=mask)
    assert np.array_equal(dm.minibatch(idx).cpu().numpy(), g["ds_states"][ref_idx][:, mask].astype(np.float32))
    # repeated indices and the last row
    rep = torch.tensor([98, 0, 98, 5, 5], dtype=torch.int64, device="cuda")
    assert np.array_equal(ds.minibatch(rep).cpu().numpy(), g["ds_states"][[98, 0, 98, 5, 5]].astype(np.float32))
    from olympic_hip._ffi import OlyError
    with pytest.raises(OlyError):
        ds.minibatch(torch.tensor([99], dtype=torch.int64, device="cuda"))    # 99 rows: 0 .. 98
    # same upload serves the trajectory cursor kernels
    cur = eng.traj_reset(torch.tensor([1], dtype=torch.int32, device="cuda"), torch.tensor([7], dtype=torch.int32, device="cuda"))
    assert np.array_equal(cur[3].cpu().numpy()[0, 2:], g["table"][2:, 1, 7])


def test_discriminator_trainer_draws_expert_batches_on_the_device(golden):
    """DiscriminatorTrainer fed by an ExpertDataset: no host array of demonstrations exists; one fit
    epoch = gather + concatenate + standardise + VDB loss step, finite and reproducible."""
    from olympic_hip.engine import Engine
    from olympic_hip.gail import DiscriminatorReward, DiscriminatorTrainer, ExpertDataset, VariationalDiscriminator, VDBLoss
    g = golden("trajectory.npz")
    eng = Engine(0)
    ds = ExpertDataset(eng, _traj_from_golden(g))
    losses = []
    for _ in range(2):
        torch.manual_seed(0)
        net = VariationalDiscriminator(32).cuda()
        reward = DiscriminatorReward(eng, net, state_mask=np.arange(32))
        tr = DiscriminatorTrainer(reward, ds, VDBLoss(0.1, 1e-5), n_epochs=3)
        assert tr.demo is None
        gen = torch.Generator(device="cuda").manual_seed(1)
        plcy = torch.randn(128, 32, device="cuda", generator=gen)
        losses.append(tr.fit(plcy, generator=gen))
    assert np.isfinite(losses[0]).all() and losses[0] == losses[1]
