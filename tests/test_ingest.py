"""Sample ingest ring: columns written in place give the reference's np.stack(axis=1) layout."""
import numpy as np
import pytest
import torch

from srl_amd import namedarray as na
from srl_amd.namedarray import recursive_aggregate
from srl_amd.runtime import synthetic
from srl_amd.runtime.ingest import SampleRing


def traj(seed, T=6, obs_spec=synthetic.CARTPOLE_OBS):
    arr = synthetic.make_sample_arrays(seed=seed, T=T, B=1, obs_spec=obs_spec, action_dims=3)
    return synthetic.to_sample_batch({k: v[:, 0] for k, v in arr.items()})


def leaves(x):
    return {k: v for k, v in na.flatten(x) if v is not None}


def test_columns_equal_reference_stacking_and_keep_wire_dtypes():
    trajs = [traj(s, obs_spec=synthetic.ATARI_OBS if False else synthetic.CARTPOLE_OBS) for s in range(5)]
    ring = SampleRing(trajs[0], batch_size=5, slots=2)
    done = [ring.put_column(t) for t in trajs]
    assert done[:-1] == [None] * 4 and done[-1] == 0 and ring.ready() == 1
    batch = ring.get()
    ref = recursive_aggregate(trajs, lambda xs: np.stack(xs, axis=1))  # reference base/buffer.py:120-121
    got, want = leaves(batch), leaves(ref)
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), k
        assert got[k].flags["C_CONTIGUOUS"] and got[k].shape[:2] == (7, 5)
    assert type(batch) is type(trajs[0]) and type(batch.action) is type(trajs[0].action)
    assert batch.info is None and batch.on_reset.dtype == np.uint8
    ring.release(batch)
    assert ring.ready() == 0


def test_wire_format_path_slot_rotation_and_backpressure():
    trajs = [traj(10 + s) for s in range(6)]
    ring = SampleRing(trajs[0], batch_size=2, slots=2)
    for t in trajs[:4]:
        ring.put_wire(na.dumps(t, "raw_bytes"))
    assert ring.ready() == 2
    with pytest.raises(BufferError):  # both slots full and unreleased: the producer is told, nothing is overwritten
        ring.put_column(trajs[4])
    first = ring.get()
    assert np.array_equal(first.reward, np.stack([trajs[0].reward, trajs[1].reward], 1))
    ring.release(first)
    ring.put_wire(na.dumps(trajs[4], "pickle_dict"))  # other encodings go through namedarray.loads
    ring.put_column(trajs[5])
    second = ring.get()
    assert np.array_equal(second.obs.obs, np.stack([trajs[2].obs.obs, trajs[3].obs.obs], 1))
    third = ring.get()
    assert third.metadata["ring_slot"] == first.metadata["ring_slot"]  # the released slot was refilled
    assert np.array_equal(third.action.x, np.stack([trajs[4].action.x, trajs[5].action.x], 1))
    with pytest.raises(LookupError):
        ring.get()
    with pytest.raises(KeyError):
        bad = traj(99)
        bad.obs = na.NamedArray(other=bad.obs.obs)
        ring.release(second)
        ring.put_column(bad)


def test_uint8_frames_stay_uint8():
    t = traj(3, T=2, obs_spec=synthetic.ATARI_OBS)
    ring = SampleRing(t, batch_size=3, slots=1)
    for s in range(3):
        ring.put_column(traj(3 + s, T=2, obs_spec=synthetic.ATARI_OBS))
    b = ring.get()
    assert b.obs.obs.dtype == np.uint8 and b.obs.obs.shape == (3, 3, 4, 84, 84)
    assert ring.nbytes() >= 3 * 3 * 4 * 84 * 84


def test_slot_is_published_only_after_every_column_is_written():
    """A slot whose last column has been CLAIMED may still have slower writers: it must not reach the consumer until
    every column's write has finished (threads release the GIL inside the strided numpy copies)."""
    import threading
    import time
    B = 8
    trajs = [traj(50 + s, T=4, obs_spec=synthetic.ATARI_OBS) for s in range(B)]
    ring = SampleRing(trajs[0], batch_size=B, slots=2)
    slow_started, let_go = threading.Event(), threading.Event()

    class SlowLeaf:  # the first column's frames take their time to arrive

        def __init__(self, arr):
            self.arr, self.shape, self.dtype = arr, arr.shape, arr.dtype

        def __array__(self, *a, **k):
            slow_started.set()
            let_go.wait(5.0)
            return self.arr

    slow = trajs[0]
    slow_obs = slow.obs.obs
    slow.obs = na.NamedArray(obs=SlowLeaf(slow_obs))
    th = threading.Thread(target=ring.put_column, args=(slow,))
    th.start()
    assert slow_started.wait(5.0)
    others = [threading.Thread(target=ring.put_column, args=(t,)) for t in trajs[1:]]
    for o in others:
        o.start()
    for o in others:
        o.join()
    time.sleep(0.05)
    assert ring.ready() == 0  # all columns claimed, the last claimed one written -- but column 0 is not finished
    with pytest.raises(LookupError):
        ring.get()
    let_go.set()
    th.join()
    assert ring.ready() == 1
    b = ring.get()
    assert np.array_equal(b.obs.obs[:, 0], slow_obs)
    cols = {tuple(b.reward[:, c, 0]) for c in range(B)}
    assert cols == {tuple(t.reward[:, 0]) for t in trajs}  # every trajectory landed in exactly one column


def test_bad_wire_message_leaves_no_hole():
    trajs = [traj(70 + s) for s in range(2)]
    ring = SampleRing(trajs[0], batch_size=2, slots=1)
    bad = traj(99)
    bad.obs = na.NamedArray(other=bad.obs.obs)
    with pytest.raises(KeyError):
        ring.put_wire(na.dumps(bad, "raw_bytes"))
    assert ring.put_wire(na.dumps(trajs[0], "raw_bytes")) is None
    assert ring.put_wire(na.dumps(trajs[1], "raw_bytes")) == 0  # the failed message claimed no column
    b = ring.get()
    assert np.array_equal(b.reward, np.stack([trajs[0].reward, trajs[1].reward], 1))


@pytest.mark.gpu
def test_ring_to_device_feeds_the_trainer():
    import srl_amd
    from srl_amd.api import config, trainer as trainer_api
    srl_amd.register_all()
    pol = dict(obs_dim=4, action_dim=3, hidden_dim=32, num_dense_layers=1, num_rnn_layers=0, popart=False, seed=4)
    tr_a = trainer_api.make(config.Trainer("mappo", args=dict(popart=False)), config.Policy("actor-critic", args=pol))
    tr_b = trainer_api.make(config.Trainer("mappo", args=dict(popart=False)), config.Policy("actor-critic", args=pol))
    ring = SampleRing(traj(0, T=16), batch_size=8, slots=2, device="cuda:0")
    for step in range(3):
        trajs = [traj(100 * step + s, T=16) for s in range(8)]
        for t in trajs:
            ring.put_column(t)
        dev_batch = ring.get_device()
        assert dev_batch.obs.obs.is_cuda and dev_batch.on_reset.dtype == torch.uint8
        ra = tr_a.step(dev_batch)
        ring.release(dev_batch)
        rb = tr_b.step(recursive_aggregate(trajs, lambda xs: np.stack(xs, axis=1)))
        for k in ("policy_loss", "value_loss", "entropy", "grad_norm"):
            assert abs(ra.stats[k] - rb.stats[k]) <= 1e-6 * max(1.0, abs(rb.stats[k])), (step, k)


def test_a_trajectory_of_the_wrong_shape_is_refused_before_a_column_is_claimed():
    """A write that failed after the claim would leave the column claimed and never committed: the slot would neither be
    published nor freed.  Shapes are therefore checked first, and the ring keeps every slot."""
    ring = SampleRing(traj(0), batch_size=2, slots=1)
    bad = traj(1, T=7)  # one row too many
    with pytest.raises(ValueError):
        ring.put_column(bad)
    wire = na.dumps(bad, method="raw_bytes")
    with pytest.raises(ValueError):
        ring.put_wire(wire)
    assert ring.put_column(traj(2)) is None and ring.put_column(traj(3)) == 0  # both columns still there
    ring.release(ring.get())
    assert ring.put_column(traj(4)) is None
