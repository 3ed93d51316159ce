"""Filesystem parameter store: the reference client's semantics, and the listing both clients
produced on a shared directory when the fixture was generated (tests/golden/gen_golden.py gen_paramdb)."""
import os

import numpy as np
import pytest
import torch

from srl_amd.runtime import synthetic
from srl_amd.runtime.parameter_db import FilesystemParameterDB, sample_staleness


def ck(steps):
    return {"steps": steps, "state_dict": {"w": torch.full((3,), float(steps))}}


def test_push_tag_get_gc_match_the_reference_listing(tmp_path, golden):
    g = golden("paramdb.npz")
    db = FilesystemParameterDB("exp", "trial", root=str(tmp_path), user_namespace="ns")
    db.push("pol", ck(5), version="5")
    db.push("pol", ck(12), version="12", tags="best")
    db.push("pol", ck(20), version="20", tags=["eval"])
    db.tag("pol", "5", "first")
    for v in (30, 40, 50):
        db.push("pol", ck(v), version=str(v))
    assert db.get("pol")["steps"] == 50 and db.get("pol", "best")["steps"] == 12 and db.version_of("pol", "first") == 5
    assert db.get("pol", "eval", mode="bytes")[:2] == b"PK"  # torch.save zip container, as the reference writes
    assert db.has_tag("pol", "latest") and not db.has_tag("pol", "nope")
    assert db.gc("pol", max_untagged_version_count=1) == 1  # 30 goes, 40 stays; tagged ones are kept
    assert list(g["versions_after_gc"]) == db.list_versions("pol")
    assert list(g["tags"]) == sorted(f"{t}={v}" for t, v in db.list_tags("pol"))
    assert list(g["names"]) == db.list_names()
    # a tag is a relative symlink next to the version files (what a stock SRL worker resolves)
    link = os.path.join(str(tmp_path), "ns", "exp", "trial", "pol", "latest")
    assert os.path.islink(link) and os.readlink(link) == "50"
    with pytest.raises(FileNotFoundError):
        db.get("pol", "31")
    with pytest.raises(FileNotFoundError):
        db.tag("pol", "31", "x")
    db.clear("pol")
    assert db.list_versions("pol") == []


def test_get_raises_when_the_retry_budget_is_spent(tmp_path):
    """parameter_db.py:193-217 of the reference: a checkpoint that never appears is a FileNotFoundError, blocking or not --
    never a silent None that a caller would load as a checkpoint."""
    db = FilesystemParameterDB("exp", "trial", root=str(tmp_path), user_namespace="ns")
    with pytest.raises(FileNotFoundError):
        db.get("p", "latest", block=True, retry_times=0)
    with pytest.raises(FileNotFoundError):
        db.get("p", "latest", block=False)
    with pytest.raises(FileNotFoundError):
        db.version_of("p", "latest")


def test_sample_admission_rule():
    arr = synthetic.make_sample_arrays(seed=0, T=4, B=3, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2)
    s = synthetic.to_sample_batch(arr)
    s.policy_version_steps[:] = 7
    s.policy_version_steps[0, 0] = 4
    out = sample_staleness(s, policy_version=9, preemption_steps=10)
    assert out["sample_min_policy_version"] == 4 and out["sample_version_difference"] == 5
    assert abs(out["staleness"] - (9 - s.policy_version_steps.mean())) < 1e-9
    assert sample_staleness(s, policy_version=9, preemption_steps=4) is None  # too old: dropped
    s.policy_version_steps[:] = -1
    assert sample_staleness(s, policy_version=9) is None  # no valid version on any row


@pytest.mark.gpu
def test_trainer_checkpoint_through_the_store(tmp_path):
    """A pushed trainer checkpoint (reference keys, incl. PopArt statistics and Adam moments) restores bit for bit."""
    import srl_amd
    from srl_amd.api import config, trainer as trainer_api
    srl_amd.register_all()
    pol = dict(obs_dim=4, action_dim=2, hidden_dim=32, num_dense_layers=1, num_rnn_layers=1, popart=True, chunk_len=4,
               shared_backbone=True, seed=2)
    mk = lambda: trainer_api.make(config.Trainer("mappo", args=dict(popart=True)), config.Policy("actor-critic", args=pol))
    sample = lambda s: synthetic.to_sample_batch(synthetic.make_sample_arrays(
        seed=s, T=8, B=4, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, policy_state={"hx": (1, 32)}))
    a = mk()
    a.step(sample(0))
    db = FilesystemParameterDB("e", "t", root=str(tmp_path))
    ckpt = a.get_checkpoint()
    db.push("default", ckpt, version=str(ckpt["steps"]))
    b = mk()
    b.load_checkpoint(db.get("default", "latest"))
    ra, rb = a.step(sample(1)), b.step(sample(1))
    assert ra.step == rb.step
    for k in ra.stats:  # `frames` counts what this trainer object has seen; float64 atomics order the last bits
        if k != "frames":
            assert abs(ra.stats[k] - rb.stats[k]) <= 1e-9 * max(1.0, abs(ra.stats[k])), k
    sa, sb = a.get_checkpoint()["state_dict"], b.get_checkpoint()["state_dict"]
    assert all(torch.allclose(sa[k], sb[k], rtol=0, atol=1e-7) for k in sa)
