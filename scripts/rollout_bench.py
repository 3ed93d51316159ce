"""Secondary metric of SURVEY.md 8(d): rollout-inference requests/s (a10 `ActorCriticPolicy.rollout`, a11 batcher).

One request = one observation (Atari: 4x84x84 uint8) in, (action, log-prob, value) out, host numpy on both sides
as the policy worker sees them; the observation batch is copied H2D as uint8 inside the timed region.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srl_amd
from srl_amd.api import config, policy as policy_api
from srl_amd.namedarray import NamedArray
from srl_amd.runtime.batcher import InferenceBatcher

srl_amd.register_all()
POLICY = dict(obs_dim={"obs": (4, 84, 84)}, action_dim=6, hidden_dim=512, num_dense_layers=0, num_rnn_layers=0,
              popart=False, layernorm=False, shared_backbone=True, seed=1,
              cnn_layers=dict(obs=[(32, 8, 4, 0, 'zeros'), (64, 4, 2, 0, 'zeros'), (64, 3, 1, 0, 'zeros')]))


def request(n, rng, pinned):
    obs = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
    if pinned:
        obs = torch.from_numpy(obs).pin_memory().numpy()
    return policy_api.RolloutRequest(obs=NamedArray(obs=obs), is_evaluation=np.zeros((n, 1), np.uint8),
                                     on_reset=np.zeros((n, 1), np.uint8), client_id=np.zeros((n, 1), np.int32),
                                     request_id=np.arange(n).reshape(n, 1), received_time=np.zeros((n, 1), np.int64),
                                     buffer_index=np.zeros((n, 1), np.int32))


def main():
    pol = policy_api.make(config.Policy("actor-critic", args=POLICY))
    rng = np.random.default_rng(0)
    print(f"{'requests':>9} {'host obs':>9} {'ms/call':>9} {'requests/s':>12}")
    for n in (512, 4096, 10240):
        for pinned in (False, True):
            req = request(n, rng, pinned)
            for _ in range(3):
                pol.rollout(req)
            torch.cuda.synchronize()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                pol.rollout(req)  # returns numpy: synchronises
            dt = (time.perf_counter() - t0) / reps
            print(f"{n:9d} {'pinned' if pinned else 'pageable':>9} {dt * 1e3:9.2f} {n / dt:12.0f}", flush=True)
    # the batcher in front: 8 actors' worth of 512-row requests folded into batches of <= 4096
    batcher = InferenceBatcher(pol, batch_size=4096)
    reqs = [request(512, rng, False) for _ in range(8)]
    for r in reqs:
        batcher.post(r)
    batcher.poll()
    t0 = time.perf_counter()
    rounds = 5
    for _ in range(rounds):
        for r in reqs:
            batcher.post(r)
        out = batcher.poll()
    dt = (time.perf_counter() - t0) / rounds
    rows = sum(o.action.x.shape[0] for o in out)
    print(f"batcher: 8 x 512-row requests -> {len(out)} batches, {dt * 1e3:.2f} ms/round, {rows / dt:.0f} requests/s")


if __name__ == "__main__":
    main()
