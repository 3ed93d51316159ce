"""Atari-shaped synthetic environment: uint8 ``(4, 84, 84)`` frame stacks, ``n_actions`` discrete actions.

There is no ALE on the benchmark machine; this stands in for ``legacy/environment/atari`` (frame-skip 4, 4
stacked 84x84 gray frames, ``legacy/experiments/atari.py:917-931``) with cheap deterministic pseudo-frames so
the rollout -> inference -> sample path can be exercised at the real tensor shapes.  Episode length is
geometric with mean ``mean_episode_len``; the reward is a fixed random function of (frame hash, action).
"""
import numpy as np

from srl_amd.api import environment as env_api
from srl_amd.api.env_utils import DiscreteActionSpace


class SyntheticAtariEnvironment(env_api.Environment):

    def __init__(self, n_actions: int = 6, mean_episode_len: int = 800, max_steps: int = 27000, seed: int = 0, **_):
        self._rng = np.random.default_rng(seed)
        self._n_actions = n_actions
        self._p_end = 1.0 / mean_episode_len
        self._max_steps = max_steps
        self._space = DiscreteActionSpace(n_actions, seed=seed)
        self._frames = None
        self._t = 0
        self._ret = 0.0

    @property
    def agent_count(self) -> int:
        return 1

    @property
    def observation_spaces(self):
        return [{"obs": (4, 84, 84)}]

    @property
    def action_spaces(self):
        return [self._space]

    def seed(self, seed):
        self._rng = np.random.default_rng(seed)
        return seed

    def _new_frame(self):
        return self._rng.integers(0, 256, size=(84, 84), dtype=np.uint8)

    def reset(self):
        self._frames = np.stack([self._new_frame() for _ in range(4)])
        self._t, self._ret = 0, 0.0
        return [env_api.StepResult(obs={"obs": self._frames.copy()}, reward=np.zeros(1, np.float32),
                                   done=np.zeros(1, np.uint8),
                                   info=dict(episode_length=np.zeros(1, np.float32),
                                             episode_return=np.zeros(1, np.float32)))]

    def step(self, actions):
        a = int(np.asarray(actions[0].x).reshape(-1)[0])
        self._frames = np.concatenate([self._frames[1:], self._new_frame()[None]], axis=0)
        self._t += 1
        r = float(np.sign(((int(self._frames[-1, 0, 0]) + 31 * a) % 7) - 3))  # clipped rewards in {-1, 0, 1}
        self._ret += r
        done = self._rng.random() < self._p_end
        timeout = (not done) and self._t >= self._max_steps
        info = dict(episode_length=np.array([self._t], np.float32), episode_return=np.array([self._ret], np.float32))
        return [env_api.StepResult(obs={"obs": self._frames.copy()}, reward=np.array([r], np.float32),
                                   done=np.array([done], np.uint8), info=info,
                                   truncated=np.array([timeout], np.uint8))]
