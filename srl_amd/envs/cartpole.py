"""CartPole-v1 as an ``api.environment.Environment`` (the reference ships no CartPole: SURVEY.md 0.6).

Classic cart-pole dynamics (Barto, Sutton & Anderson 1983; Euler integration, tau = 0.02 s, force 10 N,
termination at |x| > 2.4 or |theta| > 12 deg, time limit 500 steps -> ``truncated``), single agent,
observation key ``obs`` float32[4], discrete action {0, 1}.  Follows the step contract of the reference's
environments (``api/environment.py:45-152``): ``reward`` float32[1], ``done`` / ``truncated`` uint8[1].
"""
import math

import numpy as np

from srl_amd.api import environment as env_api
from srl_amd.api.env_utils import DiscreteActionSpace


class CartPoleEnvironment(env_api.Environment):
    GRAVITY, M_CART, M_POLE, HALF_LEN, FORCE, TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    X_LIMIT, THETA_LIMIT = 2.4, 12 * 2 * math.pi / 360

    def __init__(self, max_steps: int = 500, seed: int = 0, **_):
        self._rng = np.random.default_rng(seed)
        self._max_steps = max_steps
        self._space = DiscreteActionSpace(2, seed=seed)
        self._state = None
        self._t = 0
        self._ret = 0.0

    @property
    def agent_count(self) -> int:
        return 1

    @property
    def observation_spaces(self):
        return [{"obs": (4,)}]

    @property
    def action_spaces(self):
        return [self._space]

    def seed(self, seed):
        self._rng = np.random.default_rng(seed)
        return seed

    def _obs(self):
        return {"obs": self._state.astype(np.float32)}

    def reset(self):
        self._state = self._rng.uniform(-0.05, 0.05, size=4)
        self._t, self._ret = 0, 0.0
        return [env_api.StepResult(obs=self._obs(), reward=np.zeros(1, np.float32), done=np.zeros(1, np.uint8),
                                   info=dict(episode_length=np.zeros(1, np.float32),
                                             episode_return=np.zeros(1, np.float32)))]

    def step(self, actions):
        a = int(np.asarray(actions[0].x).reshape(-1)[0])
        x, x_dot, th, th_dot = self._state
        force = self.FORCE if a == 1 else -self.FORCE
        total_m = self.M_CART + self.M_POLE
        pml = self.M_POLE * self.HALF_LEN
        cos, sin = math.cos(th), math.sin(th)
        tmp = (force + pml * th_dot**2 * sin) / total_m
        th_acc = (self.GRAVITY * sin - cos * tmp) / (self.HALF_LEN * (4.0 / 3.0 - self.M_POLE * cos**2 / total_m))
        x_acc = tmp - pml * th_acc * cos / total_m
        self._state = np.array([x + self.TAU * x_dot, x_dot + self.TAU * x_acc, th + self.TAU * th_dot,
                                th_dot + self.TAU * th_acc])
        self._t += 1
        self._ret += 1.0
        failed = abs(self._state[0]) > self.X_LIMIT or abs(self._state[2]) > self.THETA_LIMIT
        timeout = (not failed) and self._t >= self._max_steps
        info = dict(episode_length=np.array([self._t], np.float32), episode_return=np.array([self._ret], np.float32))
        return [env_api.StepResult(obs=self._obs(), reward=np.ones(1, np.float32), done=np.array([failed], np.uint8),
                                   info=info, truncated=np.array([timeout], np.uint8))]
