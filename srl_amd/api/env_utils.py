"""Action containers and action spaces (mirror of reference ``api/env_utils.py``).

``DiscreteAction`` / ``ContinuousAction`` are namedarrays with the single field ``x`` (reference
``:10-25,83-95``). The reference's spaces wrap ``gym.spaces`` objects; ``gym`` is not a dependency
here, so the spaces take plain sizes (``n`` or ``nvec``) and own a numpy ``Generator``.
"""
from typing import Optional, Sequence, Union

import numpy as np

from srl_amd.api import environment
from srl_amd.namedarray import NamedArray


class DiscreteAction(NamedArray, environment.Action):

    def __init__(self, x: np.ndarray):
        super().__init__(x=x)

    def __eq__(self, other):
        assert isinstance(other, DiscreteAction), \
            "Cannot compare DiscreteAction to object of class{}".format(other.__class__.__name__)
        return self.key == other.key

    def __hash__(self):
        return hash(self.x.item())

    @property
    def key(self):
        return self.x.item()


class DiscreteActionSpace(environment.ActionSpace):
    """Discrete (``n`` int) or multi-discrete (``n`` sequence) action space."""

    def __init__(self, n: Union[int, Sequence[int]], shared=False, n_agents=-1, seed: Optional[int] = None):
        if shared and n_agents == -1:
            raise ValueError("n_agents must be given to a shared action space.")
        self.__multi = not isinstance(n, (int, np.integer))
        self.__n = np.asarray(n, dtype=np.int64) if self.__multi else int(n)
        self.__shared = shared
        self.__n_agents = n_agents
        self.__rng = np.random.default_rng(seed)

    @property
    def n(self):
        return self.__n

    def __draw(self):
        if self.__multi:
            return (self.__rng.random(len(self.__n)) * self.__n).astype(np.int32)
        return np.int32(self.__rng.integers(self.__n))

    def sample(self, available_action: np.ndarray = None) -> DiscreteAction:
        if available_action is None:
            if self.__shared:
                rows = [self.__draw() for _ in range(self.__n_agents)]
                x = np.array(rows if self.__multi else [[r] for r in rows], dtype=np.int32)
            else:
                x = np.array(self.__draw() if self.__multi else [self.__draw()], dtype=np.int32)
            return DiscreteAction(x)
        assert not self.__multi, "available_action masks are defined for single-discrete spaces"

        def draw_legal(mask):
            a = self.__draw()
            while not mask[a]:
                a = self.__draw()
            return a

        if self.__shared:
            assert available_action.shape == (self.__n_agents, self.__n)
            x = np.array([[draw_legal(available_action[i])] for i in range(self.__n_agents)], dtype=np.int32)
        else:
            assert available_action.shape == (self.__n,)
            x = np.array([draw_legal(available_action)], dtype=np.int32)
        return DiscreteAction(x)


class ContinuousAction(NamedArray, environment.Action):

    def __init__(self, x: np.ndarray):
        super().__init__(x=x)

    def __eq__(self, other):
        assert isinstance(other, ContinuousAction), \
            "Cannot compare ContinuousAction to object of class{}".format(other.__class__.__name__)
        return self.key == other.key

    @property
    def key(self):
        return self.x


class ContinuousActionSpace(environment.ActionSpace):
    """Box action space of dimension ``n`` with bounds ``[low, high]``."""

    def __init__(self, n: int, low=-1.0, high=1.0, shared=False, n_agents=-1, seed: Optional[int] = None):
        if shared and n_agents == -1:
            raise ValueError("n_agents must be given to a shared action space.")
        self.__n, self.__low, self.__high = int(n), low, high
        self.__shared, self.__n_agents = shared, n_agents
        self.__rng = np.random.default_rng(seed)

    @property
    def n(self):
        return self.__n

    def sample(self) -> ContinuousAction:
        shape = (self.__n_agents, self.__n) if self.__shared else (self.__n,)
        return ContinuousAction(self.__rng.uniform(self.__low, self.__high, size=shape).astype(np.float32))
