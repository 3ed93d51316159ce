"""Environment plugin interface (mirror of reference ``api/environment.py:15-203``).

Same class names, method names, registry behaviour (lazy module import for string registrations,
``KeyError`` on duplicate names) so that an SRL environment plugin registers here unchanged.
"""
import dataclasses
import importlib
from typing import Dict, List, Type, Union

import numpy as np

from srl_amd.api import config


class Action:
    pass


class ActionSpace:

    def sample(self, *args, **kwargs) -> Action:
        raise NotImplementedError()


class DataAugmenter:
    """Pre-processes a sample before it is sent to trainers (in place)."""

    def process(self, sample):
        raise NotImplementedError()


class NullAugmenter(DataAugmenter):

    def process(self, sample):
        return sample


@dataclasses.dataclass
class StepResult:
    """Step result of one agent; ``env.step`` returns one per agent (reference :45-54)."""
    obs: Dict
    reward: np.ndarray
    done: np.ndarray
    info: Dict
    truncated: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(shape=(1,), dtype=np.uint8))


class Environment:

    @property
    def agent_count(self) -> int:
        raise NotImplementedError()

    @property
    def observation_spaces(self) -> List[dict]:
        raise NotImplementedError()

    @property
    def action_spaces(self) -> List[ActionSpace]:
        raise NotImplementedError()

    def reset(self) -> List[StepResult]:
        raise NotImplementedError()

    def step(self, actions: List[Action]) -> List[StepResult]:
        raise NotImplementedError()

    def render(self) -> None:
        pass

    def seed(self, seed):
        raise NotImplementedError()

    def set_curriculum_stage(self, stage_name: str):
        raise NotImplementedError()


ALL_ENVIRONMENT_CLASSES = {}
ALL_ENVIRONMENT_MODULES = {}
ALL_AUGMENTER_CLASSES = {}


def register(name, env_class: Union[Type, str], module=None):
    """Register an environment class, or its name + module path for import-on-first-make."""
    if name in ALL_ENVIRONMENT_CLASSES:
        raise KeyError(f"Environment {name} already registered as {ALL_ENVIRONMENT_CLASSES[name]}. "
                       f"But got another register with env_class={env_class} and module={module}")
    if isinstance(env_class, str):
        assert module is not None, "For safe registration, specify module in api.environment.register."
        ALL_ENVIRONMENT_MODULES[name] = module
    ALL_ENVIRONMENT_CLASSES[name] = env_class


def register_relabler(name, relabeler_class):
    ALL_AUGMENTER_CLASSES[name] = relabeler_class


def make(cfg: Union[str, config.Environment]) -> Environment:
    if isinstance(cfg, str):
        cfg = config.Environment(type_=cfg)
    entry = ALL_ENVIRONMENT_CLASSES[cfg.type_]
    if isinstance(entry, str):
        if cfg.type_ not in ALL_ENVIRONMENT_MODULES:
            raise RuntimeError("Module is not registered correctly for safe registration.")
        entry = getattr(importlib.import_module(ALL_ENVIRONMENT_MODULES[cfg.type_]), entry)
        ALL_ENVIRONMENT_CLASSES[cfg.type_] = entry
    return entry(**cfg.args)


register_relabler("NULL", NullAugmenter)


def make_augmenter(cfg: Union[str, config.DataAugmenter]) -> DataAugmenter:
    if isinstance(cfg, str):
        cfg = config.DataAugmenter(type_=cfg)
    return ALL_AUGMENTER_CLASSES[cfg.type_](**cfg.args)
