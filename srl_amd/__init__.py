"""srl_amd: an MI355X-native rollout -> GAE -> PPO hot path behind SRL's plugin API.

Layout (only what the hot path needs):

* ``srl_amd.namedarray``     the ``[T, B, ...]`` SoA sample container (reference ``base/namedarray.py``)
* ``srl_amd.api``            ``environment`` / ``policy`` / ``trainer`` / ``config`` plugin surface
* ``srl_amd.hip``            ctypes binding of the C-ABI library ``csrc/libsrlhip.so`` (``include/srl_hip.h``)
* ``srl_amd.algorithm``      the ``mappo`` trainer and ``actor-critic*`` policies on HIP kernels
* ``srl_amd.runtime``        inference batcher, sample buffer, parameter broadcast
* ``srl_amd.envs``           host-side vectorised environments (CartPole, synthetic Atari-shaped)

Importing the package registers the plugins under the reference's names.
"""
__version__ = "0.1.0"


def register_all():
    """Import every plugin module so its ``register(...)`` calls run (idempotent)."""
    from srl_amd import algorithm  # noqa: F401
    from srl_amd import envs  # noqa: F401
