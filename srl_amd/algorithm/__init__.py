"""Algorithm plugins of the hot path; importing this package registers them under the reference's names:
trainers ``mappo`` / ``mappo-hip``, policies ``actor-critic`` / ``actor-critic-separate`` / ``actor-critic-shared``,
trajectory post-processor ``gae``."""
from srl_amd.algorithm import actor_critic, gae, mappo  # noqa: F401
