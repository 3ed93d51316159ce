"""Algorithm plugins of the hot path; importing this package registers them under the reference's names:
trainers ``mappo`` / ``mappo-hip`` / ``mappg``, policies ``actor-critic`` / ``actor-critic-separate`` / ``actor-critic-shared`` (+ the continuous-action names), the
per-game presets of ``game_policies`` (football, atari-vision, overcooked), the multi-agent ``smac_rnn``,
trajectory post-processor ``gae``."""
from srl_amd.algorithm import actor_critic, gae, game_policies, mappg, mappo, smac_policy  # noqa: F401
