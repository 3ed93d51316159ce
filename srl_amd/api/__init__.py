"""Plugin API mirroring the reference's ``api/`` package (environment / policy / trainer / config)."""
from srl_amd.api import config, environment, env_utils, policy, trainer  # noqa: F401
