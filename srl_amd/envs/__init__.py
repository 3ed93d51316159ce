"""Host-side environments shipped with the package (registered lazily by name)."""
from srl_amd.api import environment as _env

_env.register("cartpole", "CartPoleEnvironment", "srl_amd.envs.cartpole")
_env.register("synthetic-atari", "SyntheticAtariEnvironment", "srl_amd.envs.synthetic_atari")
