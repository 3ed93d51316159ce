"""The reference's per-game registrations that are thin presets of ``ActorCriticPolicy``.

Each class only fills in default keywords and forwards to the generic actor-critic, exactly as the reference files do
(``legacy/algorithm/ppo/game_policies/football_rnn.py:8-58``, ``atari_naive_rnn.py:7-35``, ``overcooked_rnn.py:7-29``);
registered under the same names, so an experiment config that names them resolves here.  ``seed`` defaults to a random
draw in the reference; here it defaults to 0 (pass one for anything that must be reproducible either way).
"""
from srl_amd.algorithm.actor_critic import ActorCriticPolicy
from srl_amd.api.policy import register


def _preset(name, **defaults):

    class Preset(ActorCriticPolicy):

        def __init__(self, **kwargs):
            args = dict(defaults)
            args.update({k: v for k, v in kwargs.items()})
            super().__init__(**args)

    Preset.__name__ = Preset.__qualname__ = name
    Preset.defaults = dict(defaults)
    return Preset


# football_rnn.py:8-31 -- simple115 vector observation, separate backbones by default
FootballSeparatePolicy = _preset("FootballSeparatePolicy", obs_dim=115, action_dim=19, hidden_dim=128, rnn_type="gru",
                                 num_rnn_layers=1, chunk_len=10, popart=True, shared_backbone=False, auxiliary_head=False,
                                 seed=0)
# football_rnn.py:34-55 -- stacked super-mini-map frames through the default convolution stack (cnn.py:96-98)
FootballSMMPolicy = _preset("FootballSMMPolicy", obs_dim={"obs": (4, 96, 72)}, action_dim=19, hidden_dim=128,
                            rnn_type="gru", num_rnn_layers=1, chunk_len=10, popart=True, auxiliary_head=False, seed=0)
# atari_naive_rnn.py:7-31 -- raw RGB frames, two 5x5 stride-2 convolutions, one dense layer, no recurrence by default
AtariVisionPolicy = _preset("AtariVisionPolicy", obs_dim={"obs": (3, 160, 210)}, action_dim=18, hidden_dim=32, rnn_type="gru",
                            num_rnn_layers=0, chunk_len=10, num_dense_layers=1,
                            cnn_layers={"obs": [(3, 5, 2, 0, "zeros"), (3, 5, 2, 0, "zeros")]}, popart=True,
                            auxiliary_head=False, seed=0)
# overcooked_rnn.py:7-26
OvercookedSeparatePolicy = _preset("OvercookedSeparatePolicy", obs_dim=96, action_dim=6, hidden_dim=128, rnn_type="gru",
                                   num_rnn_layers=1, chunk_len=10, popart=True, seed=0)

register("football-simple115-separate", FootballSeparatePolicy)
register("football-smm-separate", FootballSMMPolicy)
register("atari-vision", AtariVisionPolicy)
register("atari_naive_rnn", AtariVisionPolicy)
register("overcooked-separate", OvercookedSeparatePolicy)
