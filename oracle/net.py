"""Oracle: the actor-critic network, PPO analysis and rollout on PyTorch-CPU float32.

Functional restatement (explicit parameter dict keyed by the reference's ``state_dict`` names, so the
reference's checkpoints and the golden fixtures load directly) of:

* ``ActorCriticSeparate``          ``legacy/algorithm/ppo/actor_critic_policies/actor_critic_policy.py:28-143``
* ``make_models_for_obs``          ``legacy/algorithm/ppo/actor_critic_policies/utils.py:33-63``
* ``mlp``                          ``legacy/algorithm/modules/utils.py:154-161``
* ``RecurrentBackbone``            ``legacy/algorithm/modules/recurrent_backbone.py:7-66``
* ``Convolution`` (Conv2d, pad 0)  ``legacy/algorithm/modules/cnn.py:39-135``
* ``AutoResetRNN`` (GRU)           ``legacy/algorithm/modules/autoreset_rnn.py:42-66``
* ``_ppo_analyze`` / ``rollout``   ``actor_critic_policy.py:338-390`` / ``:458-528``
* ``_ppg_phase2_analyze``          ``actor_critic_policy.py:417-435`` (``analyze_aux``; the loss side is ``oracle/ppg.py``)

TEST INFRASTRUCTURE ONLY (see package doc).  Parameter *initialisation* is not restated here: the
oracle always runs on weights handed to it (from a fixture, or from the product's own initialiser).
"""
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F


def _act(name):
    return {"relu": torch.relu, "tanh": torch.tanh}[name]


class OracleActorCritic:
    """Holds ``params`` (OrderedDict name -> leaf tensor, reference state_dict order) and evaluates the net."""

    def __init__(self,
                 obs_dim: Union[int, Dict[str, Union[int, Tuple[int, ...]]]],
                 action_dim: Union[int, Sequence[int]],
                 hidden_dim: int = 128,
                 state_dim=None,
                 value_dim: int = 1,
                 chunk_len: int = 10,
                 num_dense_layers: int = 2,
                 rnn_type: str = "gru",
                 cnn_layers: Optional[Dict[str, List[Tuple]]] = None,
                 use_maxpool: Optional[Dict[str, bool]] = None,
                 num_rnn_layers: int = 0,
                 popart: bool = False,
                 activation: str = "relu",
                 layernorm: bool = True,
                 shared_backbone: bool = False,
                 continuous_action: bool = False,
                 **_ignored):
        self.popart = popart
        # float32 like the reference; tests pass dtype=torch.float64 to get a higher-precision restatement of the SAME
        # arithmetic, against which both the float32 oracle's and the device path's rounding can be measured
        self.dtype = _ignored.get("dtype", torch.float32)
        self.continuous = continuous_action
        self.std_type = _ignored.get("std_type", "fixed")
        self.obs_dim = {"obs": obs_dim} if isinstance(obs_dim, int) else dict(obs_dim)
        if state_dim is not None and isinstance(state_dim, int):
            state_dim = {"state": state_dim}
        self.state_dim = state_dim or self.obs_dim
        self.act_dims = [action_dim] if isinstance(action_dim, int) else list(action_dim)
        self.hidden, self.value_dim, self.chunk_len = hidden_dim, value_dim, chunk_len
        self.dense_layers, self.layernorm, self.activation = num_dense_layers, layernorm, activation
        self.shared = shared_backbone
        self.cnn_layers = cnn_layers or {}
        self.use_maxpool = use_maxpool or {}
        self.num_rnn_layers, self.rnn_type = num_rnn_layers, rnn_type
        assert rnn_type in ("gru", "lstm") or num_rnn_layers == 0, "oracle restates the GRU and LSTM variants"
        self.params: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    # ------------------------------------------------------------------ parameters
    def load_state_dict(self, sd):
        # the running statistics of the PopArt head are float64 nn.Parameters without gradient (modules/utils.py:80-82)
        self.params = OrderedDict()
        for k, v in sd.items():
            t = torch.as_tensor(np.asarray(v)).clone()
            if "_RunningMeanStd__" in k:
                self.params[k] = t.double()
            else:
                fixed = k == "log_std" and self.continuous and self.std_type == "fixed"  # :88-89 requires_grad=False
                self.params[k] = t.to(self.dtype).requires_grad_(not fixed)

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.params.items())

    def parameters(self):
        return list(self.params.values())

    def _p(self, name):
        return self.params[name]

    # ------------------------------------------------------------------ building blocks
    def _embed(self, prefix, dims, obs):
        """LayerNorm(obs) -> Linear/Conv stack -> ... per key, concatenated (policies/utils.py:44-62)."""
        act = _act(self.activation)
        outs = []
        for k, shape in dims.items():
            x = obs[k]
            base = f"{prefix}.{k}"
            nshape = (shape,) if isinstance(shape, int) else tuple(shape)
            x = F.layer_norm(x, nshape, self._p(f"{base}.0.weight"), self._p(f"{base}.0.bias"), 1e-5)
            if isinstance(shape, int):
                x = act(F.linear(x, self._p(f"{base}.1.0.weight"), self._p(f"{base}.1.0.bias")))
                x = F.layer_norm(x, (self.hidden,), self._p(f"{base}.1.2.weight"), self._p(f"{base}.1.2.bias"), 1e-5)
            else:
                assert len(shape) in (2, 3, 4), "Conv1d / Conv2d / Conv3d by the observation's rank (cnn.py:60-71)"
                conv = {2: F.conv1d, 3: F.conv2d, 4: F.conv3d}[len(shape)]
                max_pool = {2: F.max_pool1d, 3: F.max_pool2d, 4: F.max_pool3d}[len(shape)]
                T, B = x.shape[:2]
                x = x.flatten(0, 1)  # cnn.py:131
                cb = f"{base}.1._Convolution__model"
                layers = self.cnn_layers[k]
                pool = bool(self.use_maxpool.get(k, False))
                idx = 0  # index in the reference's nn.Sequential: [MaxPool2d(2)] Conv2d act ... Flatten mlp (cnn.py:99-126)
                for i, (_, _, stride, padding, pmode) in enumerate(layers):
                    if pool and i != len(layers) - 1:
                        x = max_pool(x, 2)
                        idx += 1
                    if padding and pmode != "zeros":  # torch/nn/modules/conv.py _conv_forward: F.pad with the mode, then no padding
                        x = F.pad(x, (padding, padding) * (len(shape) - 1), mode=pmode)
                        padding = 0
                    x = act(conv(x, self._p(f"{cb}.{idx}.weight"), self._p(f"{cb}.{idx}.bias"), stride=stride, padding=padding))
                    idx += 2
                x = x.flatten(1)
                j = 0
                fb = f"{cb}.{idx + 1}"
                while f"{fb}.{3 * j}.weight" in self.params:  # cnn.py:86-91: halve until <= 8*hidden
                    x = torch.relu(F.linear(x, self._p(f"{fb}.{3 * j}.weight"), self._p(f"{fb}.{3 * j}.bias")))
                    x = F.layer_norm(x, (x.shape[-1],), self._p(f"{fb}.{3 * j + 2}.weight"),
                                     self._p(f"{fb}.{3 * j + 2}.bias"), 1e-5)
                    j += 1
                x = x.reshape(T, B, -1)
            outs.append(x)
        return torch.cat(outs, dim=-1)

    def _rnn_key(self, prefix, n, layer):
        return f"{prefix}.rnn._AutoResetRNN__net.{n}_l{layer}"

    def _gru_step(self, prefix, layer, x, h):
        p = lambda n: self._p(self._rnn_key(prefix, n, layer))
        gi = F.linear(x, p("weight_ih"), p("bias_ih"))
        gh = F.linear(h, p("weight_hh"), p("bias_hh"))
        ir, iz, inn = gi.chunk(3, -1)
        hr, hz, hn = gh.chunk(3, -1)
        r = torch.sigmoid(ir + hr)
        z = torch.sigmoid(iz + hz)
        n = torch.tanh(inn + r * hn)
        return (1 - z) * n + z * h

    def _lstm_step(self, prefix, layer, x, hc):
        """torch.nn.LSTM cell; the AutoResetRNN state is cat(h, c) on the last axis (autoreset_rnn.py:31-39)."""
        p = lambda n: self._p(self._rnn_key(prefix, n, layer))
        H = self.hidden
        h, c = hc[..., :H], hc[..., H:]
        pre = F.linear(x, p("weight_ih"), p("bias_ih")) + F.linear(h, p("weight_hh"), p("bias_hh"))
        i, f, g, o = pre.chunk(4, -1)
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h2 = torch.sigmoid(o) * torch.tanh(c2)
        return torch.cat([h2, c2], -1)

    def _backbone(self, prefix, x, hx, on_reset):
        """dense layers (+GRU with per-step reset of the hidden state) (recurrent_backbone.py:61-66)."""
        act = _act(self.activation)
        stride = 3 if self.layernorm else 2
        for j in range(self.dense_layers):
            x = act(F.linear(x, self._p(f"{prefix}.fc.{stride * j}.weight"), self._p(f"{prefix}.fc.{stride * j}.bias")))
            if self.layernorm:
                x = F.layer_norm(x, (self.hidden,), self._p(f"{prefix}.fc.{stride * j + 2}.weight"),
                                 self._p(f"{prefix}.fc.{stride * j + 2}.bias"), 1e-5)
        if self.num_rnn_layers == 0:
            return x, hx
        y, h = self._recur(prefix, x, hx, on_reset)
        y = F.layer_norm(y, (self.hidden,), self._p(f"{prefix}.rnn_norm.weight"), self._p(f"{prefix}.rnn_norm.bias"), 1e-5)
        return y, h

    def _recur(self, prefix, x, hx, on_reset):
        # AutoResetRNN: h is zeroed at every step whose on_reset flag is set (autoreset_rnn.py:46-60);
        # stepping one row at a time is arithmetically the same as the reference's segment batching.
        h = [hx[l] for l in range(self.num_rnn_layers)]
        ys = []
        for t in range(x.shape[0]):
            inp = x[t]
            for l in range(self.num_rnn_layers):
                hl = h[l] if on_reset is None else h[l] * (1 - on_reset[t])
                if self.rnn_type == "lstm":
                    h[l] = self._lstm_step(prefix, l, inp, hl)
                    inp = h[l][..., :self.hidden]
                else:
                    h[l] = self._gru_step(prefix, l, inp, hl)
                    inp = h[l]
            ys.append(inp)
        return torch.stack(ys, 0), torch.stack(h, 0)

    # ------------------------------------------------------------------ forward
    def forward(self, obs: Dict[str, torch.Tensor], policy_state=(None, None), on_reset=None):
        """obs leaves [T, B, ...] float32 -> (logits [T,B,sum(A)], value [T,B,value_dim], new policy_state)."""
        feat = self._embed("obs_modules_dict", self.obs_dim, obs)
        a_feat, a_hx = self._backbone("actor_backbone", feat, policy_state[0], on_reset)
        if self.shared:
            c_feat, new_state = a_feat, (a_hx,)
        else:
            sfeat = self._embed("state_modules_dict", self.state_dim, obs)
            c_feat, c_hx = self._backbone("critic_backbone", sfeat, policy_state[1], on_reset)
            new_state = (a_hx, c_hx)
        logits = F.linear(a_feat, self._p("actor_head.weight"), self._p("actor_head.bias"))
        if self.continuous:  # actor_critic_policy.py:128-133: (mean, std); std from a vector or from a second head
            if self.std_type == "shared_learnable":
                self._std = F.linear(a_feat, self._p("log_std.weight"), self._p("log_std.bias")).exp()
            else:
                self._std = self._p("log_std").exp() * torch.ones_like(logits)
        elif "available_action" in obs:
            logits = logits.masked_fill(obs["available_action"] == 0, -1e10)  # actor_critic_policy.py:135-136
        head = "critic_head._PopArtValueHead__" if self.popart else "critic_head."  # popart.py:21-22,39-40
        value = F.linear(c_feat, self._p(head + "weight"), self._p(head + "bias"))
        # PPG's auxiliary value head reads the ACTOR's features (actor_critic_policy.py:105-107, 139-140)
        self._aux_value = (F.linear(a_feat, self._p("auxiliary_value_head.weight"), self._p("auxiliary_value_head.bias"))
                           if "auxiliary_value_head.weight" in self.params else None)
        return logits, value, new_state

    # ------------------------------------------------------------------ PopArt (popart.py:8-59, modules/utils.py:70-151)
    POPART_BETA, POPART_EPS = 0.99999, 1e-5  # PopArtValueHead defaults; burn_in_updates = inf: never rescales

    VALUE_HEAD = "critic_head"

    def _rms(self, n):
        return self.params[f"{self.VALUE_HEAD}._PopArtValueHead__rms._RunningMeanStd__{n}"]

    @torch.no_grad()
    def popart_mean_std(self):
        deb = self._rms("debiasing_term").clamp(min=self.POPART_EPS)
        mean = self._rms("mean") / deb
        var = (self._rms("mean_sq") / deb - mean**2).clamp(min=1e-2)
        return mean, var.sqrt()

    @torch.no_grad()
    def normalize_value(self, x):
        mean, std = self.popart_mean_std()
        return ((x.double() - mean) / std).clip(-5, 5).float()

    @torch.no_grad()
    def denormalize_value(self, x):
        mean, std = self.popart_mean_std()
        return (x.double() * std + mean).float()

    @torch.no_grad()
    def update_popart(self, x, mask):
        x, mask = x.double(), mask.double()
        x = x * mask
        factor = mask.sum()
        dims = tuple(range(x.dim() - 1))
        bm, bsq = x.sum(dims) / factor, x.square().sum(dims) / factor
        b = self.POPART_BETA
        self._rms("mean")[:] = b * self._rms("mean") + bm * (1.0 - b)
        self._rms("mean_sq")[:] = b * self._rms("mean_sq") + bsq * (1.0 - b)
        self._rms("debiasing_term")[:] = b * self._rms("debiasing_term") + 1.0 - b

    def _heads(self, logits):
        out, start = [], 0
        for d in self.act_dims:
            out.append(torch.distributions.Categorical(logits=logits[..., start:start + d]))
            start += d
        return out

    def analyze(self, obs, action, on_reset, policy_state=None, burn_in_steps=0):
        """PPO analysis (actor_critic_policy.py:338-390): new log-probs, state values, entropy, each [T,B,1].

        Without an RNN the chunking of the reference ([T,B] -> [C, B*T/C] and back) is a pure reshape
        of independent rows, so it is skipped; with a GRU the chunks start from the stored per-row
        ``policy_state`` exactly as the reference does.
        """
        burn = burn_in_steps
        if self.num_rnn_layers == 0:
            logits, value, _ = self.forward({k: v[burn:] for k, v in obs.items()})
        else:
            T = on_reset.shape[0] - burn
            C = self.chunk_len
            n = T // C
            chunk = lambda x: torch.cat(torch.split(x, T // n, dim=0), dim=1)  # modules/utils.py:164-182
            unchunk = lambda x: torch.cat(torch.split(x, x.shape[1] // n, dim=1), dim=0)
            cobs = {k: chunk(v[burn:]) for k, v in obs.items()}
            if burn == 0:
                state = tuple(chunk(s)[0].transpose(0, 1) for s in policy_state)  # :361-363
            else:
                # :365-378: the `burn` rows before every chunk are replayed without gradient from the state stored
                # at their first row; what comes out is the chunk's initial state
                win = lambda x: torch.cat([x[i * C:i * C + burn] for i in range(n)], dim=1)
                with torch.no_grad():
                    _, _, state = self.forward({k: win(v) for k, v in obs.items()},
                                               tuple(win(s)[0].transpose(0, 1) for s in policy_state), win(on_reset))
            logits, value, _ = self.forward(cobs, state, chunk(on_reset[burn:]))
            logits, value = unchunk(logits), unchunk(value)
        action = action[burn:]
        if self.continuous:  # :318-321
            std = unchunk(self._std) if (self.num_rnn_layers and self._std.shape != logits.shape) else self._std
            dist = torch.distributions.Normal(logits, std)
            return dist.log_prob(action).sum(-1, keepdim=True), value, dist.entropy().sum(-1, keepdim=True), logits
        dists = self._heads(logits)
        lp = torch.stack([d.log_prob(action[..., i]) for i, d in enumerate(dists)], -1).sum(-1, keepdim=True)
        ent = torch.stack([d.entropy() for d in dists], -1).sum(-1, keepdim=True)
        return lp, value, ent, logits

    def analyze_aux(self, obs, on_reset, policy_state=None):
        """PPG auxiliary-phase analysis (actor_critic_policy.py:417-435): per action head the NORMALISED log-probabilities
        (``Categorical(logits=...).logits``) [T,B,A_h], the auxiliary value and the critic head's value [T,B,value_dim]; recurrent
        nets are chunked from the stored states exactly as in ``analyze``."""
        if self.num_rnn_layers == 0:
            logits, value, _ = self.forward(obs)
            aux = self._aux_value
        else:
            T = on_reset.shape[0]
            n = T // self.chunk_len
            chunk = lambda x: torch.cat(torch.split(x, T // n, dim=0), dim=1)
            unchunk = lambda x: torch.cat(torch.split(x, x.shape[1] // n, dim=1), dim=0)
            state = tuple(chunk(s)[0].transpose(0, 1) for s in policy_state)
            logits, value, _ = self.forward({k: chunk(v) for k, v in obs.items()}, state, chunk(on_reset))
            logits, value, aux = unchunk(logits), unchunk(value), unchunk(self._aux_value)
        return [d.logits for d in self._heads(logits)], aux, value

    @torch.no_grad()
    def rollout_eval(self, obs, policy_state=(None, None)):
        """Deterministic (argmax) rollout (actor_critic_policy.py:480-497 with is_evaluation=1).

        obs leaves [N, ...]; returns actions [N, heads] int64, log_probs [N,1], value [N,value_dim], logits.
        """
        obs = {k: v.unsqueeze(0) for k, v in obs.items()}
        logits, value, new_state = self.forward(obs, policy_state)
        logits, value = logits.squeeze(0), value.squeeze(0)
        if self.continuous:  # :499-506 with is_evaluation = 1: the action is the mean
            dist = torch.distributions.Normal(logits, self._std.squeeze(0))
            return logits, dist.log_prob(logits).sum(-1, keepdim=True), value, logits, new_state
        dists = self._heads(logits)
        actions = torch.stack([d.probs.argmax(-1) for d in dists], -1)
        lp = torch.stack([d.log_prob(actions[..., i]) for i, d in enumerate(dists)], -1).sum(-1, keepdim=True)
        return actions, lp, value, logits, new_state


class OracleSMACNet(OracleActorCritic):
    """``SMACNet`` with flat observations (game_policies/smac_rnn.py:88-167) and ``SMACPolicy.analyze`` (:239-327).

    Leaves of a shared environment are ``[T, B, agents, ...]``; agents are merged into the batch axis (:253-266) and the
    outputs split again (:313-320).  The recurrent cell is the LSTM ``AutoResetRNN`` defaults to (autoreset_rnn.py:9):
    stored states are ``cat(h, c)``."""
    VALUE_HEAD = "value_head"

    def __init__(self, obs_dim, state_dim, act_dim, hidden_dim, chunk_len, num_rnn_layers=1, agent_shared=True, **_ignored):
        super().__init__(obs_dim={"local_obs": obs_dim}, action_dim=act_dim, hidden_dim=hidden_dim,
                         state_dim={"state": state_dim}, chunk_len=chunk_len, num_rnn_layers=num_rnn_layers,
                         rnn_type="lstm", popart=True, shared_backbone=False)
        self.agent_shared = agent_shared

    def _rnn_key(self, prefix, n, layer):
        return f"{prefix}._AutoResetRNN__net.{n}_l{layer}"

    def _base(self, root, x):
        p = self._p
        x = F.layer_norm(x, (x.shape[-1],), p(f"{root}.0.weight"), p(f"{root}.0.bias"), 1e-5)
        for j in (0, 3):  # mlp([d, H, H], ReLU, layernorm=True): Linear, ReLU, LayerNorm twice (modules/utils.py:154-161)
            x = torch.relu(F.linear(x, p(f"{root}.1.{j}.weight"), p(f"{root}.1.{j}.bias")))
            x = F.layer_norm(x, (self.hidden,), p(f"{root}.1.{j + 2}.weight"), p(f"{root}.1.{j + 2}.bias"), 1e-5)
        return x

    def forward(self, obs, policy_state=(None, None), on_reset=None):
        a, c = self._base("actor_base", obs["local_obs"]), self._base("critic_base", obs["state"])
        a_hx = c_hx = None
        if self.num_rnn_layers:
            a, a_hx = self._recur("actor_rnn", a, policy_state[0], on_reset)
            c, c_hx = self._recur("critic_rnn", c, policy_state[1], on_reset)
            a = F.layer_norm(a, (self.hidden,), self._p("actor_rnn_norm.weight"), self._p("actor_rnn_norm.bias"), 1e-5)
            c = F.layer_norm(c, (self.hidden,), self._p("critic_rnn_norm.weight"), self._p("critic_rnn_norm.bias"), 1e-5)
        logits = F.linear(a, self._p("policy_head.weight"), self._p("policy_head.bias"))
        logits = logits.masked_fill(obs["available_action"] == 0, -1e10)  # :165
        value = F.linear(c, self._p("value_head._PopArtValueHead__weight"), self._p("value_head._PopArtValueHead__bias"))
        return logits, value, (a_hx, c_hx)

    def analyze(self, obs, action, on_reset, policy_state=None, burn_in_steps=0):
        assert burn_in_steps == 0, "oracle: SMAC burn-in not restated"
        bs = on_reset.shape[1]
        merge = (lambda x: x.reshape(x.shape[0], x.shape[1] * x.shape[2], *x.shape[3:])) if self.agent_shared else (lambda x: x)
        split = (lambda x: x.reshape(x.shape[0], bs, x.shape[1] // bs, *x.shape[2:])) if self.agent_shared else (lambda x: x)
        T = on_reset.shape[0]
        n = T // self.chunk_len
        chunk = lambda x: torch.cat(torch.split(merge(x), T // n, dim=0), dim=1)
        unchunk = lambda x: torch.cat(torch.split(x, x.shape[1] // n, dim=1), dim=0)
        cobs = {k: chunk(v) for k, v in obs.items()}
        state = tuple(chunk(s)[0].transpose(0, 1) for s in policy_state)
        logits, value, _ = self.forward(cobs, state, chunk(on_reset))
        dist = torch.distributions.Categorical(logits=logits)
        lp = dist.log_prob(chunk(action).squeeze(-1)).unsqueeze(-1)
        if self.agent_shared:  # :309-311
            lp = lp.masked_fill(cobs["is_alive"] == 0, float("-inf"))
        ent = dist.entropy().unsqueeze(-1)
        return split(unchunk(lp)), split(unchunk(value)), split(unchunk(ent)), split(unchunk(logits))
