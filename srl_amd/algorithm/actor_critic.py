"""Actor-critic policy on the HIP kernels, registered under the reference's names.

Mirror of ``ActorCriticPolicy`` (reference ``legacy/algorithm/ppo/actor_critic_policies/actor_critic_policy.py``):
same constructor keywords (``:146-166``), ``rollout`` contract (``:458-528``: numpy ``RolloutRequest`` in, numpy
``RolloutResult`` out, ``action.x`` int64 ``[N, heads]``, ``log_probs`` / ``value`` ``[N, 1]``), ``analyze(target="ppo")``
contract (``:338-390``), version / checkpoint semantics of ``SingleModelPytorchPolicy`` (``api/policy.py:205-288``:
``{"steps", "state_dict"}`` with the reference's parameter names and shapes on the CPU).

Differences by design: observations stay in their wire dtype on the device (uint8 frames are normalised
inside the first convolution's gather instead of being widened to float32 first, ``:467-469``); sampling
uses a counter-based Philox stream keyed by ``(seed, call counter)`` instead of the global torch RNG, so a
rollout is reproducible regardless of how requests are batched.
"""
import os
import functools
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.distributed as dist

from srl_amd import hip
from srl_amd.algorithm import netspec as ns
from srl_amd.algorithm.hipnet import HipNet, RnnCtx
from srl_amd.algorithm.ppo_types import (CategoricalLogits, PPGPhase1AnalyzedResult, PPGPhase2AnalyzedResult, PPORolloutAnalyzedResult,
                                         SampleAnalyzedResult)
from srl_amd.api import policy as policy_api
from srl_amd.api.env_utils import DiscreteAction
from srl_amd.namedarray import NamedArray
from srl_amd.runtime.obs_ring import ObsRing, RingObs


def to_device_leaf(x, device, kind: str) -> torch.Tensor:
    """Host leaf -> contiguous device tensor in the dtype the kernels read.

    kind: "flag" (uint8), "real" (float32), "index" (int32), "obs" (uint8 stays uint8, everything else float32).
    Device tensors pass through (only the dtype is checked).
    """
    want = {"flag": torch.uint8, "real": torch.float32, "index": torch.int32}.get(kind)
    if isinstance(x, RingObs):  # rows that never left HBM since their rollout (runtime/obs_ring.py)
        return x
    if kind == "as-is":  # already in its wire dtype (wire_leaf): only the placement changes
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(x)
        return t.to(device, non_blocking=True).contiguous()
    if isinstance(x, torch.Tensor):
        t = x
    else:
        a = np.asarray(x)
        if a.dtype == np.bool_:
            a = a.view(np.uint8)
        a = np.ascontiguousarray(a)
        if not a.flags.writeable:  # broadcast views: torch wants to own writable memory
            a = a.copy()
        t = torch.from_numpy(a)
    if kind == "obs":
        want = torch.uint8 if t.dtype == torch.uint8 else torch.float32
    if t.dtype != want:
        t = t.to(want)
    return t.to(device, non_blocking=True).contiguous()


def wire_leaf(x, kind: str):
    """Like ``to_device_leaf`` but a host leaf STAYS on the host: a contiguous numpy array in the dtype the kernels read
    (what the native step driver copies straight into a captured step's static inputs).  Device tensors pass through."""
    if isinstance(x, RingObs):
        return x
    if isinstance(x, torch.Tensor):
        return to_device_leaf(x, x.device, kind) if x.is_cuda else wire_leaf(x.numpy(), kind)
    a = np.asarray(x)
    if a.dtype == np.bool_:
        a = a.view(np.uint8)
    want = {"flag": np.uint8, "real": np.float32, "index": np.int32}.get(kind)
    if kind == "obs":
        want = np.uint8 if a.dtype == np.uint8 else np.float32
    return np.ascontiguousarray(a, dtype=want)


class PendingRollout:
    """An issued ``rollout`` (``ActorCriticPolicy.rollout_async``): ``result()`` waits for its stream position and returns the
    ``RolloutResult`` (numpy leaves, copied out of the policy's pinned blocks, which the next call reuses)."""

    def __init__(self, policy, views, done, refs, state, n):
        self._policy, self._views, self._done, self._refs, self._state, self._n = policy, views, done, refs, state, n

    def result(self) -> policy_api.RolloutResult:
        if self._policy is None:
            raise RuntimeError("PendingRollout.result() was already taken")
        pol, self._policy = self._policy, None
        try:
            self._done.synchronize()
            h_action, h_logp, h_value = [v.numpy().copy() for v in self._views]
        finally:
            pol._net._serving[0] -= 1
            pol._pending_rollout = False
        refs = self._refs
        analyzed = PPORolloutAnalyzedResult(log_probs=h_logp, value=h_value, obs_ref=None if refs is None else refs.reshape(self._n, 1))
        return policy_api.RolloutResult(action=DiscreteAction(h_action), analyzed_result=analyzed, policy_state=self._state)


class ActorCriticPolicy(policy_api.Policy):

    def __init__(self,
                 obs_dim: Union[int, Dict[str, Union[int, Tuple[int]]]],
                 action_dim: Union[int, List[int]],
                 hidden_dim: int = 128,
                 state_dim=None,
                 value_dim: int = 1,
                 chunk_len: int = 10,
                 num_dense_layers: int = 2,
                 rnn_type: str = "gru",
                 cnn_layers: Dict[str, Tuple] = None,
                 use_maxpool: Dict[str, Tuple] = None,
                 num_rnn_layers: int = 1,
                 popart: bool = True,
                 activation: str = "relu",
                 layernorm: bool = True,
                 shared_backbone: bool = False,
                 continuous_action: bool = False,
                 auxiliary_head: bool = False,
                 seed=0,
                 **kwargs):
        super().__init__()
        if auxiliary_head and shared_backbone:
            raise AttributeError("Cannot use shared backbone when requiring auxiliary value head.")
        self.spec, init = ns.build_netspec(obs_dim=obs_dim, action_dim=action_dim, hidden_dim=hidden_dim,
                                           state_dim=state_dim, value_dim=value_dim, num_dense_layers=num_dense_layers,
                                           cnn_layers=cnn_layers, use_maxpool=use_maxpool,
                                           num_rnn_layers=num_rnn_layers, rnn_type=rnn_type, popart=popart,
                                           activation=activation,
                                           layernorm=layernorm, shared_backbone=shared_backbone,
                                           continuous_action=continuous_action, auxiliary_head=auxiliary_head,
                                           std_type=kwargs.get("std_type", "fixed"),
                                           init_log_std=kwargs.get("init_log_std", -0.5),
                                           seed=seed)
        self._setup(init, chunk_len, seed, kwargs.get("denormalize_value_during_rollout", False))

    masks_dead_agents = False  # SMACPolicy: log-probabilities of dead agents' steps are -inf (smac_rnn.py:309-311)

    def _setup(self, init, chunk_len, seed, denormalize_value_during_rollout=False):
        """Device network from ``self.spec`` + the bookkeeping every policy of this family shares."""
        self._net = HipNet(self.spec, self.device)
        self._net.load_reference_state(init)
        self._version = -1
        self._chunk_len = chunk_len
        self._seed = int(seed)
        self._rollout_calls = 0
        self._distributed = False
        self._obs_ring: Optional[ObsRing] = None
        self._popart_updates, self._popart_burn_in = 0, float("inf")  # PopArtValueHead defaults (popart.py:16,28-29)
        self._popart_beta = ns.POPART_BETA
        self.denormalize_value_during_rollout = denormalize_value_during_rollout

    # ------------------------------------------------------------------ bookkeeping (api/policy.py:205-288)
    @property
    def default_policy_state(self):
        """Zeros [layers, H] per backbone, batch dimension stripped (actor_critic_policy.py:229-237)."""
        L, H = self.spec.num_rnn_layers, self.spec.rnn_state_width  # LSTM: cat(h, c) (:223-225)
        if not L:
            return None
        z = lambda: np.zeros((L, H), dtype=np.float32)
        return NamedArray(hx=z()) if self.spec.shared_backbone else NamedArray(actor_hx=z(), critic_hx=z())

    def _state_keys(self):
        return (("hx", "a:"),) if self.spec.shared_backbone else (("actor_hx", "a:"), ("critic_hx", "c:"))

    def _rnn_ctx(self, policy_state, T, B, on_reset, chunk: Optional[int] = None, h0=None) -> Optional[RnnCtx]:
        """Chunking of [T, B] rows for the recurrent layers (actor_critic_policy.py:349-363): ``T // chunk_len``
        chunks, each starting from the state stored at its first row; ``on_reset`` [T, B, 1] device uint8 or None."""
        L, H = self.spec.num_rnn_layers, self.spec.rnn_state_width
        if not L:
            return None
        if policy_state is None and h0 is None:
            raise ValueError("recurrent policy: the sample / request carries no policy_state")
        K = max(T // (chunk or self._chunk_len), 1)
        C = T // K
        if K * C != T:
            raise ValueError(f"{T} rows do not split into {K} chunks of equal length (chunk_len={self._chunk_len})")
        if h0 is None:
            h0 = {}
            for key, tag in self._state_keys():
                s = to_device_leaf(policy_state[key], self.device, "real")  # [T, B, L, H]
                h0[tag] = s.reshape(T, B, L, H)[0::C].reshape(K * B, L, H).permute(1, 0, 2).contiguous()
        reset = None
        if on_reset is not None:
            reset = on_reset.reshape(K, C, B).permute(1, 0, 2).contiguous()  # chunk-major [C, K*B]
        return RnnCtx(T, B, C, h0, reset)

    @property
    def version(self) -> int:
        return self._version

    @property
    def net(self) -> HipNet:
        return self._net

    def inc_version(self):
        self._version += 1

    # ------------------------------------------------------------------ PopArt (actor_critic_policy.py:261-274)
    @property
    def popart_head(self):
        if not self.spec.popart:
            raise ValueError("Set popart=True in policy config to activate popart value head.")
        return self._net.popart_state  # float64 [mean(vd), mean_sq(vd), debiasing_term]

    def _popart_map(self, x, normalize):
        rms = self.popart_head
        xd = to_device_leaf(x, self.device, "real")
        out = torch.empty_like(xd)
        hip.popart_map(xd, rms, self.spec.value_dim, out, normalize, ns.POPART_EPS)
        return out

    def normalize_value(self, x):
        return self._popart_map(x, True)

    def denormalize_value(self, x):
        return self._popart_map(x, False)

    def update_popart(self, x, mask):
        """x float32 [..., value_dim], mask uint8/float [..., 1] (1 = counted), as PopArtValueHead.update."""
        vd = self.spec.value_dim
        xd = to_device_leaf(x, self.device, "real").reshape(-1, vd)
        md = None if mask is None else (to_device_leaf(mask, self.device, "real") != 0).to(torch.uint8).reshape(-1)
        stats = torch.zeros((vd, 3), dtype=torch.float64, device=xd.device)
        hip.masked_stats_cols(xd, md, stats, vd)
        if dist.is_initialized() and self._distributed:
            dist.all_reduce(stats)
        self.update_popart_from_stats(stats)

    def update_popart_from_stats(self, stats, count=True):
        """stats float64 [value_dim, 3] = (sum mask, sum x*mask, sum (x*mask)^2), already all-reduced."""
        if count:
            self._popart_updates += 1
        rescale = self._popart_updates + (0 if count else 1) > self._popart_burn_in  # popart.py:49 (inf: never)
        net, head = self._net, self.spec.critic_head
        hip.popart_update(stats, net.popart_state, self.spec.value_dim, self._popart_beta, ns.POPART_EPS,
                          net._p(f"{head.prefix}.weight"), net._p(f"{head.prefix}.bias"), head.in_features, rescale)

    def parameters(self):
        return [self._net.flat]

    def train_mode(self):
        pass

    def eval_mode(self):
        pass

    def load_checkpoint(self, checkpoint):
        self._version = checkpoint.get("steps", 0)
        self._net.load_reference_state(checkpoint["state_dict"])

    def get_checkpoint(self):
        return {"steps": self._version, "state_dict": self._net.reference_state()}

    def distributed(self):
        """Data-parallel replica: adopt rank 0's parameters (what the DDP constructor does, api/policy.py:219-238)."""
        if dist.is_initialized():
            self.broadcast_parameters(src=0)
            self._distributed = True

    def broadcast_parameters(self, src: int = 0, group=None):
        """One flat buffer, one broadcast (RCCL over xGMI on GPU ranks; also what inference replicas call to
        receive fresh parameters from the trainer, replacing the reference's filesystem push/pull)."""
        dist.broadcast(self._net.flat, src=src, group=group)
        self._net.params_changed()
        if self.spec.popart:  # the float64 running statistics are (gradient-less) parameters of the reference's module
            # and travel with them in the DDP constructor's broadcast (api/policy.py:219-238, popart.py:8-59)
            dist.broadcast(self._net.popart_state, src=src, group=group)
        v = torch.tensor([self._version], dtype=torch.int64, device=self._net.flat.device)
        dist.broadcast(v, src=src, group=group)
        self._version = int(v.item())

    # ------------------------------------------------------------------ HBM observation ring
    def attach_obs_ring(self, ring: Optional[ObsRing]):
        """From now on ``rollout`` stages every batch's observations in ``ring`` (in the layout this network's first
        layer reads) and returns their ring references as ``analyzed_result.obs_ref``: the actor stores them with the
        step, the trainer's feed binds the sample's observations to the ring rows instead of uploading the frames a
        second time (runtime/obs_ring.py).  ``None`` detaches."""
        if ring is not None and (dict(ring.layout) != self._net.obs_stage_layout() or
                                 dict(ring.raw_shape) != self._net.obs_raw_shapes()):
            raise ValueError("observation ring was built for a different network (layouts / shapes differ)")
        self._obs_ring = ring

    def make_obs_ring(self, capacity_rows: int, patch_rows: Optional[int] = None) -> ObsRing:
        return ObsRing.for_policy(self, capacity_rows, patch_rows)

    # ------------------------------------------------------------------ inference
    # (2048: the call is bound by the host's launches per piece as much as by the link -- 4096 rows in 2 pieces 2.74-2.92 ms,
    # in 4 pieces 3.08-3.10, in 8 pieces 4.3, same box; SRL_ROLLOUT_PIECE for the A/B)
    RING_PREV_KEY = "ring_prev"
    ROLLOUT_PIECE = int(os.environ.get('SRL_ROLLOUT_PIECE', '2048'))  # rows per piece when a big host batch is streamed in (copy of piece i+1 under the compute of i)

    def rollout(self, requests: policy_api.RolloutRequest, **kwargs) -> policy_api.RolloutResult:
        return self.rollout_async(requests, **kwargs).result()

    def rollout_async(self, requests: policy_api.RolloutRequest, **kwargs) -> "PendingRollout":
        """``rollout`` in two halves: everything is ISSUED on the current stream here -- the requests' H2D copies, the network pass,
        the sampling, the results' D2H copies into pinned blocks -- and ``PendingRollout.result()`` waits for that stream position
        and builds the ``RolloutResult``.  A policy worker that serves two groups of actors through two policies (one per stream,
        one shared observation ring) keeps one group's link transfers under the other's kernels from ONE host thread: the reference's
        actors do the same with their environment ring (actor_worker.py:634-748).  One call in flight per policy."""
        hip.require_gpu()
        if self.__dict__.get("_pending_rollout"):
            raise RuntimeError("rollout_async: the previous call's result() has not been taken (one call in flight per policy)")
        self._net._serving[0] += 1   # what the executor derives from the parameters survives from one request batch to the next
        try:
            pend = self._rollout(requests, **kwargs)
        except BaseException:
            self._net._serving[0] -= 1
            raise
        self._pending_rollout = True
        return pend

    def _rollout(self, requests: policy_api.RolloutRequest, **kwargs) -> "PendingRollout":
        host = {k: v for k, v in requests.obs.items() if v is not None}
        n = int(next(iter(host.values())).shape[0])
        # Stack-aware requests (atari_wrappers.py:211-242 `FrameStack`): `ring_prev` [n, 1] int64 holds the observation-ring
        # stamp of the same environment's previous observation (0 at an episode start) and the frame-stack keys carry only
        # their newest plane [n, 1, H, W]; the ring assembles the rows (ObsRing.put_stacked) -- a quarter of the bytes over
        # the host link that actor_critic_policy.py:467-469 moves, same actions, log-probabilities and values.  A request whose
        # predecessors the ring no longer holds raises `obs_ring.WholeStackNeeded` (dead or -1 stamps must not be sent).
        prev = host.pop(self.RING_PREV_KEY, None)
        if prev is not None and self._obs_ring is None:
            raise ValueError("stack-aware requests (`ring_prev`) need an observation ring attached to the policy")
        if prev is not None:
            prev = (prev.cpu().numpy() if isinstance(prev, torch.Tensor) else np.asarray(prev)).astype(np.int64).reshape(n)
        # (stack-aware requests carry a quarter of the bytes: they go through in pieces only from four times the rows)
        if (not self.spec.num_rnn_layers and n >= (8 if prev is not None else 2) * self.ROLLOUT_PIECE
                and not any(isinstance(v, torch.Tensor) and v.is_cuda for v in host.values())):
            action, logp, value, refs = self._rollout_streamed(host, n, requests.is_evaluation, prev)
            state = None
        else:
            obs = {k: to_device_leaf(v, self.device, "obs") for k, v in host.items()}
            state = None
            if self.spec.num_rnn_layers:  # requests carry [n, layers, H]; the state is used as given (:473-481)
                ps = requests.policy_state
                if ps is None:
                    raise ValueError("recurrent policy: the request carries no policy_state")
                state = {k: np.asarray(ps[k]) for k, _ in self._state_keys()}
            action, logp, value, refs = self._rollout_rows(obs, n, requests.is_evaluation, state, prev=prev)
            state = self._packed_last_state()
        views, done = self._results_to_host_async(action, logp, value)
        return PendingRollout(self, views, done, refs, state, n)

    def _results_to_host_async(self, *tensors):
        """Device results -> pinned blocks kept per policy: asynchronous copies and ONE event behind them (three `.cpu()` calls are
        three synchronisations through pageable staging)."""
        pins = self.__dict__.setdefault("_result_pins", {})
        out = []
        for i, t in enumerate(tensors):
            key = (i, t.dtype)
            pin = pins.get(key)
            if pin is None or pin.numel() < t.numel():
                pin = pins[key] = torch.empty(max(t.numel(), 1024), dtype=t.dtype).pin_memory()
            view = pin[:t.numel()].view(t.shape)
            view.copy_(t, non_blocking=True)
            out.append(view)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream())
        return out, done

    def _rollout_streamed(self, host, n, is_evaluation, prev=None):
        """A big batch of host observations (the policy worker's 10 240-request batches are 289 MB of frames): rows
        go through in pieces, the H2D copy of piece i+1 on a side stream under the network pass of piece i, so the
        call costs about max(copy, compute) instead of their sum.  The pieces share the call's Philox offset and number their rows
        within the whole batch (`row0`): the sampled actions do not depend on the split."""
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream()
        is_eval = np.asarray(is_evaluation).reshape(-1)
        if is_eval.size == 1:
            is_eval = np.broadcast_to(is_eval, (n,))
        heads = self.spec.act_dims
        cont = bool(self.spec.std_type)
        action = torch.empty((n, sum(heads) if cont else len(heads)), dtype=torch.float32 if cont else torch.int64,
                             device=self.device)
        logp = torch.empty((n, 1), dtype=torch.float32, device=self.device)
        value = torch.empty((n, self.spec.value_dim), dtype=torch.float32, device=self.device)
        arrays = {}
        for k, v in host.items():
            if isinstance(v, torch.Tensor):  # a host tensor (pinned: the copies below are then true asynchronous DMA; a numpy
                # view of pinned memory is pageable in torch's eyes and goes through a staging buffer)
                t = v.view(torch.uint8) if v.dtype == torch.bool else v
                if k == "available_action" and t.dtype != torch.uint8:
                    t = t.to(torch.uint8)
                elif k != "available_action" and t.dtype not in (torch.uint8, torch.float32):
                    t = t.to(torch.float32)
                arrays[k] = t.contiguous()
                continue
            a = np.asarray(v)
            if a.dtype == np.bool_:
                a = a.view(np.uint8)
            if k == "available_action" and a.dtype != np.uint8:
                a = a.astype(np.uint8)  # converted on the host: no side-stream temporary inside _rollout_rows
            elif k != "available_action" and a.dtype != np.uint8 and a.dtype != np.float32:
                a = a.astype(np.float32)
            arrays[k] = torch.from_numpy(np.ascontiguousarray(a))
        step = self.ROLLOUT_PIECE
        bounds = [(r0, min(n, r0 + step)) for r0 in range(0, n, step)]
        staged = [None, None]  # the piece being computed and the piece being copied (fresh allocations of the side
        # stream's pool, handed to the main stream with record_stream)

        def stage(i):
            r0, r1 = bounds[i]
            with torch.cuda.stream(self._copy_stream):
                dev = {k: a[r0:r1].to(self.device, non_blocking=True) for k, a in arrays.items()}
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
            staged[i & 1] = (dev, ev)

        stage(0)
        refs = [] if self._obs_ring is not None else None
        for i, (r0, r1) in enumerate(bounds):
            dev, ev = staged[i & 1]
            main.wait_event(ev)
            a_i, l_i, v_i, r_i = self._rollout_rows(dev, r1 - r0, is_eval[r0:r1], None, prev=None if prev is None else prev[r0:r1],
                                                    row0=r0, count_call=i + 1 == len(bounds))
            if refs is not None:
                refs.append(r_i)
            action[r0:r1].copy_(a_i)
            logp[r0:r1].copy_(l_i)
            value[r0:r1].copy_(v_i)
            for t in dev.values():
                t.record_stream(main)
            if i + 1 < len(bounds):
                stage(i + 1)  # issued after piece i's launches: a pageable copy blocks the host, not the GPU
        return action, logp, value, (None if refs is None else np.concatenate(refs))

    def _rollout_rows(self, obs, n, is_evaluation, state, prev=None, row0=0, count_call=True):
        """One inference pass over ``n`` independent rows: device ``(action, log_prob [n,1], value [n,vd])`` and the rows'
        observation-ring references (int64 numpy [n], or None without a ring); the new recurrent states are left in
        ``net.last_state``.  ``state``: ``{key: [n, layers, W]}`` host or device, or None."""
        obs = dict(obs)  # the caller keeps its dict whole: a streamed piece must still own EVERY staged tensor (the mask
        # included) when it hands them to the main stream with record_stream
        avail = obs.pop("available_action", None)
        if avail is not None and avail.dtype != torch.uint8:
            avail = avail.to(torch.uint8)
        is_eval = np.asarray(is_evaluation).reshape(-1)
        is_eval = to_device_leaf(np.broadcast_to(is_eval, (n,)) if is_eval.size == 1 else is_eval, self.device, "flag")
        refs = None
        if prev is not None:  # stack-aware: the ring assembles each row from its predecessor and the uploaded plane
            ring = self._obs_ring
            planes = {k: obs[k] for k in ring.keys() if tuple(obs[k].shape[1:]) != ring.raw_shape[k]}
            try:
                refs, staged = ring.put_stacked(planes, prev, full={k: obs[k] for k in ring.keys() if k not in planes})
            except (LookupError, BufferError) as e:
                # planes alone cannot be served any other way (the non-stacked path degrades through `put_or_skip`; here there
                # is no whole row to fall back on): a defined per-request error, nothing staged, no slot consumed
                from srl_amd.runtime.obs_ring import WholeStackNeeded
                raise WholeStackNeeded(f"stack-aware request refused ({e}); send whole frame stacks without `ring_prev`") from e
            obs.update(staged)
        elif self._obs_ring is not None:  # the rows stay in HBM for the trainer; the forward below reads them from there
            # a ring full of rows a training step has leased does not fail the request: the rows go unstaged (stamp -1), the
            # forward reads the batch's own rows, and the trainer uploads them from the sample when it binds it
            refs, staged = self._obs_ring.put_or_skip({k: obs[k] for k in self._obs_ring.keys()})
            if staged is not None:
                obs.update(staged)
        rnn = None
        if self.spec.num_rnn_layers:
            rnn = self._rnn_ctx(NamedArray(**{k: v[None] for k, v in state.items()}), 1, n, None)
        logits, value = self._net.forward(obs, n, keep_tape=False, rnn=rnn)
        heads = self.spec.act_dims
        logp = torch.empty((n, 1), dtype=torch.float32, device=self.device)
        if self.spec.std_type:  # Normal(mean, std): the mean when evaluating, a sample otherwise (:499-506)
            action = torch.empty((n, sum(heads)), dtype=torch.float32, device=self.device)
            ptr, ld = self._log_std()
            hip.gaussian_sample(logits, ptr, ld, is_eval, self._seed, self._rollout_calls, action, logp, row0=row0)
        else:
            action = torch.empty((n, len(heads)), dtype=torch.int64, device=self.device)
            hip.categorical_sample(logits, avail, is_eval, heads, self._seed, self._rollout_calls, action, logp, row0=row0)
        if count_call:  # one Philox offset per rollout() call: the pieces of a streamed batch share it and differ by row0
            self._rollout_calls += 1
        return action, logp, value, refs

    def _rnn_ctx_with_burn_in(self, obs, avail_unused, policy_state, on_reset, burn, T, B) -> Optional[RnnCtx]:
        """Recurrent context for rows [burn, burn + T) of leaves that start `burn` rows earlier: the `burn` rows before
        every chunk are replayed without gradient from the state stored at their first row, and what comes out is
        the chunk's initial state (actor_critic_policy.py:365-378).  obs / policy_state / on_reset: device leaves
        [burn + T (+ more), B, ...]."""
        if not self.spec.num_rnn_layers:
            return None
        if not burn:
            ps = NamedArray(**{k: v[:T] for k, v in policy_state.items()})
            return self._rnn_ctx(ps, T, B, on_reset[:T])
        Cl = self._chunk_len
        K = max(T // Cl, 1)
        def win(x):  # [K*burn, B, ...] time-major
            parts = [x[i * Cl:i * Cl + burn] for i in range(K)]
            return RingObs.cat(parts) if isinstance(x, RingObs) else torch.cat(parts, dim=0)

        n = K * burn * B
        w_obs = {k: win(v).reshape(n, *v.shape[2:]) for k, v in obs.items() if k != "available_action"}
        w_ps = NamedArray(**{k: win(v) for k, v in policy_state.items()})
        ctx = self._rnn_ctx(w_ps, K * burn, B, win(on_reset), chunk=burn)
        self._net.forward(w_obs, n, keep_tape=False, rnn=ctx)
        h0 = {tag: self._net.last_state[tag].clone() for _, tag in self._state_keys()}
        return self._rnn_ctx(None, T, B, on_reset[burn:burn + T], h0=h0)

    # ------------------------------------------------------------------ action distribution on top of the actor head
    def action_kind(self) -> str:
        return "real" if self.spec.std_type else "index"  # continuous actions are float32 [.., A]

    def _log_std(self, net=None):
        """(pointer, row pitch) of log sigma: the shared vector (pitch 0) or the second head's rows."""
        net = net or self._net
        if self.spec.std_type == "shared_learnable":
            return net.log_std_rows.data_ptr(), net.log_std_rows.stride(0)
        return net._p("log_std"), 0

    def dist_fwd(self, logits, action, avail, logp, ent, net=None):
        """log-probability of `action` and entropy under the current head outputs (:311-324).  ``net``: the executor whose
        forward pass produced ``logits`` (default: the policy's own; the trainer's second row-chunk pipeline passes its twin)."""
        if self.spec.std_type:
            ptr, ld = self._log_std(net)
            hip.gaussian_fwd(logits, ptr, ld, action, logp, ent)
        else:
            hip.categorical_fwd(logits, action, avail, self.spec.act_dims, logp, ent)

    def dist_bwd(self, logits, action, avail, d_lp, d_ent, d_logits, net=None):
        """d loss / d head outputs; returns what ``HipNet.backward`` needs besides d_logits (d log sigma rows or None)."""
        if not self.spec.std_type:
            hip.categorical_bwd(logits, action, avail, self.spec.act_dims, d_lp, d_ent, d_logits)
            return None
        net = net or self._net
        n, A = logits.shape
        ptr, ld = self._log_std(net)
        d_ls = net.ws.get("d_log_std_rows", n * A)[:n * A].view(n, A)
        hip.gaussian_bwd(logits, ptr, ld, action, d_lp, d_ent, d_logits, d_ls)
        if self.spec.std_type == "shared_learnable":
            return d_ls
        if self.spec.std_type == "separate_learnable":  # `fixed`: requires_grad=False in the reference, no gradient
            hip.colsum(d_ls.data_ptr(), A, n, A, net._g("log_std"), accumulate=True)
        return None

    def _packed_last_state(self):
        """New hidden states as the actors store them: numpy [n, layers, H] per backbone (:505-508)."""
        if not self.spec.num_rnn_layers:
            return None
        return NamedArray(**{k: self._net.last_state[tag].permute(1, 0, 2).cpu().numpy() for k, tag in self._state_keys()})

    # ------------------------------------------------------------------ training-side analysis
    def analyze(self, sample, target="ppo", **kwargs):
        """actor_critic_policy.py:272-296."""
        if target == "ppo":
            return self._ppo_analyze(sample, **kwargs)
        if target == "ppg_ppo_phase":
            return self._ppg_phase1_analyze(sample, **kwargs)
        if target == "ppg_aux_phase":
            return self._ppg_phase2_analyze(sample, **kwargs)
        raise ValueError(f"Analyze method for algorithm {target} not implemented for {self.__class__.__name__}")

    def _ppg_phase1_analyze(self, sample, **kwargs):
        """actor_critic_policy.py:392-415: the PPO analysis plus the auxiliary head's values (which take no gradient there)."""
        if self.spec.aux_head is None:
            raise RuntimeError("Cannot run ppg analysis without auxiliary_head. Try setting auxiliary_head=True.")
        ar = self._ppo_analyze(sample, **kwargs)
        T, B = ar.new_action_log_probs.shape[:2]
        reward = to_device_leaf(sample.reward, self.device, "real")
        return PPGPhase1AnalyzedResult(old_action_log_probs=ar.old_action_log_probs, new_action_log_probs=ar.new_action_log_probs,
                                       aux_values=self._net.aux_value.view(T, B, -1), state_values=ar.state_values,
                                       reward=reward, entropy=ar.entropy)

    def _ppg_phase2_analyze(self, sample, **kwargs):
        """actor_critic_policy.py:417-435: the current action distributions (per head, as normalised log-probabilities -- with
        unavailable actions at -1e10 before the normalisation, :135-136), the auxiliary and the critic head's values.  ``sample``
        carries ``obs``, ``on_reset`` and, for a recurrent policy, ``policy_state`` (the auxiliary phase's cache entry).  The
        forward context is kept: ``backward_ppg_aux`` follows."""
        if self.spec.aux_head is None:
            raise RuntimeError("Cannot run ppg analysis without auxiliary_head. Try setting auxiliary_head=True.")
        if self.spec.std_type:
            raise NotImplementedError("the auxiliary phase is built for categorical action heads")
        T, B = sample.on_reset.shape[0], sample.on_reset.shape[1]
        n = T * B
        obs = {k: to_device_leaf(v, self.device, "obs").reshape(n, *v.shape[2:]) for k, v in sample.obs.items() if v is not None}
        avail = obs.pop("available_action", None)
        obs.pop("is_alive", None)
        rnn = None
        if self.spec.num_rnn_layers:
            if sample.policy_state is None:
                raise ValueError("recurrent policy: the sample carries no policy_state")
            ps = {k: to_device_leaf(v, self.device, "real") for k, v in sample.policy_state.items()}
            rnn = self._rnn_ctx_with_burn_in({k: to_device_leaf(v, self.device, "obs") for k, v in sample.obs.items() if v is not None},
                                             None, ps, to_device_leaf(sample.on_reset, self.device, "flag"), 0, T, B)
        logits, value = self._net.forward(obs, n, keep_tape=True, rnn=rnn)
        logq = torch.empty_like(logits)
        hip.categorical_log_softmax(logits, avail, self.spec.act_dims, logq)
        self._analysis = (logits, None, avail, n)
        dists, start = [], 0
        for d in self.spec.act_dims:
            dists.append(CategoricalLogits(logq[:, start:start + d].reshape(T, B, d)))
            start += d
        self._aux_logq = logq
        return PPGPhase2AnalyzedResult(action_dists=dists, auxiliary_value=self._net.aux_value.view(T, B, -1),
                                       predicted_value=value.view(T, B, -1))

    def backward_ppg_aux(self, d_logits: torch.Tensor, d_aux: torch.Tensor, d_value: torch.Tensor):
        """Back-propagate the auxiliary loss's gradients with respect to the raw logits, the auxiliary values and the critic head's
        values of the last ``analyze(target="ppg_aux_phase")`` into ``net.grad``."""
        _, _, _, n = self._analysis
        self._net.backward(d_logits.reshape(n, -1), d_value.reshape(n, -1), None, d_aux=d_aux.reshape(n, -1))
        self._analysis = None

    def _ppo_analyze(self, sample, burn_in_steps=0, **kwargs):
        """New log-probs, state values and entropy for every row of ``sample`` (leaves ``[T, B, ...]``).

        Without a recurrent backbone the reference's chunking (``to_chunk`` / ``back_to_trajectory``,
        ``modules/utils.py:164-195``) only permutes independent rows, so rows are evaluated in place.
        The forward context is kept so that ``backward_ppo`` can follow.
        """
        burn = int(burn_in_steps)
        # [T, B, agents, ...] leaves of a shared multi-agent environment: agents are merged into the batch axis for the
        # network and split again on the way out (smac_rnn.py:253-266, :313-320); a free view on contiguous leaves
        agents = sample.on_reset.shape[2] if len(sample.on_reset.shape) == 4 else 0
        mg = (lambda t: t.reshape(t.shape[0], t.shape[1] * t.shape[2], *t.shape[3:])) if agents else (lambda t: t)
        Tall, B = sample.on_reset.shape[0], sample.on_reset.shape[1] * max(agents, 1)
        T = Tall - burn  # analysed rows: [burn, Tall) (:346-349)
        n = T * B
        obs, full_obs = {}, {}
        for k, v in sample.obs.items():
            if v is None:
                continue
            t = mg(to_device_leaf(v, self.device, "obs"))
            full_obs[k] = t
            obs[k] = t[burn:].reshape(n, *t.shape[2:])
        avail = obs.pop("available_action", None)
        alive = obs.pop("is_alive", None)
        action = mg(to_device_leaf(sample.action.x, self.device, self.action_kind()))[burn:].reshape(n, -1)
        rnn = None
        if self.spec.num_rnn_layers:
            if sample.policy_state is None:
                raise ValueError("recurrent policy: the sample carries no policy_state")
            ps = {k: mg(to_device_leaf(v, self.device, "real")) for k, v in sample.policy_state.items()}
            rnn = self._rnn_ctx_with_burn_in(full_obs, None, ps, mg(to_device_leaf(sample.on_reset, self.device, "flag")),
                                             burn, T, B)
        logits, value = self._net.forward(obs, n, keep_tape=True, rnn=rnn)
        logp = self._net.ws.get("new_logp", n)[:n]
        ent = self._net.ws.get("entropy", n)[:n]
        self.dist_fwd(logits, action, avail, logp, ent)
        self.mask_dead(logp, alive)
        self._analysis = (logits, action, avail, n)
        old = sample.analyzed_result.log_probs
        old = None if old is None else to_device_leaf(old, self.device, "real")[burn:]
        shape = (T, B // agents, agents) if agents else (T, B)
        return SampleAnalyzedResult(old_action_log_probs=old, new_action_log_probs=logp.view(*shape, 1),
                                    state_values=value.view(*shape, -1), entropy=ent.view(*shape, 1))

    def mask_dead(self, logp, alive):
        """Policies of shared multi-agent environments: no policy gradient through a dead agent's steps -- its new
        log-probability becomes -inf, the ratio 0 (smac_rnn.py:309-311).  ``alive``: rows of ``obs.is_alive`` or None."""
        if self.masks_dead_agents and alive is not None:
            logp.masked_fill_(alive.reshape(-1) == 0, float("-inf"))

    def backward_ppo(self, d_new_lp: torch.Tensor, d_value: torch.Tensor, d_entropy: torch.Tensor):
        """Back-propagate d loss / d(new log-prob, value, entropy) of the last ``analyze`` into ``net.grad``."""
        logits, action, avail, n = self._analysis
        d_logits = self._net.ws.get("d_logits", logits.numel())[:logits.numel()].view_as(logits)
        d_ls = self.dist_bwd(logits, action, avail, d_new_lp.reshape(n), d_entropy.reshape(n), d_logits)
        self._net.backward(d_logits, d_value.reshape(n, -1), d_ls)
        self._analysis = None


policy_api.register("actor-critic", ActorCriticPolicy)
policy_api.register("actor-critic-separate",
                    functools.partial(ActorCriticPolicy, shared_backbone=False, auxiliary_head=False))
policy_api.register("gym_mujoco", functools.partial(ActorCriticPolicy, continuous_action=True))  # :538
policy_api.register("actor-critic-separate-continuous-action",
                    functools.partial(ActorCriticPolicy, shared_backbone=False, auxiliary_head=False,
                                      continuous_action=True))  # :539-540
policy_api.register("actor-critic-shared",
                    functools.partial(ActorCriticPolicy, shared_backbone=True, auxiliary_head=False))
policy_api.register("actor-critic-auxiliary",
                    functools.partial(ActorCriticPolicy, shared_backbone=False, auxiliary_head=True))  # :535-536
