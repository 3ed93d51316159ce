"""``MultiAgentPPO`` on the HIP kernels, registered as trainer ``"mappo"`` (and ``"mappo-hip"``).

Mirror of reference ``legacy/algorithm/ppo/mappo.py``: same keyword hyper-parameters and defaults
(``:68-116``), same ``step(sample) -> TrainerStepResult`` contract including the in-place mutations of the
sample (``truncated`` default ``:222-223``; ``analyzed_result.adv/ret`` reset / written back as numpy,
``:224-225,254-257``), same statistics keys (``:36-47,293-324``), ``get_checkpoint`` / ``load_checkpoint``
(``:58-66``) with a torch-Adam-compatible ``optimizer_state_dict``.

What runs where (one process per GPU):

1. the sample's leaves go to HBM in their wire dtypes (no float32 widening, no one-call pipeline delay);
2. ``srl_gae_scan``: value masking + GAE scan + returns + local advantage statistics in one launch;
3. data parallel: ONE all-reduce of the 3 float64 statistics (the reference issues three, ``utils.py:58-61``);
4. per row-chunk: network forward, ``srl_categorical_fwd``, ``srl_ppo_loss_fwd_bwd`` (loss, its gradient and
   the logging sums in one pass), ``srl_categorical_bwd``, network backward into one flat gradient buffer;
5. data parallel: one all-reduce of the flat gradient (mean over ranks = DDP semantics, ``api/policy.py:219-238``);
6. ``srl_grad_sumsq`` + ``srl_adam_step``: global-norm clip fused into the Adam update;
7. a single device->host copy of the loss terms per epoch (the reference syncs ~11 times, ``:293-299``).

Loss-normalisation semantics under data parallelism are the reference's: every rank divides its masked
sums by its LOCAL mask count, gradients are then averaged over ranks, while the advantage normalisation
uses GLOBAL statistics.
"""
import logging
import contextlib
import os
from collections import defaultdict

import numpy as np
import torch
import torch.distributed as dist

from srl_amd import hip
from srl_amd.algorithm import netspec as ns
from srl_amd.algorithm.actor_critic import ActorCriticPolicy, to_device_leaf, wire_leaf
from srl_amd.api.trainer import PytorchTrainer, TrainerStepResult, register
from srl_amd.namedarray import recursive_apply
from srl_amd.runtime.obs_ring import RingObs

logger = logging.getLogger("MAPPO")

_STAT_TERMS = (("policy_loss", hip.LT_POLICY), ("value_loss", hip.LT_VALUE), ("entropy", hip.LT_ENTROPY),
               ("clip_ratio", hip.LT_CLIP), ("importance_weight", hip.LT_RATIO), ("advantage", hip.LT_ADV),
               ("value_targets", hip.LT_RET))


class _BucketReducer:
    """Gradient all-reduce overlapped with the backward pass (what DDP's bucketing does for the reference,
    api/policy.py:219-238): the flat gradient is cut into contiguous buckets in parameter order; as soon as every
    parameter of a bucket has its final gradient the bucket's all-reduce is launched asynchronously (RCCL runs it on
    its own stream while the compute stream continues with the earlier layers); `finish` launches what is left and
    makes the compute stream wait.  Sum over ranks; the mean is folded into the Adam kernel.

    Several row-chunk pipelines (`begin(grads=[net.grad, twin.grad, ...])`): every pipeline accumulates into its own
    gradient buffer on its own stream, so a bucket is final when the LAST chunk of EACH pipeline has announced its
    parameters.  Each announcement leaves an event on its pipeline's stream; when the last pipeline completes a bucket, a
    reduction stream waits for those events, folds the other pipelines' slices into the first buffer (srl_accumulate on
    just that slice) and launches the slice's all-reduce from there -- none of the pipelines waits.

    The LAST bucket to become final (the first layers' parameters, at the very end of the backward pass) has nothing left to
    overlap with: the clip and the optimiser step need it.  `finish` closes it ON the compute stream -- wait for the pipelines'
    tails, fold, all-reduce, all in stream order -- instead of through the reduction stream and the communicator's side stream
    and back (three cross-stream dependencies in a row at the one place of an update where the GPU has nothing else to run;
    SRL_LAST_BUCKET_INLINE=0: as before, A/B)."""
    LAST_INLINE = os.environ.get("SRL_LAST_BUCKET_INLINE", "1") != "0"

    def __init__(self, net, bucket_bytes, comm=None):
        self.net = net
        self.comm = comm  # srl_amd.comm.NativeComm (RCCL through the C ABI on a side stream) or None (torch.distributed)
        self.buckets = []  # [lo, hi, frozenset of the parameter prefixes inside]
        self.stats = dict(epochs=0, launched_in_backward=0, launched_in_finish=0, slices_folded=0, closed_inline=0)
        self._fold_stream = None
        lo, pending, size = None, set(), 0
        for name, info in net.spec.params.items():
            prefix = name.rsplit(".", 1)[0]
            if ".rnn._AutoResetRNN__net" in name:
                prefix = name[:name.index("._AutoResetRNN__net") + len("._AutoResetRNN__net")]
            if lo is not None and size + 4 * info.numel > bucket_bytes:  # a big tensor starts its own bucket
                self.buckets.append([lo, info.offset, pending])
                lo, pending, size = None, set(), 0
            if lo is None:
                lo = info.offset
            pending.add(prefix)
            size += 4 * info.numel
            hi = info.offset + (info.numel + 3) // 4 * 4
            if size >= bucket_bytes:
                self.buckets.append([lo, hi, pending])
                lo, pending, size = None, set(), 0
        if lo is not None:
            self.buckets.append([lo, net.spec.total_params, pending])
        self.buckets = [(lo, hi, frozenset(p)) for lo, hi, p in self.buckets]
        self.begin()
        self.stats["epochs"] = 0

    def begin(self, grads=None):
        """Start of an epoch's backward pass: nothing launched, every parameter pending in every pipeline.  (The bucket
        layout is built once per trainer; only this per-epoch state is reset.)  ``grads``: the pipelines' flat gradient
        buffers, the first one is the buffer that is reduced; None = one pipeline, ``net.grad``."""
        self.grads = [self.net.grad] if grads is None else list(grads)
        self.pending = [[set(b[2]) for b in self.buckets] for _ in self.grads]
        self.events = [[None] * len(self.buckets) for _ in self.grads]
        self.launched = [False] * len(self.buckets)
        self.works = []
        self._finishing = False
        self.stats["epochs"] += 1

    def _fold(self, i):
        """The other pipelines' slices of bucket ``i`` into the first buffer, on the reduction stream, after the point each
        pipeline's stream had reached when it completed the bucket (`finish`: after their tails)."""
        if self._fold_stream is None:
            self._fold_stream = torch.cuda.Stream(device=self.grads[0].device)
        lo, hi, _ = self.buckets[i]
        for ev in (pipe[i] for pipe in self.events):
            if ev is not None:
                self._fold_stream.wait_event(ev)
        with torch.cuda.stream(self._fold_stream):
            hip.accumulate_n(self.grads[0][lo:hi], [g[lo:hi] for g in self.grads[1:]])   # one launch, added in pipeline order
        self.stats["slices_folded"] += len(self.grads) - 1

    def _launch(self, i):
        lo, hi, _ = self.buckets[i]
        many = len(self.grads) > 1
        if many:
            self._fold(i)
        with torch.cuda.stream(self._fold_stream) if many else contextlib.nullcontext():
            if self.comm is not None:
                self.comm.all_reduce_f32_async(self.grads[0][lo:hi])  # picks up after the current stream's tail
            else:
                self.works.append(dist.all_reduce(self.grads[0][lo:hi], async_op=True))
        self.launched[i] = True
        self.stats["launched_in_finish" if self._finishing else "launched_in_backward"] += 1

    def ready(self, prefixes, pipe=0):
        """Pipeline ``pipe``'s gradients of these parameters are final in its stream order (called from the backward pass of
        that pipeline's last chunk, on its stream)."""
        many = len(self.grads) > 1
        for i, pend in enumerate(self.pending[pipe]):
            if not self.launched[i] and pend:
                pend.difference_update(prefixes)
                if not pend:
                    if many:
                        self.events[pipe][i] = self._mark()
                    if all(not other[i] for other in self.pending):
                        if self.LAST_INLINE and sum(self.launched) == len(self.buckets) - 1:
                            continue   # the last one: `finish`, on the compute stream
                        self._launch(i)

    @staticmethod
    def _mark():
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev

    def hook(self, pipe):
        return self.ready if pipe == 0 else (lambda prefixes: self.ready(prefixes, pipe))

    def finish(self, streams=()):
        """Launch what the backward passes did not release (``streams``: the pipelines' streams, all of their work enqueued),
        then make the current stream wait for every bucket."""
        self._finishing = True
        rest = [i for i in range(len(self.buckets)) if not self.launched[i]]
        if self.LAST_INLINE and len(rest) == 1 and all(not pend[rest[0]] for pend in self.pending):
            # released by every pipeline and held back by `ready`
            i = rest[0]
            lo, hi, _ = self.buckets[i]
            cur = torch.cuda.current_stream()
            for st in streams:
                cur.wait_stream(st)
            if len(self.grads) > 1:
                hip.accumulate_n(self.grads[0][lo:hi], [g[lo:hi] for g in self.grads[1:]])
                self.stats["slices_folded"] += len(self.grads) - 1
            if self.comm is not None:
                self.comm.all_reduce_f32_inline(self.grads[0][lo:hi])
            else:   # (the process group's stream picks up behind the current one, and the current one waits for it)
                dist.all_reduce(self.grads[0][lo:hi])
            self.launched[i] = True
            self.stats["closed_inline"] += 1
        if len(self.grads) > 1 and not all(self.launched):
            if self._fold_stream is None:
                self._fold_stream = torch.cuda.Stream(device=self.grads[0].device)
            # every pipeline's tail, the FIRST one's (the current stream) included: a bucket nobody released carries no event
            # from any pipeline, and its fold + all-reduce read all of their buffers
            self._fold_stream.wait_stream(torch.cuda.current_stream())
            for st in streams:
                self._fold_stream.wait_stream(st)
        for i in range(len(self.buckets)):
            if not self.launched[i]:
                self._launch(i)
        if self.comm is not None:
            self.comm.join()  # the compute stream waits for the side stream's collectives
        for w in self.works:
            w.wait()
        if len(self.grads) > 1:
            torch.cuda.current_stream().wait_stream(self._fold_stream)


class MultiAgentPPO(PytorchTrainer):

    def __init__(self, policy: ActorCriticPolicy, **kwargs):
        super().__init__(policy)
        g = kwargs.get
        self.discount_rate = g("discount_rate", 0.99)
        self.gae_lambda = g("gae_lambda", 0.97)
        self.eps_clip = g("eps_clip", 0.2)
        self.clip_value = g("clip_value", False)
        self.dual_clip = g("dual_clip", True)
        self.c_clip = g("c_clip", 3)
        self.burn_in_steps = g("burn_in_steps", 0)
        self.vtrace = g("vtrace", False)
        self.recompute_adv_on_reuse = g("recompute_adv_on_reuse", True)
        self.recompute_adv_among_epochs = g("recompute_adv_among_epochs", False)
        self.normalize_old_value = g("normalize_old_value", False)
        if self.clip_value and self.normalize_old_value != getattr(policy, "denormalize_value_during_rollout", False):
            raise ValueError(
                "Trainer `normalize_old_value` and policy `denormalize_value_during_rollout` should be consistent!")
        self.value_eps_clip = g("value_eps_clip", self.eps_clip)
        self.value_loss_weight = g('value_loss_weight', 0.5)
        self.entropy_bonus_weight = g('entropy_bonus_weight', 0.01)
        self.entropy_decay_per_steps = g("entropy_decay_per_steps", None)
        self.entropy_bonus_decay = g("entropy_bonus_decay", 0.99)
        self.max_grad_norm = g('max_grad_norm')
        self.popart = g('popart', False)
        self.bootstrap_steps = g("bootstrap_steps", 1)
        self.ppo_epochs = g("ppo_epochs", 1)
        if self.popart and not policy.net.spec.popart:
            raise ValueError("Set popart=True in policy config to activate popart value head.")  # actor_critic_policy.py:264
        if self.vtrace and self.bootstrap_steps != 1 and policy.net.spec.num_rnn_layers:
            raise NotImplementedError("V-trace with bootstrap_steps != 1 and a recurrent policy is not on the HIP path")
        if self.burn_in_steps and self.vtrace:
            # the reference itself cannot run this: analyze() drops the burn-in rows, so imp_ratio [Tb - 1 - burn] meets
            # delta [Tb - 1] in gae_trace (gae.py:63-65) and the shapes clash
            raise NotImplementedError("burn-in together with V-trace: the reference's own shapes clash (mappo.py:243-246, gae.py:65)")

        # optimiser (modules/utils.py:268-286: torch.optim.{Adam, AdamW, RMSprop, SGD}(**optimizer_config))
        name = g('optimizer', 'adam')
        if name not in ('adam', 'adamw', 'rmsprop', 'sgd'):
            raise AssertionError(f"Optimizer name {name} does not match any implemented optimizers "
                                 "(['adam', 'rmsprop', 'sgd', 'adamw']).")
        cfg = dict(g('optimizer_config', {}))
        self._opt = name
        self._adamw = name == 'adamw'
        self._betas, self._eps = (0.9, 0.999), 1e-8
        if name in ('adam', 'adamw'):
            self._lr = cfg.pop("lr", 1e-3)
            self._betas = tuple(cfg.pop("betas", (0.9, 0.999)))
            self._eps = cfg.pop("eps", 1e-8)
            self._weight_decay = cfg.pop("weight_decay", 1e-2 if self._adamw else 0.0)
        elif name == 'rmsprop':  # torch defaults: lr 1e-2, alpha 0.99, eps 1e-8, no momentum, not centred
            self._lr = cfg.pop("lr", 1e-2)
            self._alpha = cfg.pop("alpha", 0.99)
            self._eps = cfg.pop("eps", 1e-8)
            self._weight_decay = cfg.pop("weight_decay", 0.0)
            self._momentum = cfg.pop("momentum", 0.0)
            self._centered = bool(cfg.pop("centered", False))
        else:  # sgd
            self._lr = cfg.pop("lr", 1e-3)
            self._momentum = cfg.pop("momentum", 0.0)
            self._dampening = cfg.pop("dampening", 0.0)
            self._weight_decay = cfg.pop("weight_decay", 0.0)
            self._nesterov = bool(cfg.pop("nesterov", False))
            if self._nesterov and (self._momentum <= 0 or self._dampening != 0):
                raise ValueError("Nesterov momentum requires a momentum and zero dampening")  # torch/optim/sgd.py
        for flag in ("amsgrad", "maximize", "capturable", "differentiable", "fused", "foreach"):
            if cfg.get(flag) in (None, False):
                cfg.pop(flag, None)
        if cfg:
            raise NotImplementedError(f"unsupported optimizer_config entries: {sorted(cfg)}")
        value_loss = g('value_loss', 'mse')
        if value_loss not in hip.VALUE_LOSS_KINDS:
            raise AssertionError(f"Value loss name {value_loss} does not match any implemented loss functions "
                                 f"({list(hip.VALUE_LOSS_KINDS)})")
        vcfg = g('value_loss_config', {})
        self._hp = hip.PpoHparams(eps_clip=self.eps_clip, c_clip=float(self.c_clip), value_eps_clip=self.value_eps_clip,
                                  value_loss_weight=self.value_loss_weight,
                                  entropy_bonus_weight=self.entropy_bonus_weight,
                                  huber_delta=float(vcfg.get("delta", vcfg.get("beta", 1.0))), norm_eps=1e-5,
                                  dual_clip=int(bool(self.dual_clip)), clip_value=int(bool(self.clip_value)),
                                  value_loss=hip.VALUE_LOSS_KINDS[value_loss], mask_invert=1)

        net = policy.net
        # optimiser state on the flat layout: Adam (exp_avg, exp_avg_sq); RMSprop (momentum_buffer, square_avg, grad_avg);
        # SGD (momentum_buffer)
        self._m = torch.zeros_like(net.flat)
        self._v = torch.zeros_like(net.flat) if name != 'sgd' else None
        self._gavg = torch.zeros_like(net.flat) if name == 'rmsprop' and self._centered else None
        self._opt_steps = 0
        self.frames = 0
        # rows (env-steps) per forward/backward chunk: bounds the activation workspace, not the arithmetic.  Default: 16 384 for
        # nets with a convolution / pooling encoder (26 KB of taped activations per frame); 2^20 for vector-observation nets,
        # whose tape is a few hundred bytes per row (1 GB at that size) and whose fused chains (csrc/mlp_mfma.h) pay their
        # per-workgroup parameter staging and gradient fold once per launch (the C1-shaped update at 4096 x 128: 3.08 ms in four
        # chunks of 131 072 on four pipelines, 2.69 ms in one)
        convolutional = any(not isinstance(L, (ns.LinearSpec, ns.LayerNormSpec)) for enc in
                            list(net.spec.obs_encoders) + list(net.spec.state_encoders or []) for L in enc.layers)
        self.chunk_rows = int(g("chunk_rows", 16384 if convolutional else 1 << 20))
        self._world = 1
        self._dist = False
        # capture the device part of a step into a hipGraph per sample signature and replay it (launch-bound small
        # configurations; needs device-resident or equal-shaped samples; not with a process group)
        self.grad_bucket_bytes = int(g("grad_bucket_bytes", 1 << 20))
        self.use_graph = bool(g("use_graph", False))
        self._graphs = {}
        self._comm = None
        self._reducer = None
        # Row-chunk pipelines side by side on streams of their own (default 4; pipelines=1 / SRL_PIPELINES=1: one; round 4, same
        # box: 1 -> 112.5, 2 -> 111.0-112.4, 3 -> 110.4, 4 -> 108.7-109.2, 6 -> 112.5 ms, the runtime has 8 hardware queues):
        # the chunks of a batch are
        # independent up to the gradient sum, and their kernels -- each bound by something else -- fill each other's
        # stalls: 147.4 -> 141.3 ms per update.  (The first version was not bit-reproducible: beside a concurrent queue the
        # narrow head product returned different sums for a few hundred rows; traced to the packed-float32 code the
        # compiler made of that kernel's accumulators, skinny.h -- with scalar accumulators every buffer of a step is
        # bit-identical to the one-pipeline run)
        self.pipelines = int(g("pipelines", os.environ.get("SRL_PIPELINES", "4")))
        self._twin = None
        self._pipe_stream = None
        self._gae_ws = {}

    # ------------------------------------------------------------------ checkpoints (mappo.py:58-66)
    def get_checkpoint(self):
        ckpt = self.policy.get_checkpoint()
        net = self.policy.net
        names = net.ref_names()
        named = lambda t: net.flat_to_reference(t.detach().cpu())
        step_t = torch.tensor(float(self._opt_steps))
        state = {}
        # the PopArt statistics are gradient-less nn.Parameters of the reference: listed, stateless
        params = list(range(len(names) + (3 if net.spec.popart else 0)))
        common = dict(maximize=False, foreach=None, differentiable=False, params=params)
        if self._opt in ('adam', 'adamw'):
            if self._opt_steps > 0:
                m, v = named(self._m), named(self._v)
                state = {i: dict(step=step_t.clone(), exp_avg=m[n], exp_avg_sq=v[n]) for i, n in enumerate(names)}
            group = dict(lr=self._lr, betas=self._betas, eps=self._eps, weight_decay=self._weight_decay, amsgrad=False,
                         capturable=False, fused=None, **common)
        elif self._opt == 'rmsprop':
            if self._opt_steps > 0:
                sq, buf = named(self._v), named(self._m)
                ga = named(self._gavg) if self._centered else None
                for i, n in enumerate(names):
                    state[i] = dict(step=step_t.clone(), square_avg=sq[n])
                    if self._momentum > 0:
                        state[i]["momentum_buffer"] = buf[n]
                    if self._centered:
                        state[i]["grad_avg"] = ga[n]
            group = dict(lr=self._lr, momentum=self._momentum, alpha=self._alpha, eps=self._eps, centered=self._centered,
                         weight_decay=self._weight_decay, capturable=False, **common)
        else:
            if self._opt_steps > 0 and self._momentum != 0:
                buf = named(self._m)
                state = {i: dict(momentum_buffer=buf[n]) for i, n in enumerate(names)}
            group = dict(lr=self._lr, momentum=self._momentum, dampening=self._dampening, weight_decay=self._weight_decay,
                         nesterov=self._nesterov, fused=None, **common)
        ckpt.update({"optimizer_state_dict": {"state": state, "param_groups": [group]}})
        return ckpt

    def load_checkpoint(self, checkpoint, **kwargs):
        osd = checkpoint.get("optimizer_state_dict")
        if osd is not None:
            net = self.policy.net
            names = net.ref_names()
            st = osd["state"]
            flat_of = lambda key: net.reference_to_flat({n: st[i][key] for i, n in enumerate(names)})
            slots = {'adam': (("exp_avg", "_m"), ("exp_avg_sq", "_v")), 'adamw': (("exp_avg", "_m"), ("exp_avg_sq", "_v")),
                     'rmsprop': (("momentum_buffer", "_m"), ("square_avg", "_v"), ("grad_avg", "_gavg")),
                     'sgd': (("momentum_buffer", "_m"),)}[self._opt]
            for key, attr in slots:
                buf = getattr(self, attr)
                if buf is None:
                    continue
                if st and key in st[0] and st[0][key] is not None:
                    buf.copy_(flat_of(key))
                else:
                    buf.zero_()
            if st:
                # SGD keeps no step count: a restored momentum buffer means "not the first step"
                self._opt_steps = int(float(st[0]["step"])) if "step" in st[0] else 1
            else:
                self._opt_steps = 0
            grp = osd["param_groups"][0]
            # the group names its optimiser by the keys only that optimiser has (torch's load_state_dict would overwrite
            # the group and fail on foreign state; mappo.py:58-66): a checkpoint of another kind is refused, not zero-filled
            kind_keys = {'adam': ("betas",), 'adamw': ("betas",), 'rmsprop': ("alpha", "centered"), 'sgd': ("dampening", "nesterov")}
            foreign = [k for kind, keys in kind_keys.items() if kind_keys[kind] != kind_keys[self._opt] for k in keys if k in grp]
            if foreign or any(k not in grp for k in kind_keys[self._opt]):
                raise ValueError(f"optimizer_state_dict is not a `{self._opt}` state (group keys {sorted(grp)})")
            if self._opt == 'rmsprop' and bool(grp["centered"]) != self._centered:
                self._centered = bool(grp["centered"])
                self._gavg = torch.zeros_like(net.flat) if self._centered else None
                if self._centered and st and "grad_avg" in st[0]:
                    self._gavg.copy_(flat_of("grad_avg"))
            if self._opt == 'sgd':
                self._dampening, self._nesterov = grp.get("dampening", self._dampening), bool(grp.get("nesterov", self._nesterov))
            self._lr = grp["lr"]
            self._weight_decay = grp.get("weight_decay", self._weight_decay)
            if self._opt in ('adam', 'adamw'):
                self._betas, self._eps = tuple(grp["betas"]), grp["eps"]
            elif self._opt == 'rmsprop':
                self._alpha, self._eps = grp.get("alpha", self._alpha), grp.get("eps", self._eps)
                self._momentum = grp.get("momentum", self._momentum)
            else:
                self._momentum = grp.get("momentum", self._momentum)
        self.policy.load_checkpoint(checkpoint)

    def distributed(self, rank=None, world_size=None, init_method=None, **kwargs):
        super().distributed(rank=rank, world_size=world_size, init_method=init_method, **kwargs)
        self._world = dist.get_world_size() if dist.is_initialized() else 1
        self._dist = dist.is_initialized()  # collectives run whenever a group exists (also with one rank)
        self._comm = None
        if self._dist and self.policy.device != "cpu" and dist.get_backend() == "nccl":
            from srl_amd import comm
            self._comm = comm.NativeComm.from_process_group(self.policy.device)  # None: torch.distributed collectives
        self._reducer = _BucketReducer(self.policy.net, self.grad_bucket_bytes, self._comm) if self._dist else None

    def _importance_ratio(self, net, obs, avail, action, old_lp, rows, B, alive=None, pstate=None, on_reset=None):
        """exp(new_lp - old_lp) on rows [0, rows) of the sample, forward only; [rows, B, 1] float32.  Feed-forward nets go
        through in row chunks; a recurrent net walks the time axis in one piece, chunked from the stored states exactly
        like the training pass (mappo.py:243-246 analyses the same rows for the ratio and for the loss)."""
        n_all = rows * B
        flat = lambda t: t[:rows].reshape(n_all, *t.shape[2:])
        f_obs = {k: flat(v) for k, v in obs.items()}
        f_avail = None if avail is None else flat(avail)
        f_action = flat(action)
        new_lp = torch.empty(n_all, dtype=torch.float32, device=old_lp.device)
        rnn = None
        if net.spec.num_rnn_layers:
            rnn = self.policy._rnn_ctx_with_burn_in(obs, None, pstate, on_reset, 0, rows, B)
        step = n_all if rnn is not None else self.chunk_rows
        for r0 in range(0, n_all, step):
            r1 = min(n_all, r0 + step)
            n = r1 - r0
            logits, _ = net.forward({k: v[r0:r1] for k, v in f_obs.items()}, n, keep_tape=False, rnn=rnn)
            ent = net.ws.get("entropy", n)[:n]
            self.policy.dist_fwd(logits, f_action[r0:r1], None if f_avail is None else f_avail[r0:r1], new_lp[r0:r1], ent)
            self.policy.mask_dead(new_lp[r0:r1], None if alive is None else flat(alive).reshape(-1)[r0:r1])
        ratio = torch.empty(n_all, dtype=torch.float32, device=old_lp.device)
        hip.importance_ratio(new_lp, flat(old_lp).reshape(-1).contiguous(), ratio)
        return ratio.reshape(rows, B, 1)

    # ------------------------------------------------------------------ the step (mappo.py:219-328)
    def step(self, sample):
        hip.require_gpu()
        dev = self.policy.device
        net = self.policy.net
        if sample.truncated is None:
            sample.truncated = (torch.zeros_like(sample.done) if isinstance(sample.done, torch.Tensor) else
                                np.zeros_like(sample.done))  # mappo.py:222-223
        if self.recompute_adv_on_reuse:
            sample.analyzed_result.adv = sample.analyzed_result.ret = None  # :224-225

        # ---- the sample's leaves in their wire dtypes: on the device -- or, for a captured step (use_graph), wherever they
        # are: the native step driver copies them straight into the graph's static inputs ---------------------------------
        graphed = (self.use_graph and not self._dist and self._opt in ('adam', 'adamw')
                   and not any(isinstance(v, RingObs) for v in sample.obs.values()))  # ring rows are bound per sample
        leaf = (lambda x, kind: wire_leaf(x, kind)) if graphed else (lambda x, kind: to_device_leaf(x, dev, kind))
        L = dict(on_reset=leaf(sample.on_reset, "flag"), done=leaf(sample.done, "flag"),
                 truncated=leaf(sample.truncated, "flag"), reward=leaf(sample.reward, "real"),
                 old_value=leaf(sample.analyzed_result.value, "real"), old_lp=leaf(sample.analyzed_result.log_probs, "real"),
                 action=leaf(sample.action.x, self.policy.action_kind()))
        for k, v in sample.obs.items():
            if v is not None:
                L[f"obs.{k}"] = leaf(v, "obs")
        have_adv = sample.analyzed_result.adv is not None
        if have_adv:
            L["adv"] = leaf(sample.analyzed_result.adv, "real")
            L["ret"] = leaf(sample.analyzed_result.ret, "real")
        if net.spec.num_rnn_layers:
            if sample.policy_state is None:
                raise ValueError("recurrent policy: the sample carries no policy_state")
            for k, v in sample.policy_state.items():
                L[f"policy_state.{k}"] = leaf(v, "real")

        # shared multi-agent samples carry [Tb, B, agents, ...] leaves: every operation of the step is per (env, agent)
        # column, so the agents are folded into the batch axis (a view) and unfolded where results go back to the sample
        agents = L["on_reset"].shape[2] if len(L["on_reset"].shape) == 4 else 0
        if agents:
            L = {k: v.reshape(v.shape[0], v.shape[1] * v.shape[2], *v.shape[3:]) for k, v in L.items()}
        Tb, B = L["on_reset"].shape[0], L["on_reset"].shape[1]
        boot, burn = self.bootstrap_steps, self.burn_in_steps
        lo, hi = burn, Tb - boot  # valid rows (mappo.py:259)
        n_valid = (hi - lo) * B
        Nc = L["old_value"].shape[2] if len(L["old_value"].shape) > 2 else 1
        if Nc != net.spec.value_dim:
            raise ValueError(f"sample values have {Nc} channels, the policy's value head {net.spec.value_dim}")
        if Nc > 1 and len(L["reward"].shape) > 2 and L["reward"].shape[2] == 1:  # one reward for every channel (broadcast, :131)
            r = L["reward"]
            L["reward"] = (np.ascontiguousarray(np.broadcast_to(r, (*r.shape[:2], Nc))) if isinstance(r, np.ndarray) else
                           r.expand(*r.shape[:2], Nc).contiguous())
        if not have_adv and boot == 0:
            raise ValueError("bootstrap_steps == 0 requires advantages computed before the trainer")

        # ---- the device part: everything between "leaves in HBM" and "terms ready"; no host synchronisation inside, so
        # it can be captured once into a hipGraph and replayed (use_graph) -----------------------------------------
        if graphed:
            out = self._replay(L, have_adv, self._step_scalars(self.ppo_epochs), host_results=not isinstance(sample.reward, torch.Tensor))
        else:
            out = self._device_part(L, have_adv, None)
        self._opt_steps += self.ppo_epochs
        if self.popart:
            self.policy._popart_updates += self.ppo_epochs

        # ---- statistics: the only device->host synchronisation of the step (the reference syncs ~11 times per epoch) ------
        raw = out["terms"]  # [epochs, nchunks * LT_COUNT | gradient norm (float32) | 3 PopArt sums]
        if isinstance(raw, torch.Tensor):
            raw = raw.cpu().numpy()
        nch = out["nchunks"]
        train_stats = defaultdict(float)
        for erow in raw:
            row = erow[:nch * hip.LT_COUNT].reshape(nch, hip.LT_COUNT).sum(0)
            tail = erow[nch * hip.LT_COUNT:]
            msum = max(row[hip.LT_MASK], 1e-30)
            if self.popart:  # PPOStepResult.denorm_value: masked mean of the de-normalised targets (:215), all channels
                ps = tail[1:1 + 3 * Nc].reshape(Nc, 3)
                train_stats["denorm_value"] += ps[:, 1].sum() / max(ps[:, 0].sum(), 1e-30)
            for key, slot in _STAT_TERMS:  # masked_select broadcasts the mask over the value channels (:205-216)
                train_stats[key] += row[slot] / (msum * Nc if slot in (hip.LT_CLIP, hip.LT_ADV, hip.LT_RET) else msum)
            train_stats["done"] += row[hip.LT_DONE] / max(n_valid, 1)
            train_stats["truncated"] += row[hip.LT_TRUNC] / max(n_valid, 1)
            train_stats["grad_norm"] += float(tail[:1].view(np.float32)[0])
        for k in train_stats:
            train_stats[k] /= self.ppo_epochs

        # advantages / returns go back into the numpy sample so that a re-used buffer entry carries them (:254-257)
        adv_d, ret_d = out["adv"], out["ret"]
        if agents:
            adv_d, ret_d = (t.reshape(Tb, B // agents, agents, *t.shape[2:]) for t in (adv_d, ret_d))
        if not have_adv and out.get("owned"):  # the step driver already delivered fresh copies where the sample lives
            sample.analyzed_result.adv, sample.analyzed_result.ret = adv_d, ret_d
        elif not have_adv and not isinstance(sample.reward, torch.Tensor):
            sample.analyzed_result.adv = adv_d.cpu().numpy()
            sample.analyzed_result.ret = ret_d.cpu().numpy()
        elif not have_adv:
            sample.analyzed_result.adv, sample.analyzed_result.ret = adv_d.clone(), ret_d.clone()
        if self.recompute_adv_among_epochs:
            sample.analyzed_result.adv = sample.analyzed_result.ret = None

        self.policy.inc_version()  # once per step, not per epoch (:305-307)
        if self.entropy_decay_per_steps and self.policy.version % self.entropy_decay_per_steps == 0:
            self.entropy_bonus_weight *= self.entropy_bonus_decay
            self._hp.entropy_bonus_weight = self.entropy_bonus_weight
            self._graphs.clear()  # the coefficient is baked into captured launches

        self.frames += n_valid
        info = {}
        if sample.info_mask is not None and sample.info is not None:
            im = sample.info_mask[lo:hi]
            im = im.cpu().numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
            elapsed = im.sum()
            if elapsed > 0:  # episode statistics of a device-resident sample (ingest ring) come to the host here
                host_info = recursive_apply(sample.info[lo:hi],
                                            lambda x: x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x))
                info = recursive_apply(host_info * im, lambda x: x.sum()) / elapsed
                info = {k: float(v) for k, v in info.items()}
        stats = dict(frames=int(self.frames), **{k: float(v) for k, v in train_stats.items()}, **info)
        return TrainerStepResult(stats=stats, step=self.policy.version)

    # ------------------------------------------------------------------ hipGraph capture of the device part
    def _step_scalars(self, epochs):
        """Adam bias corrections of the next `epochs` optimiser steps, float32 [epochs, 2] (what srl_adam_step derives
        from its `step` argument, precomputed so that the captured launch carries no per-step scalar)."""
        rows = []
        for e in range(1, epochs + 1):
            t = self._opt_steps + e
            rows.append([self._lr / (1.0 - self._betas[0]**t), (1.0 - self._betas[1]**t)**0.5])
        return np.asarray(rows, dtype=np.float32)

    def _replay(self, L, have_adv, scal, host_results):
        """The device part as a captured hipGraph behind the native step driver (csrc/step_plan.hip): per step ONE C call
        copies the sample's leaves (host or device) into the graph's static inputs, launches the graph and brings the
        loss terms -- and fresh copies of the advantages / value targets, on the host or on the device as the sample is --
        back.  First sight of a sample signature runs eagerly (loads every kernel, sizes the workspaces), the second is
        captured, later ones replay."""
        dev = self.policy.device
        keys = sorted(L)
        key = (have_adv,) + tuple((k, tuple(L[k].shape), str(L[k].dtype).replace("torch.", "")) for k in keys)
        if key not in self._graphs:
            self._graphs[key] = None
            return self._device_part({k: to_device_leaf(v, dev, "as-is") for k, v in L.items()}, have_adv, None)
        ent = self._graphs[key]
        if ent is None:
            static = {k: to_device_leaf(L[k], dev, "as-is").clone() for k in keys}
            dscal = torch.from_numpy(scal).to(dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                out = self._device_part(static, have_adv, dscal)
            plan = hip.StepPlan(graph.raw_cuda_graph_exec())
            for k in keys:
                plan.add_input(static[k])
            plan.add_input(dscal)
            terms_host = torch.empty(out["terms"].shape, dtype=out["terms"].dtype).pin_memory()
            plan.add_output(out["terms"], terms_host)
            plan.add_output(out["adv"], None)
            plan.add_output(out["ret"], None)
            ent = (graph, plan, static, dscal, out, terms_host)
            self._graphs[key] = ent
        graph, plan, static, dscal, out, terms_host = ent
        ptr = lambda v: v.data_ptr() if isinstance(v, torch.Tensor) else v.ctypes.data
        adv = ret = None
        if not have_adv:  # fresh copies for the sample (mappo.py:254-257): numpy for a host sample, device tensors otherwise
            shape = tuple(out["adv"].shape)
            if host_results:
                adv, ret = np.empty(shape, np.float32), np.empty(shape, np.float32)
            else:
                adv, ret = torch.empty(shape, dtype=torch.float32, device=dev), torch.empty(shape, dtype=torch.float32, device=dev)
        plan.run([ptr(L[k]) for k in keys] + [scal.ctypes.data], [None, None if adv is None else ptr(adv),
                                                                 None if ret is None else ptr(ret)], sync=True)
        return dict(terms=terms_host.numpy(), nchunks=out["nchunks"], adv=adv if adv is not None else out["adv"],
                    ret=ret if ret is not None else out["ret"], owned=adv is not None)

    def _device_part(self, L, have_adv, dscal):
        """GAE / statistics / (PopArt) / epochs x (forward, loss, backward, [all-reduce], clip + Adam).  Returns device
        tensors only: per-epoch loss terms, and the padded advantages / value targets."""
        dev = self.policy.device
        net = self.policy.net
        on_reset, done, truncated, reward = L["on_reset"], L["done"], L["truncated"], L["reward"]
        old_value, old_lp, action = L["old_value"], L["old_lp"], L["action"]
        obs = {k[4:]: v for k, v in L.items() if k.startswith("obs.")}
        avail = obs.pop("available_action", None)
        alive = obs.pop("is_alive", None) if self.policy.masks_dead_agents else None
        pstate = {k[len("policy_state."):]: v for k, v in L.items() if k.startswith("policy_state.")}
        Tb, B = on_reset.shape[0], on_reset.shape[1]
        boot, burn = self.bootstrap_steps, self.burn_in_steps
        lo, hi = burn, Tb - boot
        n_valid = (hi - lo) * B
        Nc = net.spec.value_dim
        f64 = dict(dtype=torch.float64, device=dev)
        # step-persistent device state lives in the net's workspace: nothing below allocates in steady state
        stats_local = net.ws.get("mappo.stats_local", 3, torch.float64)[:3]
        # without a process group the global statistics ARE the local ones (no copy)
        stats_global = net.ws.get("mappo.stats_global", 3, torch.float64)[:3] if self._dist else stats_local
        stats_work = None  # pending all-reduce of stats_global (joined right before its first reader)
        adv_d = ret_d = None
        # what goes back to the host, in ONE float64 block zeroed once per step: per epoch the loss-term sums of every
        # row chunk (added up on the host), the gradient norm (a float32 in the low half of its slot) and, with PopArt,
        # the masked sums of the value targets
        n_valid_rows = (Tb - self.bootstrap_steps - self.burn_in_steps) * B
        chunk_rows = n_valid_rows if net.spec.num_rnn_layers else self.chunk_rows
        nchunks = max(1, -(-n_valid_rows // chunk_rows))
        stride = nchunks * hip.LT_COUNT + 1 + 3 * Nc
        block = net.ws.get("mappo.out", self.ppo_epochs * stride, torch.float64)[:self.ppo_epochs * stride]
        block.zero_()
        block = block.view(self.ppo_epochs, stride)

        for epoch in range(self.ppo_epochs):
            # what the encoder blocks derive from the parameters alone goes first: it runs while the host issues the head below
            if dscal is None:
                net.prepare_derived()
            # ---- advantages / value targets ------------------------------------------------------------------
            if adv_d is None:
                if have_adv:
                    adv_d, ret_d = L["adv"], L["ret"]
                    fused_stats = False
                else:
                    adv_d = net.ws.get("mappo.adv", Tb * B * Nc)[:Tb * B * Nc].view(Tb, B, Nc)
                    ret_d = net.ws.get("mappo.ret", Tb * B * Nc)[:Tb * B * Nc].view(Tb, B, Nc)
                    # the scan writes rows [0, Tb-1); the last row is the zero pad (:254-256).  Zeroed every step: the
                    # workspace is shared by every sample shape this trainer sees (and by every captured graph), so the
                    # row may hold another shape's advantages -- B * Nc floats, nothing next to the step
                    adv_d[Tb - 1:].zero_()
                    ret_d[Tb - 1:].zero_()
                    fused_stats = boot == 1 and burn == 0  # the scan's own sums are those of the loss rows
                    gws = self._gae_ws.get((B, Nc))
                    if gws is None:  # zeroed once; afterwards the scan's own last workgroup resets it
                        gws = self._gae_ws[(B, Nc)] = hip.gae_scan_workspace(B, Nc, dev)
                    # PopArt: the stored values are normalised; the trace runs on de-normalised ones (mappo.py:120-124)
                    trace_value = self.policy.denormalize_value(old_value) if self.popart else old_value
                    ratio = None
                    if self.vtrace:  # importance ratio of the CURRENT parameters on every rewarding step (:130-133)
                        ratio = self._importance_ratio(net, obs, avail, action, old_lp, Tb - 1, B, alive, pstate, on_reset)
                    hip.gae_scan(reward, trace_value, done, truncated, on_reset, self.discount_rate, self.gae_lambda,
                                 adv_d, ret_d, stats=stats_local if fused_stats else None, imp_ratio=ratio,
                                 workspace=gws if fused_stats else None)
                mask_rows = on_reset[1 + lo:1 + hi]  # loss_mask = 1 - on_reset[1+burn : 1+Tb-boot]  (:260-261)
                if not fused_stats and Nc == 1:
                    hip.masked_stats(adv_d[lo:hi], mask_rows, stats_local, mask_invert=True)
                elif not fused_stats:  # [n, Nc] advantages under an [n] mask: per-channel sums, the mask counted once
                    cs = net.ws.get("mappo.col_stats", 3 * Nc, torch.float64)[:3 * Nc].view(Nc, 3)  # zeroed by the call
                    hip.masked_stats_cols(adv_d[lo:hi].reshape(-1, Nc), mask_rows, cs, Nc, mask_invert=True)
                    hip.fold_col_stats(cs, Nc, stats_local)
                if self._dist:
                    stats_global.copy_(stats_local)
                    # one 24-byte message instead of three (utils.py:58-61), issued asynchronously: it crosses the
                    # links while the first chunk's forward pass runs and is joined before the first loss kernel
                    if self._comm is not None:
                        self._comm.all_reduce_f64_async(stats_global)
                        stats_work = self._comm
                    else:
                        stats_work = dist.all_reduce(stats_global, async_op=True)
                local_n = stats_local[0:1]

            flat = lambda t: t[lo:hi].reshape(n_valid, *t.shape[2:])
            # ---- PopArt: statistics of the value targets, then the loss sees normalised targets (:263-264, :173-176) ----
            loss_ret, pstats_local = ret_d, None
            if self.popart:
                pstats_local = block[epoch, stride - 3 * Nc:].view(Nc, 3)  # zeroed by srl_masked_stats_cols itself
                hip.masked_stats_cols(flat(ret_d), on_reset[1 + lo:1 + hi], pstats_local, Nc, mask_invert=True)
                pstats = net.ws.get("mappo.pstats", 3 * Nc, torch.float64)[:3 * Nc].view(Nc, 3)
                pstats.copy_(pstats_local)
                if self._dist and self._comm is not None:
                    # one message instead of utils.py:121-124's three, on the SAME communicator and side stream as the
                    # advantage statistics and the gradient buckets: one communicator orders every collective of a step
                    self._comm.all_reduce_f64_async(pstats)
                    self._comm.join()
                elif self._dist:
                    if stats_work is not None:  # never two collectives of this step in flight on different streams
                        stats_work.wait()
                        stats_work = None
                    dist.all_reduce(pstats)
                self.policy.update_popart_from_stats(pstats, count=False)
                loss_ret = torch.empty_like(ret_d)
                hip.popart_map(ret_d, net.popart_state, Nc, loss_ret, True, ns.POPART_EPS)
            loss_oldv = self.policy.normalize_value(old_value) if self.normalize_old_value else old_value  # :151-152

            # ---- forward / loss / backward over row chunks -----------------------------------------------------------
            net.zero_grad()
            f_obs = {k: flat(v) for k, v in obs.items()}
            f_avail = None if avail is None else flat(avail)
            f_alive = None if alive is None else flat(alive).reshape(-1)
            f_action, f_oldlp, f_oldv = flat(action), flat(old_lp).reshape(-1), flat(loss_oldv).reshape(-1)
            f_adv, f_ret, f_mask = flat(adv_d).reshape(-1), flat(loss_ret).reshape(-1), mask_rows.reshape(-1)
            f_done, f_trunc = flat(done).reshape(-1), flat(truncated).reshape(-1)
            # recurrent nets walk the time axis: all valid rows go through in one piece
            rnn = None
            if net.spec.num_rnn_layers:  # chunk states from the sample, or from a no-grad replay of the burn-in rows
                rnn = self.policy._rnn_ctx_with_burn_in(obs, None, pstate, on_reset, burn, hi - lo, B)
            terms = block[epoch, :nchunks * hip.LT_COUNT].view(nchunks, hip.LT_COUNT)
            reducer = self._reducer if self._dist else None
            # Two pipelines: even chunks on the compute stream with the policy's executor, odd chunks on a second stream
            # with its twin (same parameters, own workspace / tape / gradient buffer).  Not for recurrent nets (one chunk),
            # not inside a graph capture.
            two = (self.pipelines >= 2 and nchunks >= 2 and rnn is None and dscal is None and not net.force_explicit_conv)
            nets, streams = [net], [torch.cuda.current_stream()]
            if two:
                npipe = min(self.pipelines, nchunks)
                if self._twin is None or len(self._twin) != npipe - 1:
                    self._twin = [net.twin() for _ in range(npipe - 1)]
                    self._pipe_stream = [torch.cuda.Stream(device=dev) for _ in range(npipe - 1)]
                # the weight-range slots are shared: they are recomputed here, before the fork.  Before the very first
                # forward pass the layers that want one are not known yet: the other pipelines then start after the first
                # chunk, which discovers and fills them
                first_sync = not net._wamax_slot
                net.refresh_weight_ranges()
                if stats_work is not None:  # the global advantage statistics: needed by every pipeline
                    stats_work.join() if stats_work is self._comm else stats_work.wait()
                    stats_work = None
                for twin, pst in zip(self._twin, self._pipe_stream):
                    twin.flat, twin.popart_state = net.flat, net.popart_state  # (re-bound by a checkpoint load)
                    twin.zero_grad()
                    pst.wait_stream(streams[0])
                    nets.append(twin)
                    streams.append(pst)
            if reducer is not None:
                reducer.begin([x.grad for x in nets])
            net.chunks_of_one_update(True)
            net._multi[0] = len(nets) > 1
            for x in nets:
                x.reset_open_accumulations()  # (an accumulation a failed update left open is not continued)
            for ci in range(nchunks):
                r0, r1 = ci * chunk_rows, min(n_valid, (ci + 1) * chunk_rows)
                n = r1 - r0
                e = ci % len(nets)
                cnet = nets[e]
                cnet.last_chunk = ci + len(nets) >= nchunks  # this executor's last chunk of the update
                if reducer is not None and ci + len(nets) >= nchunks:
                    # a pipeline's gradients become final in the backward pass of ITS last chunk; a bucket goes out when
                    # every pipeline has released it (api/policy.py:219-238: what DDP's bucketing does)
                    cnet.grad_ready_hook = reducer.hook(e)
                with torch.cuda.stream(streams[e]):
                    c_obs = {k: v[r0:r1] for k, v in f_obs.items()}
                    c_avail = None if f_avail is None else f_avail[r0:r1]
                    logits, value = cnet.forward(c_obs, n, keep_tape=True, rnn=rnn)
                    logp = cnet.ws.get("new_logp", n)[:n]
                    ent = cnet.ws.get("entropy", n)[:n]
                    self.policy.dist_fwd(logits, f_action[r0:r1], c_avail, logp, ent, net=cnet)
                    self.policy.mask_dead(logp, None if f_alive is None else f_alive[r0:r1])
                    d_lp = cnet.ws.get("d_logp", n)[:n]
                    d_v = cnet.ws.get("d_value", n * Nc)[:n * Nc]
                    d_ent = cnet.ws.get("d_entropy", n)[:n]
                    if stats_work is not None:  # the global advantage statistics: first needed here
                        stats_work.join() if stats_work is self._comm else stats_work.wait()
                        stats_work = None
                    hip.ppo_loss_fwd_bwd(logp, f_oldlp[r0:r1], value.reshape(-1), f_oldv[r0 * Nc:r1 * Nc], f_adv[r0 * Nc:r1 * Nc],
                                         f_ret[r0 * Nc:r1 * Nc],
                                         ent, f_mask[r0:r1], self._hp, stats_global, local_n, d_lp, d_v, d_ent, terms[ci],
                                         done=f_done[r0:r1], truncated=f_trunc[r0:r1], value_dim=Nc)
                    d_logits = cnet.ws.get("d_logits", logits.numel())[:logits.numel()].view_as(logits)
                    d_ls = self.policy.dist_bwd(logits, f_action[r0:r1], c_avail, d_lp, d_ent, d_logits, net=cnet)
                    cnet.backward(d_logits, d_v.view(n, Nc), d_ls)
                if two and ci == 0 and first_sync:
                    for pst in streams[1:]:
                        pst.wait_stream(streams[0])
            net.chunks_of_one_update(False)
            net._multi[0] = False
            for x in nets:
                x.last_chunk = None
            if two and reducer is None:
                for pst in self._pipe_stream:
                    streams[0].wait_stream(pst)
                hip.accumulate_n(net.grad, [twin.grad for twin in self._twin])   # one launch, added in pipeline order

            # ---- gradient reduction, clip, Adam (mappo.py:272-284) ---------------------------------------------------
            if reducer is not None:  # the buckets not yet launched (folding the pipelines' slices), then wait for all of them
                for x in nets:
                    x.grad_ready_hook = None
                reducer.finish(streams[1:])
                for pst in streams[1:]:
                    streams[0].wait_stream(pst)
            sumsq = net.ws.get("mappo.sumsq", 1, torch.float64)[:1]  # zeroed by srl_grad_sumsq
            slot = epoch * stride + nchunks * hip.LT_COUNT
            gnorm = block.view(-1).view(torch.float32)[2 * slot:2 * slot + 1]
            hip.grad_sumsq(net.grad, sumsq)
            clip = dict(grad_scale=1.0 / self._world, sumsq=sumsq, grad_norm_out=gnorm,
                        max_norm=-1.0 if self.max_grad_norm is None else float(self.max_grad_norm))
            if self._opt in ('adam', 'adamw'):
                hip.adam_step(net.flat, net.grad, self._m, self._v, self._lr, self._betas[0], self._betas[1], self._eps,
                              self._weight_decay, self._adamw, self._opt_steps + epoch + 1,
                              step_scalars=None if dscal is None else dscal[epoch], **clip)
            elif self._opt == 'rmsprop':
                hip.rmsprop_step(net.flat, net.grad, self._v, self._m if self._momentum > 0 else None, self._gavg, self._lr,
                                 self._alpha, self._eps, self._weight_decay, self._momentum, self._centered, **clip)
            else:
                hip.sgd_step(net.flat, net.grad, self._m if self._momentum != 0 else None, self._lr, self._momentum,
                             self._dampening, self._weight_decay, self._nesterov, self._opt_steps + epoch == 0, **clip)

            net.params_changed()  # cached per-layer weight ranges (two-plane f16 forward products) are stale
            if self.recompute_adv_among_epochs and epoch + 1 < self.ppo_epochs:
                adv_d = ret_d = None
                have_adv = False

        return dict(terms=block, nchunks=nchunks, adv=adv_d, ret=ret_d)


register('mappo', MultiAgentPPO)
register('mappo-hip', MultiAgentPPO)
