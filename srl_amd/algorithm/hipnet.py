"""Device executor of the actor-critic network: forward and hand-written backward on the HIP kernels.

No autograd and no torch compute ops: PyTorch only allocates the buffers.  Every layer is one or two
launches through the C ABI (``srl_amd.hip``):

================  =========================================  ==========================================
layer             forward                                    backward (accumulates parameter gradients)
================  =========================================  ==========================================
Linear (+act)     ``srl_gemm`` X W^T + b, act in epilogue    ``srl_gemm`` dZ^T X (split-K), ``srl_colsum``,
                                                             ``srl_gemm`` dZ W with act'(X) in the epilogue
LayerNorm         ``srl_layernorm_fwd``                      ``srl_layernorm_bwd`` (act' of the producer fused)
first Conv2d      ``srl_obs_ln_stats`` + ``srl_im2col_obs_ln``  GEMMs + ``srl_obs_ln_affine_bwd``
                  (uint8 -> LN -> patches) + ``srl_gemm``
later Conv2d      ``srl_im2col_nhwc`` + ``srl_gemm``             GEMMs + ``srl_col2im_nhwc`` (act' fused)
================  =========================================  ==========================================

The gradient handed to a layer's ``backward`` is always w.r.t. its *pre-activation* output: whoever
produces a gradient w.r.t. an activation output multiplies by act'(output) in its own epilogue, so the
mask never costs a separate pass.  Activations between convolutions are NHWC, so a convolution's GEMM
output ``[n*OH*OW, Cout]`` is the next layer's input without a transpose (parameter layouts are adapted
once, at the checkpoint boundary: ``netspec.ParamInfo``).
"""
import math
import os
from collections import OrderedDict
from typing import Dict, NamedTuple, Optional

import torch

from srl_amd import hip
from srl_amd.algorithm import netspec as ns
from srl_amd.runtime.obs_ring import RingObs


class Buf(NamedTuple):
    ptr: int
    ld: int
    rows: int
    cols: int
    # device pointer of the sign bits of a ReLU output (bit e of the word array = element e of the buffer is > 0), written
    # by the producing convolution: all that a data gradient needs of the activation (srl_hip.h: y_mask / x_mask / dact_mask)
    mask: Optional[int] = None


class RnnCtx(NamedTuple):
    """How the n = T*B time-major rows of a forward pass are laid out for the recurrent layers: chunks of C time
    steps (T % C == 0), N = (T/C)*B environment columns per step; h0[tag] float32 [layers, N, H] initial states
    ("a:" actor / shared backbone, "c:" critic backbone); reset uint8 [C, N] chunk-major on_reset flags, or None
    (rollout: the state is used as given, actor_critic_policy.py:476-481)."""
    T: int
    B: int
    C: int
    h0: dict
    reset: Optional[torch.Tensor]


class Workspace:
    """Named device buffers that only ever grow (steady-state steps allocate nothing)."""

    def __init__(self, device):
        self.device = device
        self._bufs: Dict[str, torch.Tensor] = {}
        self._retired = []

    def get(self, name: str, numel: int, dtype=torch.float32) -> torch.Tensor:
        t = self._bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            if t is not None:
                # a kernel on the second stream may still be reading the buffer that is being outgrown: the allocator
                # would hand its memory to the next request of the compute stream at once.  Kept until `release_retired`.
                self._retired.append(t)
            t = torch.empty(max(int(numel), 1), dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t

    def release_retired(self):
        """Called once the compute stream has waited for every other stream that used workspace buffers."""
        self._retired.clear()

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self._bufs.values())


def _split_for(rows: int, tiles: int) -> int:
    """split-K factor for weight gradients: enough workgroups to fill 256 CUs, >= 512 rows each.  (A few hundred rows -- the
    CartPole-sized configurations -- stay in one piece: srl_gemm's small-product path walks them in 64-deep steps; on the
    general tiles ONE workgroup walking 16-deep steps took 29 us for a 64 x 64 x 256 product, the longest kernel of that
    step.)"""
    want = max(1, 512 // max(tiles, 1))
    return int(max(1, min(want, rows // 512)))


class HipNet:

    def __init__(self, spec: ns.NetSpec, device: str):
        self.spec = spec
        self.device = device
        self.on_gpu = device != "cpu"
        dev = device if self.on_gpu else "cpu"
        self.flat = torch.zeros(spec.total_params, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.flat)
        # PopArt running statistics (float64: mean[vd], mean_sq[vd], debiasing_term[1]); empty without PopArt
        self.popart_state = torch.zeros(2 * spec.value_dim + 1 if spec.popart else 0, dtype=torch.float64, device=dev)
        self.ws = Workspace(dev)
        self._tape = None
        self._rnn: Optional[RnnCtx] = None
        self.last_state: Dict[str, torch.Tensor] = {}
        # data parallel: called with the prefixes of the parameters whose gradient has just become final during
        # backward(), so that the trainer can start reducing finished buckets while the rest is still computed
        self.grad_ready_hook = None
        # operand ranges for the two-plane f16 forward products (gemm_bf16x3.h, NP == 2): max |weight| per layer, computed
        # on the device when the parameters have changed (`params_changed`), and max |activation| of the convolution
        # outputs, folded in by the producing kernel's epilogue (slots zeroed at the first use in a forward pass)
        self._wamax = torch.zeros(256, dtype=torch.float32, device=dev)
        self._wamax_slot: Dict[str, int] = {}
        self._wamax_stale = set()
        self._amax_next = -1
        self._gmax_next = -1
        # the weight gradients of the convolutions / the encoder's Linear run on a second stream beside the data-gradient
        # chain (both only read dz; the kernels are bound by different things -- staging loads against epilogue traffic -- and
        # fill each other's stalls: 156.3 -> 151.8 ms per update, same box); joined at the end of backward().
        # SRL_WGRAD_STREAM=0: everything on the compute stream (A/B)
        self._wgrad_side = os.environ.get("SRL_WGRAD_STREAM", "1") != "0"
        # ... but not while the trainer runs several row-chunk pipelines side by side (`_multi`, a list shared with the twins): the
        # other pipelines fill those stalls already, and since round 6 every large kernel of the Atari update takes the whole
        # chip -- same box, alternating: 85.61 / 85.38 ms per update with the second stream, 85.10 / 85.22 without; the 512-env
        # shard 11.44 / 11.47 against 11.31 / 11.43 -- and four streams fewer compete for the hardware queues.
        # SRL_WGRAD_STREAM=2: second stream in every case (A/B)
        self._wgrad_side_always = os.environ.get("SRL_WGRAD_STREAM", "1") == "2"
        self._multi = [False]
        # ReLU derivatives from sign-bit masks written by the producing convolution (SRL_RELU_MASK=0: from its floats)
        self._relu_masks = os.environ.get("SRL_RELU_MASK", "1") != "0"
        # small MLP chains (every layer LayerNorm / Linear, no wider than 128) in one launch per direction (csrc/mlp_small.hip);
        # SRL_MLP_FUSED=0: layer by layer (A/B)
        self._mlp_fused = os.environ.get("SRL_MLP_FUSED", "1") != "0"
        self._enc_fused = os.environ.get("SRL_ENC_FUSED", "1") != "0"
        # recurrent nets over vector observations: the whole pass in chunk-major row order (forward(): `cm`); SRL_RNN_CM=0: A/B
        self._cm_enabled = os.environ.get("SRL_RNN_CM", "1") != "0"
        # an encoder's closing LayerNorm + the heads right behind it as one launch per direction (csrc/ln_heads.hip); SRL_LN_HEADS=0: A/B
        self._lnheads = os.environ.get("SRL_LN_HEADS", "1") != "0"
        self._lnheads_dv = None
        self._infer = False
        self._cm = False
        self._mlp_cache = {}
        self._pver = [0]    # parameter version, shared with the twins (a list: one object)
        self._has_twins = [False]
        self._in_update = [False]
        # > 0 while the policy serves rollout requests (ActorCriticPolicy.rollout): like the chunks of one update, consecutive
        # requests see the same parameters unless somebody said otherwise (``params_changed``), so what an executor derived from
        # them -- folded first-layer weights, pre-split weight copies: 8 launches, 225 us of a 1.28 ms request batch -- is kept
        self._serving = [0]
        # set by the trainer's chunk loop before every chunk (True: this executor's last chunk of the update; None outside the loop):
        # what may accumulate over the chunks of an update in an executor's own workspace is closed behind the last one
        self.last_chunk = None
        self._derived_on = os.environ.get("SRL_DERIVED_CACHE", "1") != "0"  # 0: recompute for every chunk (A/B)
        self._presplit_on = self._derived_on and os.environ.get("SRL_PRESPLIT", "1") != "0"  # weights split once per update
        # what a workspace buffer holds that was derived from which parameter version.  Keyed by the buffer; one dict for the
        # executor and its twins (their buffers differ) -- since round 6 the weight-only data of the two-piece encoder block
        # lives in ONE set of buffers, `wws`, which every twin reads: whoever meets a stale buffer first recomputes it on ITS
        # stream and leaves an event (`_derived_done`); the others wait for that event once (`_dsynced`, per executor).  Before,
        # each of the four executors of an update made its own copies: 4 x 225 us of small launches, which the 512-environment
        # shard of a data-parallel rank (12.3 ms per update) does not amortise.  SRL_SHARED_DERIVED=0: per executor again (A/B)
        self._derived = {}
        self._devent = {}
        self._dsynced = {}
        self._shared_derived = os.environ.get("SRL_SHARED_DERIVED", "1") != "0"
        self._early_derived = os.environ.get("SRL_EARLY_DERIVED", "1") != "0"
        self.wws = self.ws
        self._side_stream = None
        self._side_used = False
        # SRL_EXPLICIT_CONV=1 forces the im2col + GEMM + col2im fallback (kept for geometries the implicit
        # kernels reject, and as a cross-check of the implicit path in the tests)
        self.force_explicit_conv = os.environ.get("SRL_EXPLICIT_CONV", "0") == "1"
        # rows one encoder pass takes.  Not an addressing limit any more (the convolution entry points walk a batch whose
        # tensors exceed 32-bit offsets in runs of images, csrc/conv.hip images_per_launch): a footprint choice -- 16 GiB for
        # the widest activation of the pass
        per_row = 1
        for enc in list(spec.obs_encoders) + list(spec.state_encoders or []):
            for L in enc.layers:
                if isinstance(L, ns.ConvSpec):
                    per_row = max(per_row, L.out_hw[0] * L.out_hw[1] * L.cout,
                                  (L.in_hw[0] + 2 * L.pad) * (L.in_hw[1] + 2 * L.pad) * L.cin)
                    if self.force_explicit_conv:
                        per_row = max(per_row, L.out_hw[0] * L.out_hw[1] * L.cin * L.k * L.k)
                elif isinstance(L, ns.ConvNdSpec):
                    vin = math.prod(d + 2 * p for d, p in zip(L.in_sp, L.pads)) * L.cin
                    per_row = max(per_row, vin, math.prod(L.out_sp) * max(L.cout, L.cin * math.prod(L.kern)))
                elif isinstance(L, ns.LinearSpec):
                    per_row = max(per_row, L.in_features, L.out_features)
        self.encoder_rows = max(1, (16 << 30) // (4 * per_row))
        # the pre-split block addresses an activation with 31-bit byte offsets (srl_h2_conv: n * 20 * 20 * 32 * 4 < 2 GiB): an encoder
        # it serves goes through in pieces of at most 32 768 rows (`analyze` on a whole 4096 x 128 sample; the trainer's own row
        # chunks are smaller anyway)
        from srl_amd.algorithm import h2path
        if h2path.ENABLED and any(h2path.match(enc.layers) is not None for enc in self._encoders()):
            self.encoder_rows = min(self.encoder_rows, 32768)

    # ------------------------------------------------------------------ parameters / checkpoints
    def ref_names(self):
        """The reference's state_dict keys of the float32 parameters, in order."""
        return [info.key for info in self.spec.params.values()]

    def load_reference_state(self, state: Dict[str, torch.Tensor]):
        want = self.ref_names() + (list(self.spec.popart_keys) if self.spec.popart else [])
        missing = [k for k in want if k not in state]
        extra = [k for k in state if k not in want]
        if missing or extra:
            raise KeyError(f"state_dict mismatch: missing {missing}, unexpected {extra}")
        host = torch.zeros(self.spec.total_params, dtype=torch.float32)
        for info in self.spec.params.values():
            t = torch.as_tensor(state[info.key]).detach().cpu()
            if tuple(t.shape) != info.ref_shape:
                raise ValueError(f"{info.key}: shape {tuple(t.shape)} != {info.ref_shape}")
            host[info.offset:info.offset + info.numel] = info.to_internal(t)
        self.flat.copy_(host)
        self.params_changed()
        if self.spec.popart:  # [mean(vd), mean_sq(vd), debiasing_term(1)] float64
            rms = torch.cat([torch.as_tensor(state[k]).detach().cpu().double().reshape(-1) for k in self.spec.popart_keys])
            self.popart_state.copy_(rms)

    def reference_state(self) -> "OrderedDict[str, torch.Tensor]":
        host = self.flat.detach().cpu()
        out = OrderedDict()
        params = list(self.spec.params.values())
        # the PopArt head's float64 running statistics sit right behind its weight and bias, as in the reference module's
        # state_dict (popart.py:19-31) -- i.e. in front of PPG's auxiliary head, the net's last child
        head = self.spec.popart_keys[0].split(".")[0] + "." if self.spec.popart else None
        last_of_head = max((i for i, info in enumerate(params) if head and info.key.startswith(head)), default=-1)
        for i, info in enumerate(params):
            out[info.key] = info.to_reference(host[info.offset:info.offset + info.numel])
            if i == last_of_head:
                rms, vd = self.popart_state.detach().cpu(), self.spec.value_dim
                out[self.spec.popart_keys[0]], out[self.spec.popart_keys[1]] = rms[:vd].clone(), rms[vd:2 * vd].clone()
                out[self.spec.popart_keys[2]] = rms[2 * vd:].clone()
        return out

    def flat_to_reference(self, flat_host: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        """Split any flat buffer with the parameter layout (gradients, Adam moments) into named tensors."""
        return OrderedDict((info.key, info.to_reference(flat_host[info.offset:info.offset + info.numel]))
                           for info in self.spec.params.values())

    def reference_to_flat(self, named: Dict[str, torch.Tensor]) -> torch.Tensor:
        host = torch.zeros(self.spec.total_params, dtype=torch.float32)
        for info in self.spec.params.values():
            host[info.offset:info.offset + info.numel] = info.to_internal(torch.as_tensor(named[info.key]).cpu())
        return host

    # ------------------------------------------------------------------ observation ring (runtime/obs_ring.py)
    def _encoders(self):
        return list(self.spec.obs_encoders) + list(self.spec.state_encoders or [])

    def obs_stage_layout(self) -> Dict[str, tuple]:
        """key -> layout in which an ``ObsRing`` keeps the rows of that observation: ("s2d", block) for an image whose
        strided first convolution reads the space-to-depth re-tiling (the ring then also keeps the whole-observation
        LayerNorm statistics of every row), ("raw",) otherwise."""
        out = {}
        for enc in self._encoders():
            conv = next((L for L in enc.layers if isinstance(L, ns.ConvSpec) and L.first), None)
            lay = ("s2d", int(conv.s2d)) if conv is not None and conv.s2d and not self.force_explicit_conv else ("raw",)
            if out.setdefault(enc.key, lay) != lay:
                out[enc.key] = ("raw",)
        return out

    def obs_raw_shapes(self) -> Dict[str, tuple]:
        return {enc.key: ((enc.shape,) if isinstance(enc.shape, int) else tuple(enc.shape)) for enc in self._encoders()}

    # ------------------------------------------------------------------ operand ranges (two-plane f16 forward products)
    RANGE_SLOTS = 1024  # tracked activations / gradients per pass (a piece-wise encoder pass of a deep tower uses dozens)

    def params_changed(self):
        """The trainer / a checkpoint load / a broadcast rewrote ``flat``: cached per-layer weight ranges are stale, and so is
        everything derived from the weights that an executor keeps between chunks (``_derived_fresh``)."""
        self._wamax_stale = set(self._wamax_slot)
        self._pver[0] += 1

    def _derived_fresh(self, what: str, ptr: int) -> bool:
        """Whether this executor's buffer ``ptr`` still holds ``what`` (weights regrouped for a data gradient, the first layer's
        folded weights) computed from the CURRENT parameters: those depend on the weights only, yet were recomputed for every
        chunk of an update (160 + 32 launches of 5-22 us).  Marks it fresh for the caller, who recomputes on False."""
        if not (self._in_update[0] or self._serving[0]) or not self._derived_on:  # only between the chunks of one trainer update
            # (``chunks_of_one_update``) and between the request batches a policy serves (whose parameters change through
            # load_state_dict / broadcast_parameters / the trainer, all of which say so): anybody else may have rewritten ``flat``
            self._derived.pop(ptr, None)
            return False
        # keyed by the BUFFER: two kinds of derived data may share one (the first layer's folded weights in the block kernel's
        # and in the per-position kernel's format, chosen by the row count of a call) -- fresh is what was written there last
        key, val = ptr, (what, self._pver[0], self.flat.data_ptr())
        if self._derived.get(key) == val:
            ev = self._devent.get(ptr)
            if ev is not None and self._dsynced.get(ptr) is not ev:   # written by another executor, maybe on another stream
                cur = torch.cuda.current_stream()
                if cur != ev[1]:
                    cur.wait_event(ev[0])
                self._dsynced[ptr] = ev
            return True
        self._derived[key] = val
        self._devent.pop(ptr, None)
        return False

    def _derived_done(self, ptr: int):
        """The launches that fill the SHARED buffer ``ptr`` (`wws`) are enqueued: the twins order themselves behind them."""
        if not self.on_gpu or self.wws is self.ws and not self._has_twins[0] or torch.cuda.is_current_stream_capturing():
            return
        ev = (torch.cuda.Event(), torch.cuda.current_stream())
        ev[0].record(ev[1])
        self._devent[ptr] = self._dsynced[ptr] = ev

    def _presplit(self, key: str, src_ptr: int, numel: int, range_ptr: int) -> Optional[int]:
        """The weight tensor at ``src_ptr`` as the two f16 pieces the two-piece kernels would otherwise make of it in every tile
        that stages it (``hip.presplit``), computed once per update and executor; None outside the trainer's chunk loop (a
        rollout's single pass does not repay the extra launch) or with SRL_PRESPLIT=0."""
        if not self._in_update[0] or not self._presplit_on or numel % 4 or src_ptr % 16:
            return None
        buf = self.ws.get(key, numel).data_ptr()
        if not self._derived_fresh(key, buf):
            hip.presplit(src_ptr, range_ptr, buf, numel)
        return buf

    def chunks_of_one_update(self, on: bool):
        """The trainer brackets the chunk loop of one update with this: inside it the parameters do not change, so what an
        executor derived from them for the first chunk serves the following ones."""
        self._in_update[0] = bool(on)
        if not on:
            self.last_chunk = None

    def prepare_derived(self):
        """Top of a trainer update (the parameters are final until the optimiser step): enqueue what the two-piece encoder
        blocks derive from them -- h2p weight copies in three orientations, scales, bounds, the first layer's folded weights --
        before the host starts on the update's head.  Opens the window `chunks_of_one_update` describes (the chunk loop closes
        it).  Nothing to do before the first forward pass has built the blocks.  SRL_EARLY_DERIVED=0: off (A/B)."""
        blocks = [b for b in self.__dict__.get("_h2_blocks", {}).values() if b is not None]
        if not (self.on_gpu and self._derived_on and self._early_derived and blocks):
            return
        self._in_update[0] = True
        for blk in blocks:
            blk.prepare()

    def refresh_weight_ranges(self):
        """Recompute every stale weight range now, on the current stream (before two row-chunk pipelines that share the
        slots start side by side)."""
        for prefix in sorted(self._wamax_stale):
            info = self.spec.params[f"{prefix}.weight"]
            self._weight_range(prefix, info.numel)

    def twin(self) -> "HipNet":
        """A second executor over the SAME parameters (and weight ranges) with its own workspace, tape and gradient
        buffer: two row chunks of one batch can then go through forward / backward side by side on two streams."""
        t = object.__new__(HipNet)
        t.__dict__.update(self.__dict__)
        t.ws = Workspace(self.device if self.on_gpu else "cpu")
        t.grad = torch.zeros_like(self.flat)
        t._tape, t._rnn, t.last_state = None, None, {}
        t.grad_ready_hook = None
        t._amax_next = t._gmax_next = -1
        t._side_stream, t._side_used = None, False
        t._dsynced = {}
        t._h2_blocks = {}
        self._has_twins[0] = True
        if not self._shared_derived:
            t.wws, t._derived, t._devent = t.ws, {}, {}
        return t

    def _weight_range(self, prefix: str, numel: int) -> int:
        """Device pointer of max |weight| of layer ``prefix`` (recomputed after ``params_changed``)."""
        slot = self._wamax_slot.get(prefix)
        if slot is None:
            if len(self._wamax_slot) >= self._wamax.numel():
                return None
            slot = self._wamax_slot[prefix] = len(self._wamax_slot)
            self._wamax_stale.add(prefix)
        ptr = self._wamax.data_ptr() + 4 * slot
        if prefix in self._wamax_stale:
            self._wamax[slot:slot + 1].zero_()
            hip.absmax(self._p(f"{prefix}.weight"), numel, ptr)
            self._wamax_stale.discard(prefix)
        return ptr

    H2_MIN_ROWS = int(os.environ.get("SRL_H2_MIN_ROWS", "256"))

    def _h2_for(self, enc):
        """This executor's pre-split block for an encoder that starts with the Atari stack (h2path.match), or None."""
        from srl_amd.algorithm import h2path
        if not h2path.ENABLED or not self.on_gpu:
            return None
        # one block per ENCODER (keyed by its first layer's parameter prefix, not by the observation key: the separate actor
        # and critic encoders of `shared_backbone=False` both read "obs" and have weights, gradients and scales of their own)
        cache = self.__dict__.setdefault("_h2_blocks", {})
        key = enc.layers[0].prefix if enc.layers else None
        if key not in cache:
            m = h2path.match(enc.layers)
            cache[key] = h2path.H2Cnn(self, m) if m is not None else None
        return cache[key]

    def reset_open_accumulations(self):
        """The trainer, in front of a chunk loop: an accumulation a failed update left open is not continued."""
        for blk in self.__dict__.get("_h2_blocks", {}).values():
            if blk is not None:
                blk.open, blk._open_ws = False, None

    def _act_range(self) -> int:
        """A fresh device float for the range of an activation of this forward pass (zero until its producer ran)."""
        amax = self.ws.get("act_absmax", self.RANGE_SLOTS)
        if self._amax_next < 0:
            amax.zero_()
            self._amax_next = 0
        slot = self._amax_next
        if slot >= self.RANGE_SLOTS:
            return None  # not tracked: the consumers take the three-plane kernels
        self._amax_next += 1
        return amax.data_ptr() + 4 * slot

    def _on_side(self, fn):
        """Run ``fn`` (launches of a weight gradient) on the second stream, after everything enqueued so far on the
        compute stream; without the switch, inline.  (While gradient buckets are being released, too: ``_release`` records a
        bucket's event behind BOTH streams -- until round 5 such passes ran everything inline, which cost the last chunk of every
        pipeline, i.e. EVERY chunk of an 8-GPU shard, the overlap.)"""
        if not self._wgrad_side or (self._multi[0] and not self._wgrad_side_always):
            fn()
            return
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        self._side_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side_stream):
            fn()
        self._side_used = True

    def _release(self, prefixes):
        """Data parallel: the gradients of these parameters are final in this executor's order of work.  Their weight gradients
        may be running on the second stream: the hook then runs under THAT stream, behind a wait for the compute stream, so the
        event the reducer records (and a bucket's fold / all-reduce, if this call completes it) comes after everything both
        streams have enqueued -- and the compute stream goes on with the data-gradient chain without waiting for anybody."""
        if self.grad_ready_hook is None:
            return
        if self._side_used:
            self._side_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side_stream):
                self.grad_ready_hook(prefixes)
        else:
            self.grad_ready_hook(prefixes)

    def _join_side(self):
        if self._side_used:
            torch.cuda.current_stream().wait_stream(self._side_stream)
            self._side_used = False
        self.ws.release_retired()

    def _grad_range(self, g: Optional["Buf"] = None) -> int:
        """A fresh device float for the range of a gradient of this backward pass; with ``g`` (dense rows) it is filled
        by one pass over ``g`` (srl_absmax) -- for gradients whose producer does not track it."""
        amax = self.ws.get("grad_absmax", self.RANGE_SLOTS)
        if self._gmax_next < 0:
            amax.zero_()
            self._gmax_next = 0
        slot = self._gmax_next
        if slot >= self.RANGE_SLOTS:
            return None  # not tracked: the consumers take the three-plane kernels
        self._gmax_next += 1
        ptr = amax.data_ptr() + 4 * slot
        if g is not None:
            assert g.ld == g.cols
            hip.absmax(g.ptr, g.rows * g.cols, ptr)
        return ptr

    def _p(self, name):
        return self.flat.data_ptr() + 4 * self.spec.params[name].offset

    def _g(self, name):
        return self.grad.data_ptr() + 4 * self.spec.params[name].offset

    def zero_grad(self):
        self.grad.zero_()

    # ------------------------------------------------------------------ layer kernels
    def _buf(self, name, rows, cols) -> Buf:
        t = self.ws.get(name, rows * cols)
        return Buf(t.data_ptr(), cols, rows, cols)

    # ------------------------------------------------------------------ wide dense layers on pre-split operands (round 6)
    # A Linear with thousands of inputs over thousands of rows (the football preset's tower, football_rnn.py:34-55 /
    # cnn.py:96-135: 22528 -> 11264 -> 5632 -> 2816 -> 1408 -> 704) on the kernels the Atari encoder's Linear runs on: the input
    # rows and the output gradient are split into two f16 pieces ONCE per pass (srl_h2_pack_rows: the same 4 bytes per element),
    # the weights once per update in both orientations, and the three products -- srl_h2_gemm forward and data gradient,
    # srl_h2_wgrad_dense -- move bytes by LDS-DMA and multiply three piece products.  Before, only the FIRST Linear behind the
    # convolutions knew its input's range; the others ran the six-product bf16 kernels with register staging (gemm_bf16x3.h:
    # 94 / 128 / 194 ms per launch of the 22528 -> 11264 layer at 51 200 rows, ~130-280 TFLOP/s).  SRL_H2_DENSE=0: off (A/B).
    H2_DENSE = os.environ.get("SRL_H2_DENSE", "1") != "0"
    H2D_MIN_ROWS = int(os.environ.get("SRL_H2_DENSE_MIN_ROWS", "4096"))
    (H2D_MX, H2D_SX, H2D_MY, H2D_MDZ, H2D_SDZ, H2D_NP) = range(6)   # device floats of one pass through one layer
    (H2D_SW, H2D_RW, H2D_SWT, H2D_RWT, H2D_NW) = range(5)           # ... of one layer's weights (per parameter version)

    def _h2d_ok(self, L, x: Buf) -> bool:
        return (self.H2_DENSE and self.on_gpu and x.rows >= self.H2D_MIN_ROWS and L.in_features >= 1024 and L.out_features >= 128
                and L.in_features % 32 == 0 and L.out_features % 32 == 0 and L.act in (0, hip.ACT_RELU)
                and x.ld == x.cols == L.in_features)

    def _h2d_weights(self, L, transposed: bool):
        """(h2p copy of the layer's weight in the orientation asked for, pointer of its slot array): once per parameter version."""
        K, N = L.in_features, L.out_features
        slots = self.ws.get(f"{L.prefix}.h2d.wslots", self.H2D_NW)
        name = f"{L.prefix}.h2d.wt" if transposed else f"{L.prefix}.h2d.w"
        buf = self.ws.get(name, N * K).data_ptr()
        if not self._derived_fresh(name, buf):
            amax = self._weight_range(L.prefix, N * K)
            sp = slots.data_ptr()
            if transposed:   # rows = input features: the data gradient's channels operand
                hip.h2_weights(self._p(f"{L.prefix}.weight"), K, N, 1, amax, sp + 4 * self.H2D_SWT, sp + 4 * self.H2D_RWT, buf)
            else:
                hip.h2_weights(self._p(f"{L.prefix}.weight"), N, K, 0, amax, sp + 4 * self.H2D_SW, sp + 4 * self.H2D_RW, buf)
        return buf, slots.data_ptr()

    @staticmethod
    def _h2d_pieces(n: int, width: int):
        """Row ranges whose operands stay below srl_h2_gemm's 4 GiB (32-bit byte offsets)."""
        step = max(256, ((0xfff00000 // (4 * width)) // 256) * 256)
        return [(r0, min(n, r0 + step)) for r0 in range(0, n, step)]

    def _linear_fwd_h2d(self, L, x: Buf, tag: str, x_range) -> Buf:
        n, K, N = x.rows, L.in_features, L.out_features
        sl = self.ws.get(f"{tag}{L.prefix}.h2d.slots", self.H2D_NP)
        sl.zero_()
        sp = sl.data_ptr()
        wh, wsp = self._h2d_weights(L, False)
        if x_range is None:
            x_range = sp + 4 * self.H2D_MX
            hip.absmax(x.ptr, n * K, x_range)
        xh = self.ws.get(f"{tag}{L.prefix}.h2d.xh", n * K).data_ptr()
        hip.h2_pack_rows(x.ptr, x.ld, n, K, xh, absmax=x_range, scale_out=sp + 4 * self.H2D_SX)
        y = self._buf(f"{tag}{L.prefix}.y", n, N)
        mask = self.ws.get(f"{tag}{L.prefix}.mask", n * N // 32, torch.int32).data_ptr() if L.act == hip.ACT_RELU else None
        for r0, r1 in self._h2d_pieces(n, max(K, N)):
            hip.h2_gemm(xh + 4 * r0 * K, wh, sp + 4 * self.H2D_SX, wsp + 4 * self.H2D_SW, r1 - r0, N, K, y.ptr + 4 * r0 * N,
                        bias=self._p(f"{L.prefix}.bias"), act=1 if L.act == hip.ACT_RELU else 0,
                        mask_out=(mask + 4 * (r0 * N // 32)) if mask else None, out_absmax=sp + 4 * self.H2D_MY)
        self._y_range = sp + 4 * self.H2D_MY
        # what the backward pass reads again, found by the tape record's input (the backward pass of an encoder cut into pieces
        # walks every piece under ONE tag: names would find another piece's buffers)
        self.__dict__.setdefault("_h2d_saved", {})[(L.prefix, x.ptr)] = (xh, sp)
        return y._replace(mask=mask) if mask else y

    def _linear_bwd_h2d(self, L, x: Buf, dz: Buf, in_act: int, need_dx: bool, tag: str, dz_range, dx_range) -> Optional[Buf]:
        n, K, N = x.rows, L.in_features, L.out_features
        xh, sp = self._h2d_saved[(L.prefix, x.ptr)]   # the forward pass's split input and its slots (S_X is read again)
        if dz_range is None:
            dz_range = sp + 4 * self.H2D_MDZ
            hip.absmax(dz.ptr, n * N, dz_range)
        dzh = self.ws.get(f"{tag}{L.prefix}.h2d.dzh", n * N).data_ptr()
        hip.h2_pack_rows(dz.ptr, dz.ld, n, N, dzh, absmax=dz_range, scale_out=sp + 4 * self.H2D_SDZ)

        def wg():
            side = self._side_stream is not None and torch.cuda.current_stream() == self._side_stream
            wsp = self.ws.get("h2tn_side" if side else "h2tn", hip.h2_wgrad_dense_workspace(n, N, K)).data_ptr()
            hip.h2_wgrad_dense(dzh, xh, sp + 4 * self.H2D_SDZ, sp + 4 * self.H2D_SX, n, N, K, wsp, self._g(f"{L.prefix}.weight"))
            hip.colsum(dz.ptr, dz.ld, n, N, self._g(f"{L.prefix}.bias"), accumulate=True)

        if need_dx:
            self._on_side(wg)
        else:
            wg()
        if not need_dx:
            return None
        wth, wsp = self._h2d_weights(L, True)
        dx = self._buf(f"{tag}{L.prefix}.dx", n, K)
        x_mask = x.mask if in_act == hip.ACT_RELU else None
        if in_act == hip.ACT_RELU and x_mask is None:
            # the producer wrote no sign words (a convolution with 4 or 8 channels: its y_mask needs 32-channel blocks)
            x_mask = self.ws.get(f"{tag}{L.prefix}.h2d.xmask", n * K // 32, torch.int32).data_ptr()
            hip.relu_mask(x.ptr, n * K, x_mask)
        for r0, r1 in self._h2d_pieces(n, max(K, N)):
            hip.h2_gemm(dzh + 4 * r0 * N, wth, sp + 4 * self.H2D_SDZ, wsp + 4 * self.H2D_SWT, r1 - r0, K, N, dx.ptr + 4 * r0 * K,
                        mask_in=(x_mask + 4 * (r0 * K // 32)) if x_mask else None, mask_in_h2order=False, out_absmax=dx_range)
        return dx

    CONV_SMALL = os.environ.get("SRL_CONV_SMALL", "1") != "0"

    def _conv_small(self, L, desc) -> bool:
        """A convolution on a plain NHWC activation with 4 or 8 channels on both sides, 3 x 3 or 5 x 5, stride 1, no padding: the direct
        vector-unit kernels (csrc/conv_small.hip).  SRL_CONV_SMALL=0: the implicit GEMMs (A/B)."""
        return (self.CONV_SMALL and self.on_gpu and not L.first and not L.pad and L.stride == 1 and L.k in (3, 5)
                and hip.conv2d_small_supported(desc))

    def _linear_fwd(self, L: ns.LinearSpec, x: Buf, tag: str, x_range: Optional[int] = None) -> Buf:
        """``x_range``: device float bounding max |x| when the producer tracked it (a convolution's output): with the
        weight's range the product runs on two f16 pieces per operand (srl_gemm_desc::a_absmax)."""
        self._y_range = None
        if self._h2d_ok(L, x):
            return self._linear_fwd_h2d(L, x, tag, x_range)
        y = self._buf(f"{tag}{L.prefix}.y", x.rows, L.out_features)
        w_range = self._weight_range(L.prefix, L.out_features * L.in_features) if x_range is not None else None
        w, pre = self._p(f"{L.prefix}.weight"), None
        if hip.gemm_two_piece(x.rows, L.out_features, L.in_features, x.ptr, x.ld, w, L.in_features, x_range, w_range):
            pre = self._presplit(f"{L.prefix}.w2h", w, L.out_features * L.in_features, w_range)  # also serves the data gradient
        hip.gemm(x.rows, L.out_features, L.in_features, x.ptr, x.ld, 0, pre or w, L.in_features, 0,
                 y.ptr, y.ld, bias=self._p(f"{L.prefix}.bias"), act=L.act, a_absmax=x_range, b_absmax=w_range,
                 b_presplit=pre is not None)
        return y

    def _wgrad(self, out_f, in_f, rows, dz: Buf, x_ptr, x_ld, gw_ptr, gb_ptr=None, dz_range=None, x_range=None):
        """gw += dz^T x; gb += column sums of dz -- from the same kernel when the operands allow it.  ``dz_range`` /
        ``x_range``: device floats bounding the operands (both given: two f16 pieces per operand)."""
        tiles = ((out_f + 127) // 128) * ((in_f + 127) // 128)
        split = _split_for(rows, tiles)
        # split-K slabs: launches on the second stream (weight gradients beside the data-gradient chain) have their own
        wsn = "splitk_side" if torch.cuda.current_stream() == self._side_stream else "splitk"
        wsp = self.ws.get(wsn, split * out_f * in_f).data_ptr() if split > 1 else None
        fused = gb_ptr is not None and hip.gemm_colsum_ok(out_f, in_f, rows, dz.ptr, dz.ld, x_ptr, x_ld, 1)
        if dz_range is None or x_range is None:
            dz_range = x_range = None
        hip.gemm(out_f, in_f, rows, dz.ptr, dz.ld, 1, x_ptr, x_ld, 1, gw_ptr, in_f, accumulate=True, split_k=split,
                 workspace=wsp, a_colsum=gb_ptr if fused else None, a_absmax=dz_range, b_absmax=x_range)
        if gb_ptr is not None and not fused:
            hip.colsum(dz.ptr, dz.ld, rows, out_f, gb_ptr, accumulate=True)

    def _linear_bwd(self, L: ns.LinearSpec, x: Buf, dz: Buf, in_act: int, need_dx: bool, tag: str,
                    dx_into: Optional[Buf] = None, dx_accumulate=False, x_range=None, dz_range=None, dx_range=None) -> Optional[Buf]:
        """``x_range`` (the forward pass's range of this layer's input), ``dz_range``: both known -> the two products run on
        two f16 pieces per operand; ``dx_range``: device float the data gradient's range is folded into."""
        if (dx_into is None and not dx_accumulate and self._h2d_ok(L, x) and dz.ld == dz.cols == L.out_features
                and in_act in (0, hip.ACT_RELU) and (L.prefix, x.ptr) in self.__dict__.get("_h2d_saved", {})):
            return self._linear_bwd_h2d(L, x, dz, in_act, need_dx, tag, dz_range, dx_range)
        wg = lambda: self._wgrad(L.out_features, L.in_features, x.rows, dz, x.ptr, x.ld, self._g(f"{L.prefix}.weight"),
                                 self._g(f"{L.prefix}.bias"), dz_range, x_range)
        if x_range is not None and need_dx and not dx_accumulate:
            self._on_side(wg)  # a big encoder Linear: its weight gradient beside the data gradient
        else:
            wg()
        if not need_dx:
            return None
        dx = dx_into or self._buf(f"{tag}{L.prefix}.dx", x.rows, L.in_features)
        w_range = self._weight_range(L.prefix, L.out_features * L.in_features) if dz_range is not None else None
        x_mask = x.mask if in_act == hip.ACT_RELU else None
        w, pre = self._p(f"{L.prefix}.weight"), None
        if hip.gemm_two_piece(x.rows, L.in_features, L.out_features, dz.ptr, dz.ld, w, L.in_features, dz_range, w_range):
            pre = self._presplit(f"{L.prefix}.w2h", w, L.out_features * L.in_features, w_range)  # the forward pass's copy
        hip.gemm(x.rows, L.in_features, L.out_features, dz.ptr, dz.ld, 0, pre or w, L.in_features,
                 1, dx.ptr, dx.ld, dact_src=x.ptr if in_act and x_mask is None else None, ld_dact=x.ld, dact=in_act,
                 accumulate=dx_accumulate, a_absmax=dz_range, b_absmax=w_range, out_absmax=dx_range, dact_mask=x_mask,
                 b_presplit=pre is not None)
        return dx

    def _ln_fwd(self, L: ns.LayerNormSpec, x: Buf, tag: str):
        y = self._buf(f"{tag}{L.prefix}.y", x.rows, L.dim)
        mean = self.ws.get(f"{tag}{L.prefix}.mean", x.rows)
        rstd = self.ws.get(f"{tag}{L.prefix}.rstd", x.rows)
        hip.layernorm_fwd(x.ptr, x.ld, self._p(f"{L.prefix}.weight"), self._p(f"{L.prefix}.bias"), x.rows, L.dim, y.ptr,
                          y.ld, mean.data_ptr(), rstd.data_ptr())
        return y, (mean, rstd)

    def _ln_bwd(self, L: ns.LayerNormSpec, x: Buf, saved, dy: Buf, in_act: int, need_dx: bool, tag: str, dx_range=None):
        """``dx_range``: device float the data gradient's range is folded into (for the two-piece products of the layer below)."""
        mean, rstd = saved
        dx = self._buf(f"{tag}{L.prefix}.dx", x.rows, L.dim) if need_dx else None
        hip.layernorm_bwd(dy.ptr, dy.ld, x.ptr, x.ld, self._p(f"{L.prefix}.weight"), mean.data_ptr(), rstd.data_ptr(),
                          x.rows, L.dim, dx.ptr if dx else None, dx.ld if dx else 0, in_act,
                          self._g(f"{L.prefix}.weight"), self._g(f"{L.prefix}.bias"), dx_absmax=dx_range if dx else None)
        return dx

    # ------------------------------------------------------------------ recurrent layers (GRU + auto reset)
    def _gru_fwd(self, G: ns.GruSpec, x: Buf, tag: str):
        """x: time-major [T*B, H].  Returns (y time-major [T*B, H], saved, last states [layers, N, H])."""
        ctx = self._rnn
        if ctx is None:
            raise hip.HipError("recurrent backbone: forward() needs the `rnn` context (chunking, states, on_reset)")
        T, B, C, H = ctx.T, ctx.B, ctx.C, G.hidden
        n, K = T * B, T // C
        N = K * B
        assert x.rows == n and x.cols == H and x.ld == H and T % C == 0
        if K > 1 and not self._cm:
            xc = self._buf(f"{tag}{G.prefix}.xc", n, H)
            hip.chunk_rows(x.ptr, xc.ptr, T, B, C, H)
        else:  # (one chunk, or the whole pass runs on chunk-major rows already)
            xc = x
        h0 = ctx.h0[tag]
        SW = G.state_width
        assert tuple(h0.shape) == (G.layers, N, SW) and h0.dtype == torch.float32 and h0.is_contiguous()
        rs = ctx.reset
        rptr = (lambda c: rs.data_ptr() + c * N) if rs is not None else (lambda c: None)
        last = self.ws.get(f"{tag}{G.prefix}.last", G.layers * N * SW)[:G.layers * N * SW].view(G.layers, N, SW)
        inp, saved = xc, []
        for l in range(G.layers if G.kind == "lstm" else 0):
            w_ih, w_hh = self._p(f"{G.prefix}.weight_ih_l{l}"), self._p(f"{G.prefix}.weight_hh_l{l}")
            b_ih, b_hh = self._p(f"{G.prefix}.bias_ih_l{l}"), self._p(f"{G.prefix}.bias_hh_l{l}")
            pre = self._buf(f"{tag}{G.prefix}.gi{l}", n, 4 * H)
            hin = self._buf(f"{tag}{G.prefix}.hin{l}", n, H)
            cin = self._buf(f"{tag}{G.prefix}.cin{l}", n, H)
            cnew = self._buf(f"{tag}{G.prefix}.cnew{l}", n, H)
            y = self._buf(f"{tag}{G.prefix}.y{l}", n, H)
            hip.gemm(n, 4 * H, H, inp.ptr, inp.ld, 0, w_ih, H, 0, pre.ptr, 4 * H, bias=b_ih)  # every step at once
            # the stored state is cat(h, c) per row (autoreset_rnn.py:31-39); both halves are reset together
            hip.copy2d(h0[l].data_ptr(), SW, y.ptr, H, N, H)
            hip.gru_mask_state(y.ptr, rptr(0), N, H, hin.ptr)
            hip.copy2d(h0[l].data_ptr() + 4 * H, SW, y.ptr, H, N, H)
            hip.gru_mask_state(y.ptr, rptr(0), N, H, cin.ptr)
            seq = hip.rnn_seq_supported("lstm", H)  # the chunk's time loop inside one launch (csrc/rnn_seq.hip)
            if seq:
                hip.lstm_seq_fwd(pre.ptr, w_hh, b_hh, hin.ptr, cin.ptr, rptr(0), N, H, C, y.ptr, cnew.ptr)
            for c in range(0 if seq else C):
                o4, o1 = 4 * c * N * 4 * H, 4 * c * N * H
                hip.gemm(N, 4 * H, H, hin.ptr + o1, H, 0, w_hh, H, 0, pre.ptr + o4, 4 * H, bias=b_hh, accumulate=True)
                nxt = c + 1 < C
                o1n = 4 * (c + 1) * N * H
                hip.lstm_cell_fwd(pre.ptr + o4, cin.ptr + o1, rptr(c + 1) if nxt else None, N, H, y.ptr + o1, cnew.ptr + o1,
                                  hin.ptr + o1n if nxt else None, cin.ptr + o1n if nxt else None)
            hip.copy2d(y.ptr + 4 * (C - 1) * N * H, H, last[l].data_ptr(), SW, N, H)
            hip.copy2d(cnew.ptr + 4 * (C - 1) * N * H, H, last[l].data_ptr() + 4 * H, SW, N, H)
            saved.append((inp, pre, hin, cin, cnew))
            inp = y
        for l in range(G.layers if G.kind == "gru" else 0):
            w_ih, w_hh = self._p(f"{G.prefix}.weight_ih_l{l}"), self._p(f"{G.prefix}.weight_hh_l{l}")
            b_ih, b_hh = self._p(f"{G.prefix}.bias_ih_l{l}"), self._p(f"{G.prefix}.bias_hh_l{l}")
            gi = self._buf(f"{tag}{G.prefix}.gi{l}", n, 3 * H)
            gh = self._buf(f"{tag}{G.prefix}.gh{l}", n, 3 * H)
            hin = self._buf(f"{tag}{G.prefix}.hin{l}", n, H)
            y = self._buf(f"{tag}{G.prefix}.y{l}", n, H)
            hip.gemm(n, 3 * H, H, inp.ptr, inp.ld, 0, w_ih, H, 0, gi.ptr, 3 * H, bias=b_ih)  # every step at once
            hip.gru_mask_state(h0[l].data_ptr(), rptr(0), N, H, hin.ptr)
            seq = hip.rnn_seq_supported("gru", H)
            if seq:
                hip.gru_seq_fwd(gi.ptr, gh.ptr, w_hh, b_hh, hin.ptr, rptr(0), N, H, C, y.ptr)
            for c in range(0 if seq else C):
                o3, o1 = 4 * c * N * 3 * H, 4 * c * N * H
                hip.gemm(N, 3 * H, H, hin.ptr + o1, H, 0, w_hh, H, 0, gh.ptr + o3, 3 * H, bias=b_hh)
                nxt = c + 1 < C
                hip.gru_cell_fwd(gi.ptr + o3, gh.ptr + o3, hin.ptr + o1, rptr(c + 1) if nxt else None, N, H, y.ptr + o1,
                                 hin.ptr + 4 * (c + 1) * N * H if nxt else None)
            hip.copy2d(y.ptr + 4 * (C - 1) * N * H, H, last[l].data_ptr(), H, N, H)
            saved.append((inp, gi, gh, hin))
            inp = y
        if K > 1 and not self._cm:
            ytm = self._buf(f"{tag}{G.prefix}.ytm", n, H)
            hip.chunk_rows(inp.ptr, ytm.ptr, T, B, C, H, inverse=True)
        else:
            ytm = inp
        return ytm, (saved, ctx), last

    def _gru_bwd(self, G: ns.GruSpec, saved_all, dy: Buf, in_act: int, need_dx: bool, tag: str) -> Optional[Buf]:
        saved, ctx = saved_all
        T, B, C, H = ctx.T, ctx.B, ctx.C, G.hidden
        n, K = T * B, T // C
        N = K * B
        rs = ctx.reset
        rptr = (lambda c: rs.data_ptr() + c * N) if rs is not None else (lambda c: None)
        if K > 1 and not self._cm:
            dyc = self._buf(f"{tag}{G.prefix}.dyc", n, H)
            assert dy.ld == H
            hip.chunk_rows(dy.ptr, dyc.ptr, T, B, C, H)
        else:
            dyc = dy
        dout = dyc
        for l in range(G.layers - 1 if G.kind == "lstm" else -1, -1, -1):
            inp, pre, hin, cin, cnew = saved[l]
            w_ih, w_hh = self._p(f"{G.prefix}.weight_ih_l{l}"), self._p(f"{G.prefix}.weight_hh_l{l}")
            dh = [self._buf(f"{tag}{G.prefix}.dh{i}", N, H) for i in range(2)]
            dc = [self._buf(f"{tag}{G.prefix}.dc{i}", N, H) for i in range(2)]
            ch = cc = None
            seq = hip.rnn_seq_supported("lstm", H)
            if seq:
                hip.lstm_seq_bwd(dout.ptr, dout.ld, pre.ptr, w_hh, cin.ptr, cnew.ptr, rptr(0), N, H, C)
            for c in range(-1 if seq else C - 1, -1, -1):
                o4, o1 = 4 * c * N * 4 * H, 4 * c * N * H
                hip.lstm_cell_bwd(dout.ptr + 4 * c * N * dout.ld, ch, cc, rptr(c + 1) if c + 1 < C else None, pre.ptr + o4,
                                  cin.ptr + o1, cnew.ptr + o1, N, H, dc[c & 1].ptr)
                hip.gemm(N, H, 4 * H, pre.ptr + o4, 4 * H, 0, w_hh, H, 1, dh[c & 1].ptr, H)  # d h_in(c) = d pre . W_hh
                ch, cc = dh[c & 1].ptr, dc[c & 1].ptr
            # bias gradients (column sums of d pre) come out of the weight-gradient kernels
            self._wgrad(4 * H, H, n, pre, hin.ptr, H, self._g(f"{G.prefix}.weight_hh_l{l}"),
                        self._g(f"{G.prefix}.bias_hh_l{l}"))
            self._wgrad(4 * H, H, n, pre, inp.ptr, inp.ld, self._g(f"{G.prefix}.weight_ih_l{l}"),
                        self._g(f"{G.prefix}.bias_ih_l{l}"))
            if l == 0 and not need_dx:
                return None
            dx = self._buf(f"{tag}{G.prefix}.dx{l}", n, H)
            act = in_act if l == 0 else 0
            hip.gemm(n, H, 4 * H, pre.ptr, 4 * H, 0, w_ih, H, 1, dx.ptr, H, dact_src=inp.ptr if act else None,
                     ld_dact=inp.ld, dact=act)
            dout = dx
        for l in range(G.layers - 1 if G.kind == "gru" else -1, -1, -1):
            inp, gi, gh, hin = saved[l]
            w_ih, w_hh = self._p(f"{G.prefix}.weight_ih_l{l}"), self._p(f"{G.prefix}.weight_hh_l{l}")
            dh = [self._buf(f"{tag}{G.prefix}.dh{i}", N, H) for i in range(2)]
            carry = None
            seq = hip.rnn_seq_supported("gru", H)
            if seq:
                hip.gru_seq_bwd(dout.ptr, dout.ld, gi.ptr, gh.ptr, w_hh, hin.ptr, rptr(0), N, H, C)
            for c in range(-1 if seq else C - 1, -1, -1):
                o3, o1 = 4 * c * N * 3 * H, 4 * c * N * H
                cur = dh[c & 1]
                hip.gru_cell_bwd(dout.ptr + 4 * c * N * dout.ld, carry, rptr(c + 1) if c + 1 < C else None, gi.ptr + o3,
                                 gh.ptr + o3, hin.ptr + o1, N, H, cur.ptr)
                # d h_in(c) = dh*z + d gh . W_hh: the carry of step c-1 (masked there by on_reset[c])
                hip.gemm(N, H, 3 * H, gh.ptr + o3, 3 * H, 0, w_hh, H, 1, cur.ptr, H, accumulate=True)
                carry = cur.ptr
            # parameter gradients over all steps at once (gi / gh now hold d gi / d gh)
            self._wgrad(3 * H, H, n, gh, hin.ptr, H, self._g(f"{G.prefix}.weight_hh_l{l}"),
                        self._g(f"{G.prefix}.bias_hh_l{l}"))
            self._wgrad(3 * H, H, n, gi, inp.ptr, inp.ld, self._g(f"{G.prefix}.weight_ih_l{l}"),
                        self._g(f"{G.prefix}.bias_ih_l{l}"))
            if l == 0 and not need_dx:
                return None
            dx = self._buf(f"{tag}{G.prefix}.dx{l}", n, H)
            act = in_act if l == 0 else 0  # layer 0 reads the (activated) output of the dense stack
            hip.gemm(n, H, 3 * H, gi.ptr, 3 * H, 0, w_ih, H, 1, dx.ptr, H, dact_src=inp.ptr if act else None, ld_dact=inp.ld,
                     dact=act)
            dout = dx
        if K > 1 and not self._cm:
            dxt = self._buf(f"{tag}{G.prefix}.dxt", n, H)
            hip.chunk_rows(dout.ptr, dxt.ptr, T, B, C, H, inverse=True)
            return dxt
        return dout

    # ------------------------------------------------------------------ encoders
    def _encoder_fwd(self, enc: ns.EncoderSpec, obs: torch.Tensor, n: int, tag: str, tape: list, lnheads=None) -> Buf:
        """obs: device tensor [n, *shape] (float32 vectors; uint8 or float32 images).  ``lnheads``: ``(heads, outs)`` when the
        encoder's closing LayerNorm and the heads behind it are to run as one launch (`_lnheads_ok`): the record `lnheads` then
        closes the tape and the heads' outputs are in ``outs``."""
        cur: Optional[Buf] = None
        cur_act = 0
        cur_range = None  # device float bounding max |cur| (convolution outputs), or None: range unknown
        pending_obs_ln = None
        staged = None
        if isinstance(obs, RingObs):  # rows kept in the HBM observation ring since their rollout
            if obs.rows != n:
                raise hip.HipError(f"observation `{enc.key}`: {obs.rows} ring rows for {n} network rows")
            if obs.layout[0] == "s2d":
                staged = obs  # already in the first convolution's layout, statistics beside them
            else:
                obs = obs.gather_raw(self.ws, f"{tag}{enc.key}.ring")
        # a vector encoder in front of a recurrent backbone (the multi-agent nets: LayerNorm -> Linear -> LayerNorm -> Linear ->
        # LayerNorm, 64 wide) as ONE launch per direction: the whole-trunk fusion above stops at the recurrent cell, and layer by
        # layer these 307 200-row products and LayerNorms were ~2 ms of the SMAC-sized step (SRL_ENC_FUSED=0: layer by layer, A/B)
        if (self._mlp_fused and self._enc_fused and staged is None and isinstance(obs, torch.Tensor) and obs.dim() == 2
                and obs.dtype == torch.float32 and enc.layers and all(isinstance(L, (ns.LayerNormSpec, ns.LinearSpec)) for L in enc.layers)
                and (isinstance(enc.layers[-1], ns.LayerNormSpec) or enc.layers[-1].act == 0) and n >= 512):
            rec = self._fused_layers_fwd((tag, "enc", enc.key), f"{tag}enc.{enc.key}.", list(enc.layers), obs, n)
            if rec is not None:
                tape.append(("fusedenc", rec, None, None, 0))
                return rec["feat"]
        h2 = self._h2_for(enc)
        skip = 0
        for L in enc.layers:
            if skip:  # layers the pre-split block below has already run
                skip -= 1
                continue
            if lnheads is not None and L is enc.layers[-1] and cur is not None and cur.ld == cur.cols == L.dim and cur.rows == n:
                heads, outs = lnheads
                mean = self.ws.get(f"{tag}{L.prefix}.mean", n)
                rstd = self.ws.get(f"{tag}{L.prefix}.rstd", n)
                slabs = tape[-1][3].get("fc_slabs") if (tape and tape[-1][0] == "h2cnn") else None
                xptr, ks, stride, xb, xact, keep = slabs if slabs else (cur.ptr, 1, 0, None, 0, False)
                # (a split product: this launch adds the slabs, the bias and the ReLU while it reads them -- and, when a backward
                # pass follows, writes the finished rows to `cur`)
                hip.ln_heads_fwd(xptr, cur.ld, n, L.dim, self._p(f"{L.prefix}.weight"), self._p(f"{L.prefix}.bias"),
                                 [self._p(f"{h.prefix}.weight") for h in heads], [self._p(f"{h.prefix}.bias") for h in heads],
                                 [h.out_features for h in heads], [o.data_ptr() for o in outs], [h.out_features for h in heads],
                                 mean.data_ptr(), rstd.data_ptr(), x_slabs=ks, x_slab_stride=stride, x_bias=xb, x_act=xact,
                                 x_out=cur.ptr if (slabs and not self._infer) else None, ldxo=cur.ld)
                tape.append(("lnheads", L, cur, (mean, rstd, heads), cur_act))
                return None
            if isinstance(L, ns.LayerNormSpec):
                if cur is None:
                    if obs.dtype != torch.float32:
                        raise hip.HipError(f"vector observation `{enc.key}` must be float32, got {obs.dtype}")
                    if obs.shape[0] != n or math.prod(obs.shape[1:]) != L.dim:
                        raise hip.HipError(f"vector observation `{enc.key}`: sample rows of shape {tuple(obs.shape[1:])} for a "
                                           f"network built for ({L.dim},) ({n} rows expected, {obs.shape[0]} given)")
                    cur = Buf(obs.data_ptr(), L.dim, n, L.dim)
                y, saved = self._ln_fwd(L, cur, tag)
                tape.append(("ln", L, cur, saved, cur_act))
                cur, cur_act, cur_range = y, 0, None
            elif isinstance(L, ns.LinearSpec):
                if cur.cols != L.in_features:  # Flatten after the convolution stack: [n*OH*OW, C] -> [n, OH*OW*C]
                    assert cur.rows * cur.cols == n * L.in_features and cur.ld == cur.cols
                    cur = Buf(cur.ptr, L.in_features, n, L.in_features, cur.mask)
                y = self._linear_fwd(L, cur, tag, cur_range)
                tape.append(("linear", L, cur, cur_range, cur_act))
                cur, cur_act, cur_range = y, L.act, self._y_range   # (the wide-layer path measures its output's range)
            elif isinstance(L, ns.ObsLayerNormSpec):
                pending_obs_ln = L
                if isinstance(obs, torch.Tensor) and (obs.dim() < 2 or obs.shape[0] != n or
                                                      math.prod(obs.shape[1:]) != math.prod(L.shape)):
                    # the kernels take sizes from the network's spec: a sample of another shape must not reach them
                    raise hip.HipError(f"image observation `{enc.key}`: sample rows of shape {tuple(obs.shape[1:])} "
                                       f"for a network built for {tuple(L.shape)} ({n} rows expected, {obs.shape[0]} given)")
                if L.explicit:  # written out once, channels-last float32; the convolutions then see a plain activation
                    c, h, w = L.shape
                    is_u8 = obs.dtype == torch.uint8
                    if not is_u8 and obs.dtype != torch.float32:
                        raise hip.HipError(f"image observation `{enc.key}` must be uint8 or float32, got {obs.dtype}")
                    mean = self.ws.get(f"{tag}{L.prefix}.mean", n)
                    rstd = self.ws.get(f"{tag}{L.prefix}.rstd", n)
                    hip.obs_ln_stats(obs.data_ptr(), is_u8, n, c * h * w, mean.data_ptr(), rstd.data_ptr())
                    y = self._buf(f"{tag}{L.prefix}.y", n * h * w, c)
                    hip.obs_ln_nhwc(obs.data_ptr(), is_u8, mean.data_ptr(), rstd.data_ptr(), self._p(f"{L.prefix}.weight"),
                                    self._p(f"{L.prefix}.bias"), n, c, h, w, y.ptr)
                    tape.append(("obsln", L, obs, (is_u8, mean, rstd, n), 0))
                    cur, cur_act, cur_range = y, 0, None
            elif isinstance(L, ns.PoolSpec):
                (h, w), (ph, pw) = L.in_hw, L.out_hw
                assert cur.ld == L.c and cur.rows == n * h * w
                y = self._buf(f"{tag}{L.prefix}.y", n * ph * pw, L.c)
                hip.maxpool2_nhwc_fwd(cur.ptr, n, h, w, L.c, y.ptr)
                tape.append(("pool", L, cur, n, cur_act))
                cur, cur_act, cur_range = y, 0, None  # the activation's derivative is applied by the pooling backward at the winner
            elif isinstance(L, ns.PoolNdSpec):
                assert cur.ld == L.c and cur.rows == n * math.prod(L.in_sp)
                y = self._buf(f"{tag}{L.prefix}.y", n * math.prod(L.out_sp), L.c)
                hip.maxpool_ndhwc_fwd(cur.ptr, n, (*L.in_sp, L.c), L.win, y.ptr)
                tape.append(("poolnd", L, cur, n, cur_act))
                cur, cur_act, cur_range = y, 0, None
            elif isinstance(L, ns.ConvNdSpec):
                sp = L.in_sp
                assert cur.ld == L.cin and cur.rows == n * math.prod(sp)
                if any(L.pads):
                    sp = tuple(d + 2 * p for d, p in zip(sp, L.pads))
                    xp = self._buf(f"{tag}{L.prefix}.xp", n * math.prod(sp), L.cin)
                    hip.pad_ndhwc(cur.ptr, n, (*L.in_sp, L.cin), L.pads, xp.ptr, L.pad_mode)
                    cur = xp
                m, kdim = n * math.prod(L.out_sp), L.cin * math.prod(L.kern)
                P = self._buf(f"{tag}{L.prefix}.P", m, kdim)
                hip.im2col_ndhwc(cur.ptr, n, (*sp, L.cin), L.kern, L.stride, P.ptr)
                y = self._buf(f"{tag}{L.prefix}.y", m, L.cout)
                hip.gemm(m, L.cout, kdim, P.ptr, kdim, 0, self._p(f"{L.prefix}.weight"), kdim, 0, y.ptr, y.ld,
                         bias=self._p(f"{L.prefix}.bias"), act=L.act)
                tape.append(("convnd", L, cur, (P, n, sp), cur_act))
                cur, cur_act, cur_range = y, L.act, None
            elif isinstance(L, ns.ConvSpec):
                oh, ow = L.out_hw
                m = n * oh * ow
                kdim = L.cin * L.k * L.k
                h, w = L.in_hw
                if L.pad:  # zero-padded copy: the implicit-GEMM gather needs no bounds logic
                    assert not L.first and cur.ld == L.cin and cur.rows == n * h * w
                    h, w = h + 2 * L.pad, w + 2 * L.pad
                    xp = self._buf(f"{tag}{L.prefix}.xp", n * h * w, L.cin)
                    if L.pad_mode:  # reflect / replicate / circular borders: the general kernel (D = 1)
                        hip.pad_ndhwc(cur.ptr, n, (1, L.in_hw[0], L.in_hw[1], L.cin), (0, L.pad, L.pad), xp.ptr, L.pad_mode)
                    else:
                        hip.pad_nhwc(cur.ptr, n, L.in_hw[0], L.in_hw[1], L.cin, L.pad, xp.ptr)
                    cur = xp
                desc = hip.conv_desc(n, h, w, L.cin, L.k, L.k, L.stride, L.cout, L.act)
                # implicit GEMM (no patch matrix) whenever the geometry allows; explicit im2col otherwise
                implicit = not self.force_explicit_conv and hip.conv2d_supported(desc, L.first)
                if L.first and L.s2d:  # strided first layer on the space-to-depth'd observation (channels-last)
                    b = L.s2d
                    desc = hip.conv_desc(n, h // b, w // b, L.cin * b * b, L.k // b, L.k // b, 1, L.cout, L.act)
                    implicit = True
                    if self.force_explicit_conv or not hip.conv2d_supported(desc, 2):
                        raise hip.HipError("space-to-depth parameter layout needs the implicit convolution path "
                                           "(build the policy with SRL_EXPLICIT_CONV=1 to use the fallback)")
                y = self._buf(f"{tag}{L.prefix}.y", m, L.cout)
                if implicit and self._relu_masks and L.act == hip.ACT_RELU and L.cout % 32 == 0:
                    y = y._replace(mask=self.ws.get(f"{tag}{L.prefix}.mask", m * L.cout // 32, torch.int32).data_ptr())
                P = None if implicit else self._buf(f"{tag}{L.prefix}.P", m, kdim)
                saved = None
                if L.first:
                    c = pending_obs_ln.shape[0]
                    is_u8 = obs.dtype == torch.uint8
                    if not is_u8 and obs.dtype != torch.float32:
                        raise hip.HipError(f"image observation `{enc.key}` must be uint8 or float32, got {obs.dtype}")
                    gam, bet = self._p(f"{pending_obs_ln.prefix}.weight"), self._p(f"{pending_obs_ln.prefix}.bias")
                    if staged is not None and not (implicit and L.s2d and staged.layout == ("s2d", int(L.s2d))):
                        raise hip.HipError(f"observation `{enc.key}`: ring layout {staged.layout} does not fit this network")
                    row_index = None
                    if staged is not None and staged.span is None and hip.conv2d_obs_row_index_supported(desc, is_u8, True):
                        # the byte kernels read the ring's rows in place, through the sample's slot index: no pass over
                        # the frames besides the convolution's own
                        src, mean, rstd, row_index = staged.in_place()
                    elif staged is not None:
                        # no re-tiling pass and no statistics pass: both were done once, when the rollout uploaded the row
                        src, mean, rstd = staged.resolve(self.ws, f"{tag}{L.prefix}")
                    else:
                        mean = self.ws.get(f"{tag}{pending_obs_ln.prefix}.mean", n)
                        rstd = self.ws.get(f"{tag}{pending_obs_ln.prefix}.rstd", n)
                        src = obs
                        if L.s2d:
                            src = self.ws.get(f"{tag}{L.prefix}.s2d", n * c * h * w, dtype=obs.dtype)
                            hip.obs_space_to_depth(obs.data_ptr(), is_u8, n, c, h, w, L.s2d, src.data_ptr(), mean.data_ptr(),
                                                   rstd.data_ptr())
                        else:
                            hip.obs_ln_stats(obs.data_ptr(), is_u8, n, c * h * w, mean.data_ptr(), rstd.data_ptr())
                    if (h2 is not None and implicit and is_u8 and L.s2d and n >= self.H2_MIN_ROWS
                            and hip.conv2d_obs_row_index_supported(desc, is_u8, True)):
                        # the whole convolution stack and the Linear behind it on pre-split activations (h2path.py)
                        # (inference with the closing LayerNorm + heads as the consumer: the Linear's reduction may be split
                        # over workgroups, the consumer adds the slabs)
                        split = None
                        if lnheads is not None and len(enc.layers) == 6 and enc.layers[4] is h2.fc:
                            split = "infer" if self._infer else "train"
                        y2, saved2 = h2.forward(tag, staged, obs, n, is_u8, src, mean, rstd, row_index, split_fc=split)
                        tape.append(("h2cnn", h2, None, saved2, 0))
                        cur, cur_act, cur_range = y2, h2.fc.act, None
                        skip = 3
                        continue
                    y_range = self._act_range() if implicit else None
                    if implicit:
                        fws = self.ws.get(f"{L.prefix}.folded", hip.conv2d_obs_fwd_workspace(desc)).data_ptr()
                        hip.conv2d_obs_fwd(desc, src.data_ptr(), is_u8, mean.data_ptr(), rstd.data_ptr(), gam, bet,
                                           self._p(f"{L.prefix}.weight"), self._p(f"{L.prefix}.bias"), y.ptr,
                                           channels_last=bool(L.s2d), row_index=row_index, y_absmax=y_range, y_mask=y.mask,
                                           ws_ptr=fws, reuse_folded=self._derived_fresh(f"{L.prefix}.folded:{desc.n >= 64}:{desc.n >= 32}", fws))
                    else:
                        hip.im2col_obs_ln(obs.data_ptr(), is_u8, mean.data_ptr(), rstd.data_ptr(), gam, bet, n, c, h, w,
                                          L.k, L.k, L.stride, P.ptr)
                    saved = (src, is_u8, mean, rstd, pending_obs_ln, bool(L.s2d), row_index)
                else:
                    assert cur.ld == L.cin and cur.rows == n * h * w
                    small = implicit and self._conv_small(L, desc)
                    y_range = self._act_range() if implicit and not small else None
                    if small:
                        # 4 / 8 channels on both sides: a direct vector-unit kernel (csrc/conv_small.hip), not a matrix-core tile
                        hip.conv2d_small_fwd(desc, cur.ptr, self._p(f"{L.prefix}.weight"), self._p(f"{L.prefix}.bias"), y.ptr)
                    elif implicit:
                        w_range = self._weight_range(L.prefix, L.cout * kdim) if cur_range is not None and not L.pad else None
                        xr = cur_range if w_range is not None else None
                        w, pre = self._p(f"{L.prefix}.weight"), None
                        if hip.conv2d_fwd_two_piece(desc, xr, w_range):
                            pre = self._presplit(f"{L.prefix}.w2h", w, L.cout * kdim, w_range)
                        hip.conv2d_nhwc_fwd(desc, cur.ptr, pre or w, self._p(f"{L.prefix}.bias"),
                                            y.ptr, x_absmax=xr, w_absmax=w_range, y_absmax=y_range, y_mask=y.mask,
                                            presplit=pre is not None)
                    else:
                        hip.im2col_nhwc(cur.ptr, n, h, w, L.cin, L.k, L.k, L.stride, P.ptr)
                if not implicit:
                    hip.gemm(m, L.cout, kdim, P.ptr, kdim, 0, self._p(f"{L.prefix}.weight"), kdim, 0, y.ptr, y.ld,
                             bias=self._p(f"{L.prefix}.bias"), act=L.act)
                tape.append(("conv", L, cur, (P, saved, n, desc, cur_range), cur_act))
                cur, cur_act, cur_range = y, L.act, y_range
            else:  # pragma: no cover
                raise TypeError(L)
        return cur

    def _chain_bwd(self, records: list, dy: Buf, tag: str, need_input_grad=False) -> Optional[Buf]:
        """Walk tape records of one sequential chain backwards; dy is w.r.t. the chain's (pre-activation) output."""
        g = dy
        g_range = None  # device float bounding max |g| when its producer tracked it
        for idx in range(len(records) - 1, -1, -1):
            kind, L, x, saved, in_act = records[idx]
            need_dx = idx > 0 or need_input_grad
            if kind == "ln":
                # the layer below wants its gradient's range (a Linear whose input's range the forward pass tracked): the
                # LayerNorm's backward folds it in while it writes dx, instead of one more pass over dx
                below = records[idx - 1] if idx > 0 else None
                want = need_dx and below is not None and below[0] == "linear" and below[3] is not None
                dxr = self._grad_range() if want else None
                g = self._ln_bwd(L, x, saved, g, in_act, need_dx, tag, dx_range=dxr)
                g_range = dxr
            elif kind == "linear":
                x_range = saved  # the forward pass's range of this layer's input (a convolution's output), or None
                if x_range is not None and g_range is None and g.ld == g.cols:
                    g_range = self._grad_range(g)  # one pass over dz: its producer (a LayerNorm) does not track it
                dx_range = self._grad_range() if (x_range is not None and g_range is not None and need_dx) else None
                g = self._linear_bwd(L, x, g, in_act, need_dx, tag, x_range=x_range,
                                     dz_range=g_range if x_range is not None else None, dx_range=dx_range)
                g_range = dx_range
            elif kind == "fusedenc":
                self._fused_bwd(L, g.ptr, g.ld)   # (releases its layers' buckets itself)
                g, g_range = None, None
                continue
            elif kind == "lnheads":   # closing LayerNorm + heads: g is d loss / d first head's output, the second head's gradient
                # was left in self._lnheads_dv by backward()
                mean, rstd, heads = saved
                dys = [g] + ([self._lnheads_dv] if len(heads) > 1 else [])
                dx = self._buf(f"{tag}{L.prefix}.dx", x.rows, L.dim)
                # a pre-split block below wants max |dx|: tracked here while dx is written, instead of one more pass over it
                below = records[idx - 1] if idx > 0 else None
                amax = below[1].open_backward(below[3]) if (below is not None and below[0] == "h2cnn") else None
                hip.ln_heads_bwd(x.ptr, x.ld, x.rows, L.dim, self._p(f"{L.prefix}.weight"), self._p(f"{L.prefix}.bias"),
                                 mean.data_ptr(), rstd.data_ptr(), [self._p(f"{h.prefix}.weight") for h in heads],
                                 [h.out_features for h in heads], [d.ptr for d in dys], [d.ld for d in dys], in_act, dx.ptr, dx.ld,
                                 self._g(f"{L.prefix}.weight"), self._g(f"{L.prefix}.bias"),
                                 [self._g(f"{h.prefix}.weight") for h in heads], [self._g(f"{h.prefix}.bias") for h in heads],
                                 dx_absmax=amax)
                self._release([h.prefix for h in heads] + [L.prefix])
                g, g_range = dx, ("tracked" if amax is not None else None)
                continue
            elif kind == "fusedtail":   # what follows the recurrent layers, head included: g is d loss / d head output
                g, g_range = self._fused_bwd(L, g.ptr, g.ld), None
                continue
            elif kind == "h2cnn":
                L.backward(saved, g, dy_ranged=(g_range == "tracked"))
                g, g_range = None, None
                continue   # (the block released its layers one by one)
            elif kind == "gru":
                g = self._gru_bwd(L, saved, g, in_act, need_dx, tag)
                g_range = None
            elif kind == "obsln":
                is_u8, mean, rstd, n = saved
                c, h, w = L.shape
                assert g.ld == c and g.rows == n * h * w
                hip.obs_ln_nhwc_bwd(g.ptr, x.data_ptr(), is_u8, mean.data_ptr(), rstd.data_ptr(), n, c, h, w,
                                    self._g(f"{L.prefix}.weight"), self._g(f"{L.prefix}.bias"))
                g, g_range = None, None
            elif kind == "poolnd":
                n = saved
                dx = self._buf(f"{tag}{L.prefix}.dx", n * math.prod(L.in_sp), L.c)
                hip.maxpool_ndhwc_bwd(g.ptr, x.ptr, n, (*L.in_sp, L.c), L.win, in_act, dx.ptr)
                g, g_range = dx, None
            elif kind == "convnd":
                P, n, sp = saved
                m, kdim = g.rows, L.cin * math.prod(L.kern)
                assert g.ld == L.cout and m == n * math.prod(L.out_sp)
                self._wgrad(L.cout, kdim, m, g, P.ptr, kdim, self._g(f"{L.prefix}.weight"))
                hip.colsum(g.ptr, g.ld, m, L.cout, self._g(f"{L.prefix}.bias"), accumulate=True)
                hip.gemm(m, kdim, L.cout, g.ptr, g.ld, 0, self._p(f"{L.prefix}.weight"), kdim, 1, P.ptr, kdim)  # dP over P
                dx = self._buf(f"{tag}{L.prefix}.dx", n * math.prod(sp), L.cin)
                hip.col2im_ndhwc(P.ptr, n, (*sp, L.cin), L.kern, L.stride, x.ptr if in_act else None, in_act, dx.ptr)
                if any(L.pads):
                    dxc = self._buf(f"{tag}{L.prefix}.dxc", n * math.prod(L.in_sp), L.cin)
                    hip.crop_ndhwc(dx.ptr, n, (*L.in_sp, L.cin), L.pads, dxc.ptr, L.pad_mode)
                    dx = dxc
                g, g_range = dx, None
            elif kind == "pool":
                n = saved
                (h, w), (ph, pw) = L.in_hw, L.out_hw
                assert g.ld == L.c and g.rows == n * ph * pw
                dx = self._buf(f"{tag}{L.prefix}.dx", n * h * w, L.c)
                hip.maxpool2_nhwc_bwd(g.ptr, x.ptr, n, h, w, L.c, in_act, dx.ptr)
                g, g_range = dx, None
            elif kind == "conv":
                P, first_saved, n, desc, x_range = saved
                kdim = L.cin * L.k * L.k
                m = g.rows
                if P is None:  # implicit-GEMM path
                    assert g.ld == L.cout
                    gw, gb, wp = self._g(f"{L.prefix}.weight"), self._g(f"{L.prefix}.bias"), self._p(f"{L.prefix}.weight")
                    if L.first:
                        obs, is_u8, mean, rstd, lnspec, chlast, row_index = first_saved
                        blk = self.__dict__.get("_h2_blocks", {}).get(lnspec.prefix)
                        wsz = hip.conv2d_obs_bwd_workspace(desc)
                        if blk is not None and blk.open:
                            # a ragged last chunk below H2_MIN_ROWS: the executor's earlier chunks left their position sums open
                            # in the pre-split block's workspace -- this chunk adds to them and closes (h2path.first_layer_bwd)
                            blk.first_layer_bwd(n, (obs, is_u8, mean, rstd, row_index), g.ptr, self._grad_range(g))
                        else:
                            hip.conv2d_obs_bwd(desc, obs.data_ptr(), is_u8, mean.data_ptr(), rstd.data_ptr(),
                                               self._p(f"{lnspec.prefix}.weight"), self._p(f"{lnspec.prefix}.bias"), wp, g.ptr,
                                               gw, gb, self._g(f"{lnspec.prefix}.weight"), self._g(f"{lnspec.prefix}.bias"),
                                               self.ws.get("conv_obs_bwd", wsz).data_ptr(), channels_last=chlast, row_index=row_index)
                        g = None
                    elif self._conv_small(L, desc):
                        wws = self.ws.get("conv_small_wgrad", hip.conv2d_small_wgrad_workspace(desc)).data_ptr()
                        self._on_side(lambda d=desc, xp=x.ptr, gp=g.ptr: hip.conv2d_small_wgrad(d, xp, gp, wws, gw, gb))
                        h, w = L.in_hw
                        dx = self._buf(f"{tag}{L.prefix}.dx", n * h * w, L.cin)
                        hip.conv2d_small_dgrad(desc, g.ptr, wp, x.ptr if in_act else None, in_act, dx.ptr)
                        g, g_range = dx, None
                    else:
                        wsz = hip.conv2d_wgrad_workspace(desc)
                        if g_range is None and not L.pad and g.ld == g.cols:
                            g_range = self._grad_range(g)
                        two = g_range is not None and not L.pad
                        wgrad_ws = self.ws.get("conv_wgrad", wsz).data_ptr()
                        self._on_side(lambda d=desc, xp=x.ptr, gp=g.ptr, xr=x_range if two else None,
                                      gr=g_range if two and x_range is not None else None: hip.conv2d_nhwc_wgrad(
                                          d, xp, gp, gw, wgrad_ws, gb, x_absmax=xr, dz_absmax=gr))
                        wt = self.ws.get(f"{L.prefix}.wt", hip.conv2d_dgrad_weight_elems(desc))
                        if not self._derived_fresh(f"{L.prefix}.wt", wt.data_ptr()):  # once per update, not per chunk
                            hip.conv2d_dgrad_repack(desc, wp, wt.data_ptr())
                        h, w = L.in_hw[0] + 2 * L.pad, L.in_hw[1] + 2 * L.pad
                        dx = self._buf(f"{tag}{L.prefix}.dx", n * h * w, L.cin)
                        dx_range = self._grad_range() if two else None
                        x_mask = x.mask if in_act == hip.ACT_RELU else None  # the derivative from sign bits, not floats
                        wr = self._weight_range(L.prefix, L.cout * kdim) if two else None
                        wtp, pre = wt.data_ptr(), None
                        if hip.conv2d_dgrad_two_piece(desc, g_range if two else None, wr):
                            pre = self._presplit(f"{L.prefix}.wt2h", wtp, hip.conv2d_dgrad_weight_elems(desc), wr)
                        hip.conv2d_nhwc_dgrad(desc, g.ptr, pre or wtp, x.ptr if in_act and x_mask is None else None, in_act,
                                              dx.ptr, dz_absmax=g_range if two else None, w_absmax=wr,
                                              dx_absmax=dx_range, x_mask=x_mask, presplit=pre is not None)
                        g = self._crop(L, dx, n, tag)
                        g_range = dx_range
                    if g is not None and idx > 0:
                        prev_out_cols = self._out_cols(records[idx - 1])
                        if g.cols != prev_out_cols:
                            g = Buf(g.ptr, prev_out_cols, g.rows * g.cols // prev_out_cols, prev_out_cols)
                    self._notify_ready(kind, L, saved)
                    continue
                self._wgrad(L.cout, kdim, m, g, P.ptr, kdim, self._g(f"{L.prefix}.weight"))
                hip.colsum(g.ptr, g.ld, m, L.cout, self._g(f"{L.prefix}.bias"), accumulate=True)
                # dP = dZ W, written over the patch matrix (its last reader was the weight gradient above)
                hip.gemm(m, kdim, L.cout, g.ptr, g.ld, 0, self._p(f"{L.prefix}.weight"), kdim, 1, P.ptr, kdim)
                if L.first:
                    obs, is_u8, mean, rstd, lnspec, _, _ = first_saved
                    c, h, w = lnspec.shape
                    hip.obs_ln_affine_bwd(P.ptr, obs.data_ptr(), is_u8, mean.data_ptr(), rstd.data_ptr(), n, c, h, w, L.k,
                                          L.k, L.stride, self._g(f"{lnspec.prefix}.weight"),
                                          self._g(f"{lnspec.prefix}.bias"))
                    g = None
                else:
                    h, w = L.in_hw[0] + 2 * L.pad, L.in_hw[1] + 2 * L.pad
                    dx = self._buf(f"{tag}{L.prefix}.dx", n * h * w, L.cin)
                    hip.col2im_nhwc(P.ptr, n, h, w, L.cin, L.k, L.k, L.stride, x.ptr if in_act else None, in_act, dx.ptr)
                    g, g_range = self._crop(L, dx, n, tag), None
            self._notify_ready(kind, L, saved)
            if g is not None and idx > 0:
                # a Flatten between this record's input and the previous record's output: reshape the gradient
                prev_out_cols = self._out_cols(records[idx - 1])
                if g.cols != prev_out_cols:
                    assert g.ld == g.cols and (g.rows * g.cols) % prev_out_cols == 0
                    g = Buf(g.ptr, prev_out_cols, g.rows * g.cols // prev_out_cols, prev_out_cols)
        return g

    def _crop(self, L, dx: Buf, n: int, tag: str) -> Buf:
        """Gradient of a zero-padded input: drop the border."""
        if not L.pad:
            return dx
        h, w = L.in_hw
        out = self._buf(f"{tag}{L.prefix}.dxc", n * h * w, L.cin)
        if L.pad_mode:
            hip.crop_ndhwc(dx.ptr, n, (1, h, w, L.cin), (0, L.pad, L.pad), out.ptr, L.pad_mode)
        else:
            hip.crop_nhwc(dx.ptr, n, h, w, L.cin, L.pad, out.ptr)
        return out

    def _notify_ready(self, kind, L, saved):
        if self.grad_ready_hook is None or kind in ("pool", "poolnd"):
            return
        done = [L.prefix]
        if kind == "conv" and L.first and saved[1] is not None:
            done.append(saved[1][4].prefix)  # the observation LayerNorm fused into the first convolution
        self._release(done)

    @staticmethod
    def _out_cols(record):
        kind, L = record[0], record[1]
        if kind == "ln":
            return L.dim
        if kind == "h2cnn":
            return L.H
        if kind in ("fusedenc", "fusedtail"):
            return L["feat"].cols
        if kind == "lnheads":
            return L.dim
        if kind == "linear":
            return L.out_features
        if kind == "gru":
            return L.hidden
        if kind in ("pool", "poolnd"):
            return L.c
        if kind == "obsln":
            return L.shape[0]
        return L.cout

    def _lnheads_ok(self, encoders, backbone, heads, n) -> bool:
        """The encoder's closing LayerNorm and the heads right behind it (no backbone layers between) as one launch."""
        if not self._lnheads or backbone or len(encoders) != 1 or n > self.encoder_rows or not heads or n < 64:
            return False
        last = encoders[0].layers[-1] if encoders[0].layers else None
        if not isinstance(last, ns.LayerNormSpec) or len(encoders[0].layers) < 2 or any(h.in_features != last.dim or h.act for h in heads):
            return False
        return hip.ln_heads_supported(last.dim, [h.out_features for h in heads])

    def _trunk_fwd(self, tag, encoders, backbone, obs: Dict[str, torch.Tensor], n: int, head=None, head_out=None, lnheads=None):
        """``head`` / ``head_out``: the head behind this trunk and the tensor its output goes to -- when the layers that follow the
        last recurrent layer are LayerNorm / Linear no wider than 64, they and the head run as ONE launch per direction
        (`fusedtail`; the tape's last field says whether the head went in)."""
        for enc in encoders:
            if enc.key not in obs:
                raise KeyError(f"observation key `{enc.key}` missing from the sample (has {list(obs)})")
        # the encoders see independent rows: more rows than one launch may address (recurrent nets hand over every
        # row of the sample at once) go through in pieces, each piece with its own tape; the features of all pieces
        # are gathered into one [n, width] block for the backbone, which walks the time axis
        step = self.encoder_rows
        pieces = [(r0, min(n, r0 + step)) for r0 in range(0, n, step)]
        widths = [enc.out_dim for enc in encoders]
        width = sum(widths)
        enc_tapes = []
        feat = None
        for pi, (r0, r1) in enumerate(pieces):
            ptag = tag if len(pieces) == 1 else f"{tag}piece{pi}:"
            tapes, outs = [], []
            for enc in encoders:
                tape = []
                outs.append(self._encoder_fwd(enc, obs[enc.key][r0:r1], r1 - r0, ptag, tape, lnheads=lnheads))
                tapes.append(tape)
            enc_tapes.append(tapes)
            if lnheads is not None and tapes[0] and tapes[0][-1][0] == "lnheads":   # (one piece, one encoder: _lnheads_ok)
                return None, 0, (enc_tapes, [], widths, pieces, "lnheads")
            if len(pieces) == 1 and len(outs) == 1:
                feat = outs[0]
                break
            if feat is None:
                feat = self._buf(f"{tag}concat", n, width)
            col = 0
            for o in outs:
                hip.copy2d(o.ptr, o.ld, feat.ptr + 4 * (r0 * width + col), width, r1 - r0, o.cols)
                col += o.cols
        bb_tape = []
        cur, cur_act = feat, 0
        last_rnn = max((i for i, L in enumerate(backbone) if isinstance(L, ns.GruSpec)), default=-1)
        for li, L in enumerate(backbone):
            if (head is not None and li == last_rnn + 1 and last_rnn >= 0 and self._mlp_fused and self._enc_fused and n >= 512
                    and cur_act == 0 and all(isinstance(M, (ns.LayerNormSpec, ns.LinearSpec)) for M in backbone[li:])):
                rec = self._fused_layers_fwd((tag, "tail"), f"{tag}tail.", list(backbone[li:]) + [head], cur, n, out=head_out, head=True,
                                             need_dx=True)
                if rec is not None:
                    bb_tape.append(("fusedtail", rec, None, None, 0))
                    return rec["feat"], 0, (enc_tapes, bb_tape, widths, pieces, True)
            if isinstance(L, ns.LinearSpec):
                y = self._linear_fwd(L, cur, tag)
                bb_tape.append(("linear", L, cur, None, cur_act))
                cur, cur_act = y, L.act
            elif isinstance(L, ns.GruSpec):
                y, saved, last = self._gru_fwd(L, cur, tag)
                bb_tape.append(("gru", L, cur, saved, cur_act))
                self.last_state[tag] = last
                cur, cur_act = y, 0
            else:
                y, saved = self._ln_fwd(L, cur, tag)
                bb_tape.append(("ln", L, cur, saved, cur_act))
                cur, cur_act = y, 0
        return cur, cur_act, (enc_tapes, bb_tape, widths, pieces, False)

    def _trunk_bwd(self, tag, trunk_tape, dfeat: Buf):
        enc_tapes, bb_tape, widths, pieces, _ = trunk_tape
        g = self._chain_bwd(bb_tape, dfeat, tag, need_input_grad=True) if bb_tape else dfeat
        hook = self.grad_ready_hook
        for pi, (r0, r1) in enumerate(pieces):
            # encoder gradients accumulate over the pieces: they are final (and may be reduced) after the last one
            self.grad_ready_hook = hook if pi == len(pieces) - 1 else None
            col = 0
            for tape, wdt in zip(enc_tapes[pi], widths):
                sub = Buf(g.ptr + 4 * (r0 * g.ld + col), g.ld, r1 - r0, wdt)
                self._chain_bwd(tape, sub, tag, need_input_grad=False)
                col += wdt
            if len(pieces) > 1:
                self._join_side()  # the pieces share their backward buffers: the next one must not overtake this one's
                # weight gradients on the second stream
        self.grad_ready_hook = hook

    # ------------------------------------------------------------------ small MLP chains in one launch per direction
    def _fused_fwd(self, tag, encoders, backbone, head, obs, n: int, out: Optional[torch.Tensor] = None):
        """The trunk (one vector-observation encoder + backbone) -- and, when ``head`` is given, the head behind it -- as ONE
        launch if every layer is a LayerNorm or a Linear no wider than 128 (``hip.mlp_fwd``).  Returns a record for
        ``_fused_bwd`` (``feat`` = the chain's output as a Buf, ``act`` = the activation that produced it) or None."""
        if not self._mlp_fused or self._rnn is not None or self.spec.num_rnn_layers or len(encoders) != 1:
            return None
        enc = encoders[0]
        x = obs.get(enc.key) if isinstance(obs, dict) else None
        layers = list(enc.layers) + list(backbone) + ([head] if head is not None else [])
        if isinstance(x, RingObs) and x.layout[0] != "s2d" and x.rows == n:  # raw vector rows kept in the HBM observation ring
            x = x.gather_raw(self.ws, f"{tag}{enc.key}.ring")
        return self._fused_layers_fwd((tag, head is not None), tag, layers, x, n, out, head is not None)

    def _fused_layers_fwd(self, key, tag, layers, x, n: int, out: Optional[torch.Tensor] = None, head=False, need_dx=False):
        """``layers`` (LayerNorm / Linear, no wider than 128) on the float32 rows ``x`` [n, in] (a tensor, or a dense Buf of the
        workspace) as one launch, or None.  ``need_dx``: the backward pass must also return d loss / d x (the chain sits behind
        other layers) -- only chains that keep no tape can."""
        if isinstance(x, Buf):
            if x.rows != n or x.ld != x.cols or not layers or len(layers) > hip.MLP_MAX_LAYERS:
                return None
            xptr, xcols = x.ptr, x.cols
        else:
            if (not isinstance(x, torch.Tensor) or x.dtype != torch.float32 or x.dim() != 2 or x.shape[0] != n or
                    not x.is_contiguous() or not layers or len(layers) > hip.MLP_MAX_LAYERS):
                return None
            xptr, xcols = x.data_ptr(), x.shape[1]
        for L in layers:
            if isinstance(L, ns.LayerNormSpec):
                ok = L.dim <= hip.MLP_MAX_WIDTH
            elif isinstance(L, ns.LinearSpec):
                ok = max(L.in_features, L.out_features) <= hip.MLP_MAX_WIDTH
            else:
                ok = False
            if not ok:
                return None
        key = (*key, self.flat.data_ptr(), self.grad.data_ptr())
        ent = self._mlp_cache.get(key)
        if ent is None:
            desc = []
            for L in layers:
                w, b = f"{L.prefix}.weight", f"{L.prefix}.bias"
                if isinstance(L, ns.LayerNormSpec):
                    desc.append((0, L.dim, L.dim, 0, self._p(w), self._p(b), self._g(w), self._g(b)))
                else:
                    desc.append((1, L.in_features, L.out_features, L.act, self._p(w), self._p(b), self._g(w), self._g(b)))
            arr = hip.mlp_layers(desc)
            tld = hip.mlp_tape_floats(arr)
            if tld < 0 or desc[0][1] != xcols:
                return None
            last = layers[-1]
            ent = self._mlp_cache[key] = (arr, tld, last.out_features if isinstance(last, ns.LinearSpec) else last.dim,
                                          last.act if isinstance(last, ns.LinearSpec) else 0, hip.mlp_bwd_max_rows(arr))
        arr, tld, width, act, max_rows = ent
        if n > max_rows:
            return None
        # (the matrix-core chain keeps no tape: its backward pass walks forward again from x -- hip.mlp_tape_floats_at is 0 there)
        taped = hip.mlp_tape_floats_at(arr, n) != 0
        if need_dx and taped:
            return None
        tape = self.ws.get(f"{tag}mlp.tape", n * tld) if taped else None
        y = out if out is not None else self.ws.get(f"{tag}mlp.y", n * width)
        hip.mlp_fwd(arr, xptr, xcols, n, tape.data_ptr() if tape is not None else 0, tld, y.data_ptr(), width)
        return dict(arr=arr, x=x, xptr=xptr, xcols=xcols, tape=tape, tld=tld, n=n, feat=Buf(y.data_ptr(), width, n, width), act=act,
                    head=head, prefixes=[L.prefix for L in layers], need_dx=need_dx, tag=tag)

    def _fused_bwd(self, rec, dy_ptr: int, lddy: int) -> Optional[Buf]:
        dx = None
        if rec["need_dx"]:
            dx = self._buf(f"{rec['tag']}mlp.dx", rec["n"], rec["xcols"])
            hip.mlp_bwd_dx(rec["arr"], rec["xptr"], rec["xcols"], rec["n"], dy_ptr, lddy, dx.ptr, dx.ld)
        else:
            hip.mlp_bwd(rec["arr"], rec["xptr"], rec["xcols"], rec["n"], rec["tape"].data_ptr() if rec["tape"] is not None else 0,
                        rec["tld"], dy_ptr, lddy)
        self._release(rec["prefixes"])  # one launch: every layer of the chain is final behind it
        return dx

    @staticmethod
    def _head_in(tape):
        """Whether the tower's head ran inside its trunk's last launch: a fused chain, a trunk ending in a `fusedtail` (True), or
        an encoder closing in `lnheads` ("lnheads": with a shared backbone BOTH heads ran there).  Falsy otherwise."""
        if isinstance(tape, dict):
            return bool(tape["head"])
        return tape[4] if (tape is not None and len(tape) > 4) else False

    # ------------------------------------------------------------------ public: forward / backward
    def forward(self, obs: Dict[str, torch.Tensor], n: int, keep_tape: bool = True, rnn: Optional[RnnCtx] = None):
        """obs leaves [n, ...] on the device.  Returns (logits [n, sum(A)], value [n, value_dim]) tensors
        (views of workspace buffers, valid until the next forward).  Recurrent nets: ``rnn`` describes the time
        structure of the rows; the final hidden states are left in ``self.last_state`` ("a:" / "c:")."""
        hip.require_gpu()
        sp = self.spec
        self._rnn = rnn
        self._infer = not keep_tape   # nothing of this pass is kept for a backward pass
        self._amax_next = -1
        self.last_state = {}
        if sp.num_rnn_layers and (rnn is None or rnn.T * rnn.B != n):
            raise hip.HipError("recurrent backbone: `rnn` context missing or inconsistent with the row count")
        atot = sum(sp.act_dims)
        logits_t = self.ws.get("logits", n * atot)
        value_t = self.ws.get("value", n * sp.value_dim)
        # Recurrent nets over vector observations, more than one chunk: the recurrent layers want the rows chunk-major (step c of
        # every chunk side by side), everything else is row-wise.  Instead of re-ordering the 64-wide features in front of and
        # behind every recurrent layer, in both directions (8 launches, 0.45 ms of the SMAC-sized step), the OBSERVATIONS are put
        # into chunk-major order once, the whole pass runs on them, and only the heads' outputs (and their gradients on the way
        # back) are put back into time-major order: a fifth of the bytes.
        encs = list(sp.obs_encoders) + ([] if sp.shared_backbone else list(sp.state_encoders))
        cm = bool(self._cm_enabled and sp.num_rnn_layers and rnn is not None and rnn.T // rnn.C > 1
                  and sp.std_type != "shared_learnable" and sp.aux_head is None
                  and all(isinstance(obs.get(e.key), torch.Tensor) and obs[e.key].dim() == 2 and obs[e.key].dtype == torch.float32
                          and obs[e.key].is_contiguous() for e in encs))
        self._cm = cm
        logits_out, value_out = logits_t, value_t
        if cm:
            obs = dict(obs)
            for key in {e.key for e in encs}:
                x = obs[key]
                xc = self.ws.get(f"cm.obs.{key}", x.numel())[:x.numel()].view(x.shape)
                hip.chunk_rows(x.data_ptr(), xc.data_ptr(), rnn.T, rnn.B, rnn.C, x.shape[1])
                obs[key] = xc
            logits_t = self.ws.get("cm.logits", n * atot)
            value_t = self.ws.get("cm.value", n * sp.value_dim)
        # CartPole-sized nets: trunk + head of a separate actor / critic as one launch each (trunk only when the heads share it)
        # (not with PPG's auxiliary value head: it reads the actor trunk's features beside the actor head)
        heads_in = not sp.shared_backbone and sp.std_type != "shared_learnable" and sp.aux_head is None
        fa = self._fused_fwd("a:", sp.obs_encoders, sp.actor_backbone, sp.actor_head if heads_in else None, obs, n,
                             out=logits_t if heads_in else None)
        if fa is not None:
            a_feat, a_act, a_tape = fa["feat"], fa["act"], fa
        else:
            lnh = None
            if sp.std_type != "shared_learnable" and sp.aux_head is None and not sp.num_rnn_layers:
                hs = [sp.actor_head, sp.critic_head] if sp.shared_backbone else [sp.actor_head]
                if self._lnheads_ok(sp.obs_encoders, sp.actor_backbone, hs, n):
                    lnh = (hs, [logits_t, value_t][:len(hs)])
            a_feat, a_act, a_tape = self._trunk_fwd("a:", sp.obs_encoders, sp.actor_backbone, obs, n,
                                                    head=sp.actor_head if heads_in else None, head_out=logits_t, lnheads=lnh)
        if sp.shared_backbone:
            c_feat, c_act, c_tape = a_feat, a_act, None
        else:
            fc = self._fused_fwd("c:", sp.state_encoders, sp.critic_backbone, sp.critic_head if heads_in else None, obs, n,
                                 out=value_t if heads_in else None)
            if fc is not None:
                c_feat, c_act, c_tape = fc["feat"], fc["act"], fc
            else:
                lnh = None
                if sp.aux_head is None and not sp.num_rnn_layers and self._lnheads_ok(sp.state_encoders, sp.critic_backbone, [sp.critic_head], n):
                    lnh = ([sp.critic_head], [value_t])
                c_feat, c_act, c_tape = self._trunk_fwd("c:", sp.state_encoders, sp.critic_backbone, obs, n,
                                                        head=sp.critic_head if heads_in else None, head_out=value_t, lnheads=lnh)
        if not self._head_in(a_tape):
            hip.gemm(n, atot, sp.hidden_dim, a_feat.ptr, a_feat.ld, 0, self._p(f"{sp.actor_head.prefix}.weight"), sp.hidden_dim, 0,
                     logits_t.data_ptr(), atot, bias=self._p(f"{sp.actor_head.prefix}.bias"))
        if not self._head_in(c_tape) and not (sp.shared_backbone and self._head_in(a_tape) == "lnheads"):
            hip.gemm(n, sp.value_dim, sp.hidden_dim, c_feat.ptr, c_feat.ld, 0, self._p(f"{sp.critic_head.prefix}.weight"),
                     sp.hidden_dim, 0, value_t.data_ptr(), sp.value_dim, bias=self._p(f"{sp.critic_head.prefix}.bias"))
        self.log_std_rows = None
        if sp.std_type == "shared_learnable":  # log sigma from a second head on the actor features (:93, :131-132)
            ls_t = self.ws.get("log_std_rows", n * atot)
            hip.gemm(n, atot, sp.hidden_dim, a_feat.ptr, a_feat.ld, 0, self._p("log_std.weight"), sp.hidden_dim, 0,
                     ls_t.data_ptr(), atot, bias=self._p("log_std.bias"))
            self.log_std_rows = ls_t[:n * atot].view(n, atot)
        self.aux_value = None
        if sp.aux_head is not None:  # PPG: a second value estimate from the ACTOR's features (actor_critic_policy.py:139-140)
            aux_t = self.ws.get("aux_value", n * sp.value_dim)
            hip.gemm(n, sp.value_dim, sp.hidden_dim, a_feat.ptr, a_feat.ld, 0, self._p(f"{sp.aux_head.prefix}.weight"), sp.hidden_dim, 0,
                     aux_t.data_ptr(), sp.value_dim, bias=self._p(f"{sp.aux_head.prefix}.bias"))
            self.aux_value = aux_t[:n * sp.value_dim].view(n, sp.value_dim)
        self._tape = (n, a_feat, a_act, a_tape, c_feat, c_act, c_tape, cm, rnn) if keep_tape else None
        if cm:  # the heads' outputs back into time-major order
            hip.chunk_rows(logits_t.data_ptr(), logits_out.data_ptr(), rnn.T, rnn.B, rnn.C, atot, inverse=True)
            hip.chunk_rows(value_t.data_ptr(), value_out.data_ptr(), rnn.T, rnn.B, rnn.C, sp.value_dim, inverse=True)
            logits_t, value_t = logits_out, value_out
        return logits_t[:n * atot].view(n, atot), value_t[:n * sp.value_dim].view(n, sp.value_dim)

    def backward(self, d_logits: torch.Tensor, d_value: torch.Tensor, d_log_std_rows: Optional[torch.Tensor] = None,
                 d_aux: Optional[torch.Tensor] = None):
        """Accumulate d loss / d parameters into ``self.grad`` given d loss / d logits and d loss / d value (and, with a
        `shared_learnable` Gaussian head, d loss / d log sigma per row; with PPG's auxiliary head, d loss / d auxiliary value
        -- None in the PPO phase, where that head takes no gradient: phasic_policy_gradient.py:177-180)."""
        if self._tape is None:
            raise hip.HipError("backward() without a preceding forward(keep_tape=True)")
        sp = self.spec
        self._gmax_next = -1
        # (row order and time structure travel with the tape.  They do NOT make an interleaved pass safe: a forward(keep_tape=False)
        # between a taped forward and this call drops the tape -- the branch above raises -- and would have overwritten the shared
        # workspace buffers the tape points at (logits, value, a_feat, mlp.y); inference beside training uses an executor of its own)
        n, a_feat, a_act, a_tape, c_feat, c_act, c_tape, self._cm, self._rnn = self._tape
        atot = sum(sp.act_dims)
        dl = Buf(d_logits.data_ptr(), atot, n, atot)
        dv = Buf(d_value.data_ptr(), sp.value_dim, n, sp.value_dim)
        if self._cm:  # the pass ran on chunk-major rows (forward()): so do the gradients
            rnn = self._rnn
            dlc, dvc = self._buf("cm.d_logits", n, atot), self._buf("cm.d_value", n, sp.value_dim)
            hip.chunk_rows(dl.ptr, dlc.ptr, rnn.T, rnn.B, rnn.C, atot)
            hip.chunk_rows(dv.ptr, dvc.ptr, rnn.T, rnn.B, rnn.C, sp.value_dim)
            dl, dv = dlc, dvc
        a_head_in, c_head_in = self._head_in(a_tape), self._head_in(c_tape)
        da = None if a_head_in else self._linear_bwd(sp.actor_head, a_feat, dl, a_act, True, "a:")
        if sp.std_type == "shared_learnable":
            dls = Buf(d_log_std_rows.data_ptr(), atot, n, atot)
            self._linear_bwd(ns.LinearSpec("log_std", sp.hidden_dim, atot, 0), a_feat, dls, a_act, True, "a:", dx_into=da,
                             dx_accumulate=True)
            self._release(["log_std"])
        if sp.aux_head is not None:
            if d_aux is not None:
                dax = Buf(d_aux.data_ptr(), sp.value_dim, n, sp.value_dim)
                self._linear_bwd(sp.aux_head, a_feat, dax, a_act, True, "a:", dx_into=da, dx_accumulate=True)
            self._release([sp.aux_head.prefix])
        if sp.shared_backbone and not a_head_in:
            self._release([sp.actor_head.prefix])

        def trunk_bwd(tag, tape, dfeat, dhead):  # fused chains: one launch, all of its layers released behind it
            if isinstance(tape, dict):
                self._fused_bwd(tape, dhead.ptr if tape["head"] else dfeat.ptr, dhead.ld if tape["head"] else dfeat.ld)
            else:  # (a trunk whose last record took the head in starts from the head's gradient)
                self._trunk_bwd(tag, tape, dhead if tape[4] else dfeat)

        if sp.shared_backbone and a_head_in == "lnheads":   # both heads' gradients go into the closing launch
            self._lnheads_dv = dv
            trunk_bwd("a:", a_tape, da, dl)
        elif sp.shared_backbone:
            self._linear_bwd(sp.critic_head, c_feat, dv, c_act, True, "a:", dx_into=da, dx_accumulate=True)
            self._release([sp.critic_head.prefix])
            trunk_bwd("a:", a_tape, da, dl)
        else:
            dc = None if c_head_in else self._linear_bwd(sp.critic_head, c_feat, dv, c_act, True, "c:")
            rel = [h.prefix for h, inside in ((sp.actor_head, a_head_in), (sp.critic_head, c_head_in)) if not inside]
            if rel:   # (a head that ran inside its trunk's closing launch is released there)
                self._release(rel)
            trunk_bwd("a:", a_tape, da, dl)
            trunk_bwd("c:", c_tape, dc, dv)
        self._join_side()
        self._tape = None
