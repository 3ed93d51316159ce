"""Architecture description + parameter table + initialisation of the actor-critic network.

The network family is the reference's ``ActorCriticSeparate`` (``actor_critic_policy.py:28-143``):
per observation key ``LayerNorm -> (Linear | Conv2d stack + Flatten + Linear) -> ReLU -> LayerNorm``
(``policies/utils.py:33-63``, ``modules/cnn.py:39-135``), concatenation, ``dense_layers x (Linear -> act
[-> LayerNorm])`` (``modules/recurrent_backbone.py:40``, ``modules/utils.py:154-161``), and linear actor /
critic heads.  Parameter *names and reference shapes* are exactly the reference's ``state_dict`` keys,
so checkpoints interchange; the device copy may use a different *internal layout* per parameter
(``ParamInfo.layout``), converted at the checkpoint boundary only:

* ``conv_nhwc``:   Conv2d weight ``[Cout, Cin, KH, KW]`` stored ``[Cout, KH, KW, Cin]`` (activations are NHWC
  between convolutions so that GEMM outputs need no transposition);
* ``fc_from_chw``: first Linear after Flatten, ``[out, C*H*W]`` stored ``[out, H*W*C]`` for the same reason.

Initialisation replays the reference's construction order call for call (default ``nn.Linear`` /
``nn.Conv2d`` reset, then the orthogonal re-initialisations) on CPU tensors, so the same ``seed`` gives
bit-identical initial weights (checked against the reference in ``tests/golden/gen_golden.py``).
"""
import dataclasses
import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple, Union

import torch

ACTS = {"relu": 1, "tanh": 2}


@dataclasses.dataclass
class ParamInfo:
    name: str
    ref_shape: Tuple[int, ...]
    offset: int = 0  # in floats, into the flat buffer
    layout: str = "plain"  # plain | conv_nhwc | fc_from_chw | conv_s2d | ln_s2d
    chw: Optional[Tuple[int, int, int]] = None  # for fc_from_chw
    s2d: int = 0  # block size of the space-to-depth layouts
    ref_name: str = ""  # the reference's state_dict key when it differs from ``name`` (PopArt head)

    @property
    def key(self):
        return self.ref_name or self.name

    @property
    def numel(self):
        return int(math.prod(self.ref_shape))

    def to_internal(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach().to(torch.float32).reshape(self.ref_shape)
        if self.layout == "conv_nhwc":  # [Cout, Cin, *kernel] -> [Cout, *kernel, Cin] (any number of spatial dimensions)
            t = t.permute(0, *range(2, t.dim()), 1)
        elif self.layout == "fc_from_chw":
            c, h, w = self.chw
            t = t.reshape(self.ref_shape[0], c, h, w).permute(0, 2, 3, 1)
        elif self.layout == "conv_s2d":  # [Cout, Cin, KH, KW] -> [Cout, jh, jw, (ci, ph, pw)], kh = jh*s + ph
            co, ci, kh, kw = self.ref_shape
            b = self.s2d
            t = t.reshape(co, ci, kh // b, b, kw // b, b).permute(0, 2, 4, 1, 3, 5)
        elif self.layout == "ln_s2d":  # [C, H, W] -> [H/s, W/s, (c, ph, pw)]
            c, h, w = self.ref_shape
            b = self.s2d
            t = t.reshape(c, h // b, b, w // b, b).permute(1, 3, 0, 2, 4)
        return t.contiguous().reshape(-1)

    def to_reference(self, flat: torch.Tensor) -> torch.Tensor:
        if self.layout == "conv_nhwc":
            co, ci, *kern = self.ref_shape
            nd = len(kern)
            return flat.reshape(co, *kern, ci).permute(0, nd + 1, *range(1, nd + 1)).contiguous()
        if self.layout == "fc_from_chw":
            c, h, w = self.chw
            return flat.reshape(self.ref_shape[0], h, w, c).permute(0, 3, 1, 2).reshape(self.ref_shape).contiguous()
        if self.layout == "conv_s2d":
            co, ci, kh, kw = self.ref_shape
            b = self.s2d
            return flat.reshape(co, kh // b, kw // b, ci, b, b).permute(0, 3, 1, 4, 2, 5).reshape(self.ref_shape).contiguous()
        if self.layout == "ln_s2d":
            c, h, w = self.ref_shape
            b = self.s2d
            return flat.reshape(h // b, w // b, c, b, b).permute(2, 0, 3, 1, 4).reshape(self.ref_shape).contiguous()
        return flat.reshape(self.ref_shape).clone()


# ---------------------------------------------------------------------------------------------------
# layer descriptors (pure data; the device executor lives in hipnet.py)
@dataclasses.dataclass
class LayerNormSpec:
    prefix: str  # parameter prefix: <prefix>.weight / <prefix>.bias
    dim: int


@dataclasses.dataclass
class LinearSpec:
    prefix: str
    in_features: int
    out_features: int
    act: int  # 0 none | 1 relu | 2 tanh


@dataclasses.dataclass
class GruSpec:
    """AutoResetRNN(GRU) of a RecurrentBackbone (recurrent_backbone.py:50-58, autoreset_rnn.py:42-66): parameters
    ``<prefix>.weight_ih_l{l} [3H, H]``, ``weight_hh_l{l} [3H, H]``, ``bias_ih_l{l}``, ``bias_hh_l{l} [3H]`` (gate order
    r, z, n as in torch.nn.GRU); the hidden state is zeroed at every step whose on_reset flag is set."""
    prefix: str  # "<backbone>.rnn._AutoResetRNN__net"
    hidden: int
    layers: int
    kind: str = "gru"  # "lstm": gates i|f|g|o ([4H, H] weights); the stored state is cat(h, c) (autoreset_rnn.py:31-39)

    @property
    def gates(self):
        return 4 if self.kind == "lstm" else 3

    @property
    def state_width(self):
        return 2 * self.hidden if self.kind == "lstm" else self.hidden


@dataclasses.dataclass
class ConvSpec:
    prefix: str
    cin: int
    cout: int
    k: int
    stride: int
    in_hw: Tuple[int, int]
    out_hw: Tuple[int, int]
    act: int
    first: bool  # reads the (layer-normed) NCHW observation directly
    s2d: int = 0  # first layer only: the observation is space-to-depth'd by this factor (= stride) before the gather
    pad: int = 0  # padding on every side (in_hw is the unpadded input)
    pad_mode: int = 0  # nn.Conv2d's padding_mode: 0 zeros, 1 reflect, 2 replicate, 3 circular


@dataclasses.dataclass
class PoolSpec:
    """nn.MaxPool2d(2) between convolutions (modules/cnn.py:61-62): floor(H/2) x floor(W/2) windows, no parameters."""
    prefix: str
    c: int
    in_hw: Tuple[int, int]
    out_hw: Tuple[int, int]


@dataclasses.dataclass
class ConvNdSpec:
    """nn.Conv1d / nn.Conv3d of an observation with one or three spatial dimensions (modules/cnn.py:60-71), on a
    channels-last volume [n, D, H, W, C] (Conv1d: D = H = 1; the tuples below are padded with leading 1s / 0s to three
    entries).  Explicit patch matrix + dense GEMM."""
    prefix: str
    cin: int
    cout: int
    kern: Tuple[int, int, int]
    stride: int
    pads: Tuple[int, int, int]
    in_sp: Tuple[int, int, int]  # unpadded input volume
    out_sp: Tuple[int, int, int]
    act: int
    pad_mode: int = 0


@dataclasses.dataclass
class PoolNdSpec:
    """nn.MaxPool1d(2) / nn.MaxPool3d(2) between those convolutions."""
    prefix: str
    c: int
    in_sp: Tuple[int, int, int]
    out_sp: Tuple[int, int, int]
    win: Tuple[int, int, int]


@dataclasses.dataclass
class ObsLayerNormSpec:
    """LayerNorm over a whole image observation (C,H,W): fused into the first convolution's gather, or (``explicit``,
    encoders with padding or max-pooling) written out once as a float32 channels-last image."""
    prefix: str
    shape: Tuple[int, int, int]
    explicit: bool = False


@dataclasses.dataclass
class EncoderSpec:
    key: str
    shape: Union[int, Tuple[int, ...]]
    layers: list  # LayerNormSpec | LinearSpec | ObsLayerNormSpec | ConvSpec
    out_dim: int


@dataclasses.dataclass
class NetSpec:
    obs_encoders: List[EncoderSpec]
    actor_backbone: list
    state_encoders: Optional[List[EncoderSpec]]  # None when the backbone is shared
    critic_backbone: Optional[list]
    actor_head: LinearSpec
    critic_head: LinearSpec
    act_dims: List[int]
    hidden_dim: int
    value_dim: int
    shared_backbone: bool
    params: "OrderedDict[str, ParamInfo]"
    total_params: int
    popart: bool = False  # critic head is a PopArtValueHead: float64 running statistics ride along (POPART_KEYS)
    num_rnn_layers: int = 0  # GRU / LSTM layers at the end of each backbone (GruSpec + the rnn_norm LayerNormSpec)
    std_type: Optional[str] = None  # continuous actions: "fixed" | "separate_learnable" | "shared_learnable" (log_std)
    rnn_state_width: int = 0  # per-layer width of the stored policy state (H, or 2H for LSTM: cat(h, c))
    popart_keys: Tuple[str, str, str] = ()  # state_dict keys of the float64 running statistics (set in __post_init__)
    aux_head: Optional[LinearSpec] = None  # PPG: auxiliary value head on the actor's features (`auxiliary_head=True`)

    def __post_init__(self):
        if not self.popart_keys:
            self.popart_keys = POPART_KEYS


# state_dict keys of the PopArt head (popart.py:21-22,30-31; modules/utils.py:80-82), in the reference's order
POPART_W = "critic_head._PopArtValueHead__weight"
POPART_B = "critic_head._PopArtValueHead__bias"
_RMS = "critic_head._PopArtValueHead__rms._RunningMeanStd__"
POPART_KEYS = (_RMS + "mean", _RMS + "mean_sq", _RMS + "debiasing_term")
POPART_BETA, POPART_EPS = 0.99999, 1e-5  # PopArtValueHead defaults (the policy never overrides them)


def _allow_s2d():
    import os
    return os.environ.get("SRL_EXPLICIT_CONV", "0") != "1"  # the im2col fallback works on the planar layouts


def _conv_out(size, k, s, p=0):
    return (size + 2 * p - (k - 1) - 1) // s + 1


class _Builder:
    """Walks the reference's constructor in order, recording parameters (and initialising them)."""

    def __init__(self, seed: Optional[int]):
        self.params: "OrderedDict[str, ParamInfo]" = OrderedDict()
        self.values: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self.init = seed is not None
        if self.init:
            torch.manual_seed(seed)

    def _add(self, name, shape, value=None, layout="plain", chw=None, s2d=0, ref_name=""):
        self.params[name] = ParamInfo(name, tuple(shape), layout=layout, chw=chw, s2d=s2d, ref_name=ref_name)
        if self.init:
            self.values[ref_name or name] = value

    # default resets of nn.Linear / nn.Conv2d (torch/nn/modules/linear.py, conv.py)
    def _default_wb(self, wshape):
        if not self.init:
            return None, None
        w = torch.empty(wshape)
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        fan_in = int(math.prod(wshape[1:]))
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        b = torch.empty(wshape[0])
        torch.nn.init.uniform_(b, -bound, bound)
        return w, b

    def layernorm(self, prefix, shape, s2d=0):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        lay = "ln_s2d" if s2d else "plain"
        self._add(f"{prefix}.weight", shape, torch.ones(shape) if self.init else None, lay, s2d=s2d)
        self._add(f"{prefix}.bias", shape, torch.zeros(shape) if self.init else None, lay, s2d=s2d)

    def linear(self, prefix, fin, fout, layout="plain", chw=None, ref_names=("", "")):
        w, b = self._default_wb((fout, fin))
        self._add(f"{prefix}.weight", (fout, fin), w, layout, chw, ref_name=ref_names[0])
        self._add(f"{prefix}.bias", (fout,), b, ref_name=ref_names[1])

    def conv(self, prefix, cin, cout, k, layout, s2d=0, nd=2):
        w, b = self._default_wb((cout, cin) + (k,) * nd)
        self._add(f"{prefix}.weight", (cout, cin) + (k,) * nd, w, layout, s2d=s2d)
        self._add(f"{prefix}.bias", (cout,), b)

    def uniform(self, name, shape, bound):
        v = None
        if self.init:
            v = torch.empty(shape)
            torch.nn.init.uniform_(v, -bound, bound)
        self._add(name, shape, v)

    def orthogonal(self, name, gain):
        if self.init:
            torch.nn.init.orthogonal_(self.values[self.params[name].key], gain=gain)

    def zero(self, name):
        if self.init:
            self.values[self.params[name].key].zero_()


PAD_MODES = {"zeros": 0, "reflect": 1, "replicate": 2, "circular": 3}  # nn.ConvNd's padding_mode (modules/cnn.py:107-113)


def _pad_mode(padding, padding_mode, extents) -> int:
    """Validate one layer's padding against nn.ConvNd's rules; returns the mode code of the kernels."""
    if isinstance(padding, (tuple, list, str)):
        raise NotImplementedError("per-axis / named padding is not implemented on the HIP path")
    if padding_mode not in PAD_MODES:
        raise ValueError(f"padding_mode must be one of {list(PAD_MODES)}, got `{padding_mode}`")  # torch's own check
    if padding > 8:
        raise NotImplementedError("padding wider than 8 is not implemented on the HIP path")
    if padding and padding_mode == "reflect" and any(padding >= d for d in extents):
        raise ValueError("reflect padding must be smaller than the input extent")
    if padding and padding_mode == "circular" and any(padding > d for d in extents):
        raise ValueError("circular padding must not exceed the input extent")
    return PAD_MODES[padding_mode] if padding else 0


def _small_first_on() -> bool:
    import os
    return os.environ.get("SRL_CONV_SMALL", "1") != "0" and os.environ.get("SRL_CONV_SMALL_FIRST", "1") != "0"


def _build_encoders(b: _Builder, root: str, dims: Dict, hidden: int, act: int, act_name: str, cnn_layers: Dict,
                    use_maxpool: Optional[Dict] = None):
    encs = []
    for key, shape in dims.items():
        base = f"{root}.{key}"
        if isinstance(shape, int):
            b.layernorm(f"{base}.0", shape)
            b.linear(f"{base}.1.0", shape, hidden)
            b.layernorm(f"{base}.1.2", hidden)
            layers = [LayerNormSpec(f"{base}.0", shape), LinearSpec(f"{base}.1.0", shape, hidden, act),
                      LayerNormSpec(f"{base}.1.2", hidden)]
            encs.append(EncoderSpec(key, shape, layers, hidden))
            continue
        shape = tuple(shape)
        if len(shape) in (2, 4):
            encs.append(_build_nd_encoder(b, key, base, shape, hidden, act, act_name, cnn_layers.get(key),
                                          bool((use_maxpool or {}).get(key, False))))
            continue
        if len(shape) != 3:
            raise NotImplementedError(f"observation `{key}` of shape {shape}: vectors and observations with one to three "
                                      "spatial dimensions (C, ...) are implemented")
        cb = f"{base}.1._Convolution__model"
        c, h, w = shape
        cfg = cnn_layers.get(key)
        if cfg is None:  # modules/cnn.py:96-98 default stack
            cfg = [(c, 5, 1, 0, "zeros"), (c * 2, 3, 1, 0, "zeros"), (c, 3, 1, 0, "zeros")]
        # a strided first convolution whose stride divides kernel and image is run on the space-to-depth'd
        # observation (contiguous patch rows); its weight and the LayerNorm tables then live in that layout
        k0, s0 = cfg[0][1], cfg[0][2]
        pool = bool((use_maxpool or {}).get(key, False))
        # padding or pooling: the general path -- LayerNorm written out channels-last, every convolution (the first
        # included) an NHWC implicit GEMM over a padded / pooled activation
        generic = pool or any(layer[3] != 0 for layer in cfg)
        # a first layer with 4 or 8 channels on both sides, 3 x 3 / 5 x 5 (4 input channels), stride 1 -- the default stack on a
        # 4- or 8-plane observation (the football preset) -- runs as a direct vector-unit convolution on the written-out LayerNorm
        # (csrc/conv_small.hip, round 6; mirrors srl_conv2d_small_supported): the fused first-layer kernels put its 4 output
        # channels on 256 x 32 matrix-core tiles (59 + 122 ms per football-sized update against ~25 GB of activations)
        if (not generic and _small_first_on() and s0 == 1 and k0 in (3, 5) and c in (4, 8) and cfg[0][0] in (4, 8)
                and not (k0 == 5 and c == 8)):
            generic = True
        s2d = s0 if (not generic and _allow_s2d() and s0 >= 1 and k0 % s0 == 0 and h % s0 == 0 and w % s0 == 0 and
                     (c * s0 * s0) % 4 == 0) else 0
        b.layernorm(f"{base}.0", shape, s2d=s2d)
        layers = [ObsLayerNormSpec(f"{base}.0", shape, explicit=generic)]
        gain = torch.nn.init.calculate_gain(act_name)
        idx = 0  # position in the reference's nn.Sequential (modules/cnn.py:57-72)
        for i, (cout, k, stride, padding, padding_mode) in enumerate(cfg):
            if pool and i != len(cfg) - 1:  # the pooling layer sits BEFORE convolution i (modules/cnn.py:61-63)
                ph, pw = h // 2, w // 2
                if ph <= 0 or pw <= 0:
                    raise ValueError(f"CNN Dimension error, got {(ph, pw)} after max-pooling")
                layers.append(PoolSpec(f"{cb}.{idx}", c, (h, w), (ph, pw)))
                h, w = ph, pw
                idx += 1
            mode = _pad_mode(padding, padding_mode, (h, w))
            oh, ow = _conv_out(h + 2 * padding, k, stride), _conv_out(w + 2 * padding, k, stride)
            if oh <= 0 or ow <= 0:
                raise ValueError(f"CNN Dimension error, got {(oh, ow)} after convolution")
            name = f"{cb}.{idx}"
            idx += 2  # the convolution and its activation
            first = i == 0 and not generic
            b.conv(name, c, cout, k, ("conv_s2d" if s2d else "plain") if first else "conv_nhwc", s2d=s2d if first else 0)
            b.orthogonal(f"{name}.weight", gain)  # modules/cnn.py:73-84 (use_orthogonal=True)
            b.zero(f"{name}.bias")
            layers.append(ConvSpec(name, c, cout, k, stride, (h, w), (oh, ow), act, first=first,
                                   s2d=s2d if first else 0, pad=int(padding), pad_mode=mode))
            c, h, w = cout, oh, ow
        sizes = [c * h * w]
        while sizes[-1] > hidden * 8:  # modules/cnn.py:86-91
            sizes.append(sizes[-1] // 2)
        sizes.append(hidden)
        fb = f"{cb}.{idx + 1}"  # nn.Flatten takes index idx
        for j in range(len(sizes) - 1):
            first_fc = j == 0
            b.linear(f"{fb}.{3 * j}", sizes[j], sizes[j + 1], "fc_from_chw" if first_fc else "plain",
                     (c, h, w) if first_fc else None)
            b.layernorm(f"{fb}.{3 * j + 2}", sizes[j + 1])
            layers.append(LinearSpec(f"{fb}.{3 * j}", sizes[j], sizes[j + 1], 1))  # utils.mlp default ReLU
            layers.append(LayerNormSpec(f"{fb}.{3 * j + 2}", sizes[j + 1]))
        encs.append(EncoderSpec(key, shape, layers, hidden))
    return encs


def _build_nd_encoder(b: _Builder, key: str, base: str, shape, hidden: int, act: int, act_name: str, cfg, pool: bool):
    """Convolution encoder of a (C, L) or (C, D, H, W) observation: same module sequence as the image encoder
    (modules/cnn.py:93-135) with nn.Conv1d / nn.Conv3d and MaxPool1d / MaxPool3d."""
    nd = len(shape) - 1
    c = shape[0]
    sp = (1,) * (3 - nd) + tuple(shape[1:])
    lead = 3 - nd
    cb = f"{base}.1._Convolution__model"
    if cfg is None:  # modules/cnn.py:96-98 default stack
        cfg = [(c, 5, 1, 0, "zeros"), (c * 2, 3, 1, 0, "zeros"), (c, 3, 1, 0, "zeros")]
    b.layernorm(f"{base}.0", shape)
    layers = [ObsLayerNormSpec(f"{base}.0", (c, 1, int(math.prod(sp))), explicit=True)]
    gain = torch.nn.init.calculate_gain(act_name)
    idx = 0
    for i, (cout, k, stride, padding, padding_mode) in enumerate(cfg):
        if pool and i != len(cfg) - 1:
            win = (1,) * lead + (2,) * nd
            out = tuple(d // w for d, w in zip(sp, win))
            if min(out) <= 0:
                raise ValueError(f"CNN Dimension error, got {out[lead:]} after max-pooling")
            layers.append(PoolNdSpec(f"{cb}.{idx}", c, sp, out, win))
            sp = out
            idx += 1
        mode = _pad_mode(padding, padding_mode, sp[lead:])
        kern = (1,) * lead + (k,) * nd
        pads = (0,) * lead + (int(padding),) * nd
        out = tuple(_conv_out(d + 2 * p, kk, stride) if j >= lead else 1 for j, (d, p, kk) in enumerate(zip(sp, pads, kern)))
        if min(out) <= 0:
            raise ValueError(f"CNN Dimension error, got {out[lead:]} after convolution")
        name = f"{cb}.{idx}"
        idx += 2
        b.conv(name, c, cout, k, "conv_nhwc", nd=nd)
        b.orthogonal(f"{name}.weight", gain)
        b.zero(f"{name}.bias")
        layers.append(ConvNdSpec(name, c, cout, kern, stride, pads, sp, out, act, pad_mode=mode))
        c, sp = cout, out
    vox = int(math.prod(sp))
    sizes = [c * vox]
    while sizes[-1] > hidden * 8:
        sizes.append(sizes[-1] // 2)
    sizes.append(hidden)
    fb = f"{cb}.{idx + 1}"
    for j in range(len(sizes) - 1):
        first_fc = j == 0
        b.linear(f"{fb}.{3 * j}", sizes[j], sizes[j + 1], "fc_from_chw" if first_fc else "plain", (c, 1, vox) if first_fc else None)
        b.layernorm(f"{fb}.{3 * j + 2}", sizes[j + 1])
        layers.append(LinearSpec(f"{fb}.{3 * j}", sizes[j], sizes[j + 1], 1))
        layers.append(LayerNormSpec(f"{fb}.{3 * j + 2}", sizes[j + 1]))
    return EncoderSpec(key, tuple(shape), layers, hidden)


def _build_backbone(b: _Builder, root: str, in_dim: int, hidden: int, dense_layers: int, act: int, layernorm: bool,
                    num_rnn_layers: int = 0, rnn_type: str = "gru"):
    layers = []
    stride = 3 if layernorm else 2
    names = []
    d = in_dim
    for j in range(dense_layers):
        p = f"{root}.fc.{stride * j}"
        b.linear(p, d, hidden)
        names += [f"{p}.weight", f"{p}.bias"]
        layers.append(LinearSpec(p, d, hidden, act))
        if layernorm:
            q = f"{root}.fc.{stride * j + 2}"
            b.layernorm(q, hidden)
            names += [f"{q}.weight", f"{q}.bias"]
            layers.append(LayerNormSpec(q, hidden))
        d = hidden
    for n in names:  # recurrent_backbone.py:41-47: orthogonal on >=2-D weights, zero on every bias
        if n.endswith("weight") and len(b.params[n].ref_shape) >= 2:
            b.orthogonal(n, math.sqrt(2))
        if n.endswith("bias"):
            b.zero(n)
    if num_rnn_layers:
        # nn.GRU.reset_parameters: every tensor uniform(+-1/sqrt(H)) in _flat_weights order, then
        # recurrent_backbone.py:54-58: orthogonal (gain 1) on the matrices, zeros on the biases
        rp = f"{root}.rnn._AutoResetRNN__net"
        bound = 1.0 / math.sqrt(hidden)
        ng = 4 if rnn_type == "lstm" else 3
        for l in range(num_rnn_layers):
            b.uniform(f"{rp}.weight_ih_l{l}", (ng * hidden, hidden), bound)
            b.uniform(f"{rp}.weight_hh_l{l}", (ng * hidden, hidden), bound)
            b.uniform(f"{rp}.bias_ih_l{l}", (ng * hidden,), bound)
            b.uniform(f"{rp}.bias_hh_l{l}", (ng * hidden,), bound)
        b.layernorm(f"{root}.rnn_norm", hidden)
        for l in range(num_rnn_layers):
            b.orthogonal(f"{rp}.weight_ih_l{l}", 1.0)
            b.orthogonal(f"{rp}.weight_hh_l{l}", 1.0)
            b.zero(f"{rp}.bias_ih_l{l}")
            b.zero(f"{rp}.bias_hh_l{l}")
        layers.append(GruSpec(rp, hidden, num_rnn_layers, rnn_type))
        layers.append(LayerNormSpec(f"{root}.rnn_norm", hidden))
    return layers


def build_netspec(obs_dim, action_dim, hidden_dim=128, state_dim=None, value_dim=1, num_dense_layers=2,
                  cnn_layers=None, use_maxpool=None, num_rnn_layers=0, popart=False, activation="relu", layernorm=True,
                  shared_backbone=False, continuous_action=False, auxiliary_head=False, seed: Optional[int] = None,
                  rnn_type="gru", **_unused):
    """Returns ``(NetSpec, values)``; ``values`` is the name -> CPU tensor dict of initial weights (reference
    layout) when ``seed`` is given, else ``None``."""
    if num_rnn_layers and rnn_type not in ("gru", "lstm"):
        raise NotImplementedError(f"rnn_type `{rnn_type}`: only the GRU and LSTM cells of AutoResetRNN are on the HIP path")
    if auxiliary_head and shared_backbone:  # actor_critic_policy.py:196-197
        raise AttributeError("Cannot use shared backbone when requiring auxiliary value head.")
    std_type = _unused.get("std_type", "fixed")
    if continuous_action and std_type not in ("fixed", "separate_learnable", "shared_learnable"):
        raise NotImplementedError(f"Standard deviation type {std_type} not implemented.")
    if activation not in ACTS:
        raise NotImplementedError(f"Activation function {activation} not implemented.")
    act = ACTS[activation]
    obs_dims = {"obs": obs_dim} if isinstance(obs_dim, int) else dict(obs_dim)
    if state_dim is not None and isinstance(state_dim, int):
        state_dim = {"state": state_dim}
    act_dims = [action_dim] if isinstance(action_dim, int) else list(action_dim)
    cnn_layers = cnn_layers or {}

    # LAPACK's QR inside orthogonal_ rounds differently with different thread counts; pin it so that a seed
    # always gives the same weights (and the same as the single-threaded reference run the fixtures record)
    threads = torch.get_num_threads()
    if seed is not None:
        torch.set_num_threads(1)
    try:
        return _build(obs_dims, state_dim, act_dims, cnn_layers, hidden_dim, value_dim, num_dense_layers, act, activation,
                      layernorm, shared_backbone, seed, popart, num_rnn_layers, rnn_type,
                      (std_type, float(_unused.get("init_log_std", -0.5))) if continuous_action else None,
                      use_maxpool=use_maxpool, auxiliary_head=bool(auxiliary_head))
    finally:
        torch.set_num_threads(threads)


def _build(obs_dims, state_dim, act_dims, cnn_layers, hidden_dim, value_dim, num_dense_layers, act, activation, layernorm,
           shared_backbone, seed, popart=False, num_rnn_layers=0, rnn_type="gru", continuous=None, use_maxpool=None,
           auxiliary_head=False):
    b = _Builder(seed)
    if continuous is not None and continuous[0] != "shared_learnable":
        # one vector of log standard deviations; a direct nn.Parameter of the net, so it leads the state_dict.
        # `fixed` is excluded from the optimiser in the reference (requires_grad=False): its gradient stays zero here
        b._add("log_std", (sum(act_dims),), continuous[1] * torch.ones(sum(act_dims)) if b.init else None)
    obs_enc = _build_encoders(b, "obs_modules_dict", obs_dims, hidden_dim, act, activation, cnn_layers, use_maxpool)
    actor_bb = _build_backbone(b, "actor_backbone", hidden_dim * len(obs_dims), hidden_dim, num_dense_layers, act,
                               layernorm, num_rnn_layers, rnn_type)
    state_enc = critic_bb = None
    if not shared_backbone:
        sdims = state_dim or obs_dims
        state_enc = _build_encoders(b, "state_modules_dict", sdims, hidden_dim, act, activation, cnn_layers, use_maxpool)
        critic_bb = _build_backbone(b, "critic_backbone", hidden_dim * len(sdims), hidden_dim, num_dense_layers, act,
                                    layernorm, num_rnn_layers, rnn_type)
    b.linear("actor_head", hidden_dim, sum(act_dims))
    b.orthogonal("actor_head.weight", 0.01)  # actor_critic_policy.py:109-112
    b.zero("actor_head.bias")
    std_type = None
    if continuous is not None:  # Normal(mean, std): actor_critic_policy.py:85-96
        std_type, init_log_std = continuous
        if std_type == "shared_learnable":
            b.linear("log_std", hidden_dim, sum(act_dims))
            b.orthogonal("log_std.weight", 0.01)
            b.zero("log_std.bias")
        # (the vector forms were registered first: a module's own parameters precede its sub-modules' in state_dict)
    if popart:
        # PopArtValueHead (popart.py:19-26): the nn.Linear default reset, no orthogonal re-initialisation; the same
        # linear map on the device, only its state_dict keys and the float64 running statistics differ
        b.linear("critic_head", hidden_dim, value_dim, ref_names=(POPART_W, POPART_B))
        if b.init:
            b.values[POPART_KEYS[0]] = torch.zeros(value_dim, dtype=torch.float64)
            b.values[POPART_KEYS[1]] = torch.zeros(value_dim, dtype=torch.float64)
            b.values[POPART_KEYS[2]] = torch.zeros(1, dtype=torch.float64)
    else:
        b.linear("critic_head", hidden_dim, value_dim)
        b.orthogonal("critic_head.weight", 0.01)
        b.zero("critic_head.bias")
    if auxiliary_head:  # PPG's second value head, on the ACTOR's features (actor_critic_policy.py:105-107); the module's last child
        b.linear("auxiliary_value_head", hidden_dim, value_dim)
        b.orthogonal("auxiliary_value_head.weight", 0.01)
        b.zero("auxiliary_value_head.bias")

    off = 0
    for info in b.params.values():
        info.offset = off
        off += (info.numel + 3) // 4 * 4  # 16-byte aligned starts (float4 staging in the GEMM)
    spec = NetSpec(obs_enc, actor_bb, state_enc, critic_bb, LinearSpec("actor_head", hidden_dim, sum(act_dims), 0),
                   LinearSpec("critic_head", hidden_dim, value_dim, 0), act_dims, hidden_dim, value_dim, shared_backbone,
                   b.params, off, popart, num_rnn_layers,
                   std_type, (2 * hidden_dim if rnn_type == "lstm" else hidden_dim) if num_rnn_layers else 0,
                   aux_head=LinearSpec("auxiliary_value_head", hidden_dim, value_dim, 0) if auxiliary_head else None)
    return spec, (b.values if b.init else None)


def popart_keys_of(head: str):
    rms = f"{head}._PopArtValueHead__rms._RunningMeanStd__"
    return (rms + "mean", rms + "mean_sq", rms + "debiasing_term")


def build_smac_netspec(obs_dim: int, state_dim: int, act_dim: int, hidden_dim: int, num_rnn_layers: int = 1,
                       act_init_gain: float = 0.01, seed: Optional[int] = None):
    """``SMACNet`` with flat (not agent-specific) observations (``game_policies/smac_rnn.py:88-167``): actor on
    ``local_obs``, critic on ``state``, each ``LayerNorm -> mlp([d, H, H], ReLU, layernorm=True)`` -> ``AutoResetRNN``
    -> ``LayerNorm``; heads ``policy_head`` and the PopArt ``value_head``.  The recurrent cell is an **LSTM**:
    ``AutoResetRNN``'s default ``rnn_type`` (``autoreset_rnn.py:9``) is what ``SMACNet`` gets (``:125-126``), so the
    stored state is ``cat(h, c)``, ``2 * hidden_dim`` wide.  Parameters are registered in the reference's
    ``state_dict`` order and initialised by replaying its constructor (``:137-160``)."""
    H = hidden_dim
    threads = torch.get_num_threads()
    if seed is not None:
        torch.set_num_threads(1)
    try:
        b = _Builder(seed)

        def base(root, din):
            b.layernorm(f"{root}.0", din)
            b.linear(f"{root}.1.0", din, H)
            b.layernorm(f"{root}.1.2", H)
            b.linear(f"{root}.1.3", H, H)
            b.layernorm(f"{root}.1.5", H)
            return [LayerNormSpec(f"{root}.0", din), LinearSpec(f"{root}.1.0", din, H, 1), LayerNormSpec(f"{root}.1.2", H),
                    LinearSpec(f"{root}.1.3", H, H, 1), LayerNormSpec(f"{root}.1.5", H)]

        def rnn(root):
            bound = 1.0 / math.sqrt(H)
            for l in range(num_rnn_layers):
                b.uniform(f"{root}.weight_ih_l{l}", (4 * H, H), bound)
                b.uniform(f"{root}.weight_hh_l{l}", (4 * H, H), bound)
                b.uniform(f"{root}.bias_ih_l{l}", (4 * H,), bound)
                b.uniform(f"{root}.bias_hh_l{l}", (4 * H,), bound)

        a_layers = base("actor_base", obs_dim)
        c_layers = base("critic_base", state_dim)
        a_bb = c_bb = []
        if num_rnn_layers:
            ar, cr = "actor_rnn._AutoResetRNN__net", "critic_rnn._AutoResetRNN__net"
            rnn(ar)
            rnn(cr)
            b.layernorm("actor_rnn_norm", H)
            b.layernorm("critic_rnn_norm", H)
            a_bb = [GruSpec(ar, H, num_rnn_layers, "lstm"), LayerNormSpec("actor_rnn_norm", H)]
            c_bb = [GruSpec(cr, H, num_rnn_layers, "lstm"), LayerNormSpec("critic_rnn_norm", H)]
        b.linear("policy_head", H, act_dim)
        b.linear("value_head", H, 1, ref_names=("value_head._PopArtValueHead__weight", "value_head._PopArtValueHead__bias"))
        pkeys = popart_keys_of("value_head")
        if b.init:
            b.values[pkeys[0]] = torch.zeros(1, dtype=torch.float64)
            b.values[pkeys[1]] = torch.zeros(1, dtype=torch.float64)
            b.values[pkeys[2]] = torch.zeros(1, dtype=torch.float64)
        # re-initialisation in the reference's order (:137-160)
        for root in ("actor_base", "critic_base"):
            for lin in (f"{root}.1.0", f"{root}.1.3"):
                b.orthogonal(f"{lin}.weight", math.sqrt(2))
                b.zero(f"{lin}.bias")
        if num_rnn_layers:
            for root in (ar, cr):
                for l in range(num_rnn_layers):
                    b.orthogonal(f"{root}.weight_ih_l{l}", 1.0)
                    b.orthogonal(f"{root}.weight_hh_l{l}", 1.0)
                    b.zero(f"{root}.bias_ih_l{l}")
                    b.zero(f"{root}.bias_hh_l{l}")
        b.orthogonal("policy_head.weight", act_init_gain)
        b.zero("policy_head.bias")
    finally:
        torch.set_num_threads(threads)
    off = 0
    for info in b.params.values():
        info.offset = off
        off += (info.numel + 3) // 4 * 4
    spec = NetSpec([EncoderSpec("local_obs", obs_dim, a_layers, H)], a_bb, [EncoderSpec("state", state_dim, c_layers, H)], c_bb,
                   LinearSpec("policy_head", H, act_dim, 0), LinearSpec("value_head", H, 1, 0), [act_dim], H, 1, False,
                   b.params, off, True, num_rnn_layers, None, 2 * H if num_rnn_layers else 0, pkeys)
    return spec, (b.values if b.init else None)
