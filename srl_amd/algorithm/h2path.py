"""The Atari encoder (modules/cnn.py:93-135: 8x8/4 -> 4x4/2 -> 3x3/1 convolutions, Flatten, Linear 3136 -> 512, ReLU after each)
on pre-split ("h2") activations: ``csrc/h2conv.h`` (image-stationary convolutions, their data and weight gradients) and
``csrc/h2gemm.h`` (the Linear's products through an LDS-DMA ring).  Round 3's kernels kept activations float32 in HBM and
split them into two f16 pieces again in every tile that staged them (the staging was what bounded them); here the producer's
epilogue splits once and writes the pieces -- the same 4 bytes per element -- and the consumers move bytes.

One object per executor (``HipNet`` and its twin): workspace buffers, the per-update weight preparations and the device
floats that carry scales / ranges between kernels.  ``forward`` returns the Linear's float32 output (what the layers behind
the encoder read); ``backward`` takes the gradient with respect to it.  Formats between the layers:

    frames (uint8, space-to-depth'd)  --obs_fwd_h2-->  a1: h2p rows, parity-class pixel order      + sign words (natural order)
    a1  --conv2 fwd-->  a2: planar 9x9x64 + sign bytes (h2 order)  --conv3 fwd-->  a3: h2p rows [n, 3136] + sign bytes
    a3  --Linear (h2gemm)-->  y float32 [n, 512] + sign words
    dy (float32) --pack--> h2p rows --Linear dgrad (h2gemm, mask a3)--> dz3: h2p rows --conv3 dgrad (mask a2)--> dz2: planar
    dz2 --conv2 dgrad (mask a1)--> dz1: float32 NHWC --> the first layer's backward (obs_bwd, round 2)
    weight gradients: conv3 (a2, dz3), conv2 (a1, dz2) image-stationary; the Linear's through the round-3 kernel.
"""
import os
import weakref
from typing import Optional

import torch

from srl_amd import hip
from srl_amd.algorithm import netspec as ns

ENABLED = os.environ.get("SRL_H2", "1") != "0"

# device floats of one block.  Two arrays, indexed by the same enumeration: what a PASS measures and derives (ranges of the
# activations / gradients and the power-of-two scales made of them: data-dependent, so one array per forward pass -- per `tag`,
# i.e. per trunk and per piece of `_trunk_fwd`: a second forward before the first one's backward must not overwrite them) and
# what depends on the WEIGHTS only (one array per block and parameter version).
# (the measured ranges of a pass sit side by side: one fill zeroes the forward's three, one the backward's four)
(M_A1, M_A2, M_A3, M_DY, M_DZ3, M_DZ2, M_DZ1, S_A1, S_A2, S_A3, S_DY, S_DZ3, S_DZ2,
 S_W2, R_W2, B_W2, S_W3, R_W3, B_W3, S_WF, R_WF, S_W3G, R_W3G, S_W2G, R_W2G, S_WFT, R_WFT, N_SLOTS) = range(28)
N_PASS_SLOTS = S_W2  # indices below: per pass; from here on: per weights


def match(layers) -> Optional[tuple]:
    """(obs LayerNorm, conv1, conv2, conv3, Linear) when the encoder starts with exactly the stack these kernels are compiled
    for (h2conv.h's geometries), else None."""
    if len(layers) < 5:
        return None
    ln, c1, c2, c3, fc = layers[:5]
    if not (isinstance(ln, ns.ObsLayerNormSpec) and not ln.explicit and tuple(ln.shape) == (4, 84, 84)):
        return None
    if not all(isinstance(c, ns.ConvSpec) for c in (c1, c2, c3)) or not isinstance(fc, ns.LinearSpec):
        return None
    geo = [(c.cin, c.cout, c.k, c.stride, tuple(c.in_hw), c.pad, c.act) for c in (c1, c2, c3)]
    want = [(4, 32, 8, 4, (84, 84), 0, hip.ACT_RELU), (32, 64, 4, 2, (20, 20), 0, hip.ACT_RELU), (64, 64, 3, 1, (9, 9), 0, hip.ACT_RELU)]
    if geo != want or not (c1.first and c1.s2d == 4) or c2.first or c3.first:
        return None
    if (fc.in_features, fc.act) != (3136, hip.ACT_RELU) or fc.out_features % 32:
        return None
    return ln, c1, c2, c3, fc


class H2Cnn:

    def __init__(self, net, layers):
        # (a proxy: the executor keeps its blocks in `_h2_blocks`; a strong reference back would make every dropped trainer wait
        # for the cycle collector with tens of GB of workspace -- an OOM in a process that builds several, e.g. the test suite)
        self.net = weakref.proxy(net)
        self.ln, self.c1, self.c2, self.c3, self.fc = layers
        self.H = self.fc.out_features
        # every workspace buffer of this block carries the block's own name: an executor may hold several (separate actor and
        # critic encoders both read the key "obs"; two image keys under one trunk), each with its own weights and scales
        self.pfx = f"h2[{self.c1.prefix}]."
        # the first layer's position sums (Q, R, C) of the chunks so far are accumulating in this block's workspace
        # (srl_conv2d_obs_bwd's `phase`): whoever runs this encoder's next backward continues, the executor's last chunk closes
        self.open = False
        self._open_ws = None

    # ------------------------------------------------------------------ helpers
    def _slots(self, tag=None):
        """tag None: the per-weights array; else the array of the pass `tag`."""
        if tag is None:   # weight-only data: one copy for an executor and its twins (hipnet.py `wws`)
            return self.net.wws.get(f"{self.pfx}wslots", N_SLOTS)
        return self.net.ws.get(f"{tag}{self.pfx}slots", N_SLOTS)

    def _slot(self, i, tag=None):
        assert (i < N_PASS_SLOTS) == (tag is not None), i
        return self._slots(tag).data_ptr() + 4 * i

    def _bytes(self, name, nbytes):
        return self.net.ws.get(name, (nbytes + 3) // 4).data_ptr()

    def _wbytes(self, name, nbytes):
        return self.net.wws.get(self.pfx + name, (nbytes + 3) // 4).data_ptr()

    def _prepare_weights(self):
        """Once per parameter version and block: h2p copies of the weights in the orientations the kernels read, their
        scales and the row norms that bound the outputs."""
        net = self.net
        key = self.pfx + "weights"
        marker = self._wbytes("weights", 16)
        if net._derived_fresh(key, marker):
            return
        p = net._p
        W = self._slot
        desc2 = hip.conv_desc(1, 20, 20, 32, 4, 4, 2, 64, hip.ACT_RELU)
        desc3 = hip.conv_desc(1, 9, 9, 64, 3, 3, 1, 64, hip.ACT_RELU)
        for L, K, s_w, r_w, b_w, name in ((self.c2, 512, S_W2, R_W2, B_W2, "w2"), (self.c3, 576, S_W3, R_W3, B_W3, "w3")):
            amax = net._weight_range(L.prefix, 64 * K)
            hip.h2_weights(p(f"{L.prefix}.weight"), 64, K, 0, amax, W(s_w), W(r_w), self._wbytes(name, 64 * K * 4))
            self._slots()[b_w:b_w + 1].zero_()
            hip.absmax(p(f"{L.prefix}.bias"), 64, W(b_w))
        amax = net._weight_range(self.c3.prefix, 64 * 576)
        hip.h2_weights(p(f"{self.c3.prefix}.weight"), 64, 576, 2, amax, W(S_W3G), W(R_W3G), self._wbytes("w3g", 64 * 576 * 4), desc=desc3)
        amax = net._weight_range(self.c2.prefix, 64 * 512)
        hip.h2_weights(p(f"{self.c2.prefix}.weight"), 128, 256, 2, amax, W(S_W2G), W(R_W2G), self._wbytes("w2g", 128 * 256 * 4), desc=desc2)
        amax = net._weight_range(self.fc.prefix, self.H * 3136)
        hip.h2_weights(p(f"{self.fc.prefix}.weight"), self.H, 3136, 0, amax, W(S_WF), W(R_WF), self._wbytes("wf", self.H * 3136 * 4))
        hip.h2_weights(p(f"{self.fc.prefix}.weight"), 3136, self.H, 1, amax, W(S_WFT), W(R_WFT), self._wbytes("wft", self.H * 3136 * 4))
        net._derived_done(marker)

    def _folded(self):
        """(descriptor of one image, buffer of the first layer's folded weights).  Its own buffer: the per-position kernels of the
        layer-by-layer path keep another format under "<prefix>.folded"."""
        desc1 = hip.conv_desc(1, 21, 21, 64, 2, 2, 1, 32, hip.ACT_RELU)
        return desc1, self.net.wws.get(f"{self.c1.prefix}.folded.h2", hip.conv2d_obs_fwd_workspace(desc1)).data_ptr()

    def prepare(self):
        """Everything of this block that depends on the parameters alone, enqueued now: the trainer calls it at the top of an
        update, where the GPU otherwise idles behind the host (leaf copies, the GAE scan and a dozen fills take the host
        ~0.4 ms to issue and the GPU ~0.05 ms to run) -- 0.25 ms of small launches that used to sit between the first chunk's
        staging and its first convolution."""
        net = self.net
        self._prepare_weights()
        desc1, fws = self._folded()
        if not net._derived_fresh(f"{self.c1.prefix}.folded:h2", fws):
            if hip.conv2d_obs_fold_h2(desc1, net._p(f"{self.ln.prefix}.weight"), net._p(f"{self.ln.prefix}.bias"),
                                      net._p(f"{self.c1.prefix}.weight"), net._p(f"{self.c1.prefix}.bias"), fws):
                net._derived_done(fws)
            else:
                net._derived.pop(fws, None)

    # ------------------------------------------------------------------ forward
    # Inference batches: the Linear's 3136-long reduction split over FC_SPLITK workgroups per tile -- the rows alone leave most CUs
    # idle (2048 rows are 32 tiles: 104 us; in 8 k-ranges 27 + 4 us for adding the slabs).  The SAME eight k-ranges for every row
    # count that takes this path: a row's result must not depend on the batch it arrives in (the streamed pieces of a large batch
    # and one small batch sample the same actions, bit for bit).  SRL_FC_SPLITK=1 switches it off (A/B).
    FC_SPLITK = int(os.environ.get("SRL_FC_SPLITK", "8"))

    def fc_splits(self, n: int) -> int:
        return self.FC_SPLITK if (n <= 8192 and self.H % 128 == 0 and self.FC_SPLITK > 1) else 1

    # Training chunks with the closing LayerNorm + heads as the consumer (opt-in, SRL_FC_TRAIN_SPLITK=2): 256 channels per workgroup
    # stage a third fewer bytes per multiply-add (the forward is bound by its ring fill) but are only 128 tiles at 16 384
    # rows -- two k-ranges restore the workgroup count, and the consumer adds the slabs and writes the finished rows for the backward
    # pass.  Measured, alternating runs on one box: 85.66 / 85.36 ms per update unsplit, 85.54 / 85.67 with two ranges, 86.0 / 86.1
    # with three / four: what the smaller fill gains the slabs' traffic takes back.  Off by default.
    FC_TRAIN_SPLITK = int(os.environ.get("SRL_FC_TRAIN_SPLITK", "1"))

    def forward(self, tag, staged, obs, n, is_u8, src, mean, rstd, row_index, split_fc=None):
        """The four layers on `n` rows.  src / mean / rstd / row_index: the first layer's staged frames and statistics as
        `_encoder_fwd` resolved them.  Returns (y Buf float32 [n, H], saved).  ``split_fc`` ("infer" | "train": the consumer
        finishes the product -- `srl_ln_heads_fwd`'s x_slabs): the Linear writes raw partial sums, `saved["fc_slabs"]` = (pointer,
        slabs, stride in floats, bias pointer, activation) tells the consumer where; y is then only valid BEHIND the consumer's
        launch ("train": it writes the finished rows there; "infer": nothing does, nothing reads them)."""
        net, ws = self.net, self.net.ws
        self._prepare_weights()
        t = f"{tag}{self.pfx}"
        W = self._slot
        P = lambda i: self._slot(i, tag)
        self._slots(tag)[M_A1:M_A3 + 1].zero_()
        a1 = self._bytes(f"{t}a1", n * 400 * 32 * 4)
        a2 = self._bytes(f"{t}a2", n * 81 * 64 * 4)
        a3 = self._bytes(f"{t}a3", n * 49 * 64 * 4)
        m1 = ws.get(f"{t}m1", n * 400, torch.int32).data_ptr()
        m2 = self._bytes(f"{t}m2", n * 81 * 8)
        m3 = self._bytes(f"{t}m3", n * 49 * 8)
        # first layer: frames -> a1
        desc1 = hip.conv_desc(n, 21, 21, 64, 2, 2, 1, 32, hip.ACT_RELU)
        fws = self._folded()[1]
        reuse = net._derived_fresh(f"{self.c1.prefix}.folded:h2", fws)
        hip.conv2d_obs_fwd_h2(desc1, src.data_ptr(), mean.data_ptr(), rstd.data_ptr(), net._p(f"{self.ln.prefix}.weight"),
                              net._p(f"{self.ln.prefix}.bias"), net._p(f"{self.c1.prefix}.weight"), net._p(f"{self.c1.prefix}.bias"),
                              a1, P(S_A1), fws, row_index, P(M_A1), m1, reuse_folded=reuse, ent_order=2,
                              records=self._bytes(f"{t}records", 16 * (n + 32)))   # (the folded weights are shared: `wws`)
        if not reuse:
            net._derived_done(fws)
        # conv2, conv3
        hip.h2_conv(hip.H2_CONV2_FWD, a1, self._wbytes("w2", 0), P(S_A1), W(S_W2), n, a2, P(M_A2),
                    bias=net._p(f"{self.c2.prefix}.bias"), act=1, out_scale=P(S_A2), bound_in=P(M_A1),
                    bound_w=W(R_W2), bound_b=W(B_W2), mask_out=m2)
        hip.h2_conv(hip.H2_CONV3_FWD, a2, self._wbytes("w3", 0), P(S_A2), W(S_W3), n, a3, P(M_A3),
                    bias=net._p(f"{self.c3.prefix}.bias"), act=1, out_scale=P(S_A3), bound_in=P(M_A2),
                    bound_w=W(R_W3), bound_b=W(B_W3), mask_out=m3)
        # Linear: float32 out (the layers behind it take the ReLU derivative from these floats, as after `_linear_fwd`)
        saved = dict(n=n, a1=a1, a2=a2, a3=a3, m1=m1, m2=m2, m3=m3, first=(src, is_u8, mean, rstd, row_index), tag=tag)
        y = net._buf(f"{tag}{self.fc.prefix}.y", n, self.H)
        ks, wide = 1, False
        if split_fc == "infer":
            ks = self.fc_splits(n)
        elif split_fc == "train" and self.FC_TRAIN_SPLITK > 1 and self.H % 256 == 0 and n >= 8192:
            ks, wide = self.FC_TRAIN_SPLITK, True
        if ks > 1:
            slabs = net._buf(f"{tag}{self.fc.prefix}.yslabs", ks * n, self.H)
            hip.h2_gemm_splitk(a3, self._wbytes("wf", 0), P(S_A3), W(S_WF), n, self.H, 3136, slabs.ptr, ks, wide=wide)
            saved["fc_slabs"] = (slabs.ptr, ks, n * self.H, net._p(f"{self.fc.prefix}.bias"), self.fc.act, wide)
            return y, saved
        hip.h2_gemm(a3, self._wbytes("wf", 0), P(S_A3), W(S_WF), n, self.H, 3136, y.ptr,
                    bias=net._p(f"{self.fc.prefix}.bias"), act=1)
        return y, saved

    # ------------------------------------------------------------------ backward
    def open_backward(self, saved) -> int:
        """Zero the backward pass's measured ranges and return the address of dy's: a producer that tracks max |dy| while it writes
        dy (srl_ln_heads_bwd) leaves it there and calls ``backward(..., dy_ranged=True)``."""
        tag = saved["tag"]
        self._slots(tag)[M_DY:M_DZ1 + 1].zero_()
        return self._slot(M_DY, tag)

    def backward(self, saved, dy, dy_ranged=False):
        """dy: Buf float32 [n, H], the gradient with respect to the Linear's pre-activation (its ReLU derivative already
        applied by the layer above).  Adds every parameter gradient of the five layers."""
        net, ws = self.net, self.net.ws
        n, tag = saved["n"], saved["tag"]
        g = net._g
        t = f"{tag}{self.pfx}"
        W = self._slot
        P = lambda i: self._slot(i, tag)
        assert dy.ld == dy.cols == self.H and dy.rows == n
        if not dy_ranged:
            self.open_backward(saved)
            # dy -> h2p rows (its range from one pass: the producer is a float32 kernel that did not track it)
            hip.absmax(dy.ptr, n * self.H, P(M_DY))
        dyh = self._bytes(f"{t}dy", n * self.H * 4)
        if self.FC_WGRAD_TN and self.H <= 2048:
            # the Linear's bias gradient = the column sums of dy, from the pass that splits it (a pass of its own: 32 us per chunk)
            pws = ws.get("h2pack_colsum", hip.h2_pack_rows_colsum_workspace(n, self.H)).data_ptr()
            hip.h2_pack_rows_colsum(dy.ptr, self.H, n, self.H, dyh, pws, g(f"{self.fc.prefix}.bias"), absmax=P(M_DY), scale_out=P(S_DY))
        else:
            hip.h2_pack_rows(dy.ptr, self.H, n, self.H, dyh, absmax=P(M_DY), scale_out=P(S_DY))
        # Linear: weight gradient beside the data-gradient chain
        net._on_side(lambda: self._fc_wgrad(n, dy, dyh, saved["a3"], tag))
        # Linear data gradient -> dz3 (h2p rows [n, 49, 64]), ReLU derivative of a3 from its sign bytes
        dz3 = self._bytes(f"{t}dz3", n * 3136 * 4)
        hip.h2_gemm(dyh, self._wbytes("wft", 0), P(S_DY), W(S_WFT), n, 3136, self.H, dz3, out_h2=True,
                    out_scale=P(S_DZ3), bound_in=P(M_DY), bound_w=W(R_WFT), out_absmax=P(M_DZ3),
                    mask_in=saved["m3"], mask_in_h2order=True)
        net._release([self.fc.prefix])   # data parallel: the Linear's 6.4 MB bucket leaves under the convolutions' backward
        # conv3: weight gradient (a2, dz3) beside its data gradient -> dz2 (planar)
        wws3 = ws.get(self.pfx + "wgrad3", hip.h2_wgrad_workspace(hip.H2_WGRAD_CONV3)).data_ptr()
        net._on_side(lambda: hip.h2_wgrad(hip.H2_WGRAD_CONV3, saved["a2"], dz3, P(S_A2), P(S_DZ3), n, wws3,
                                          g(f"{self.c3.prefix}.weight"), g(f"{self.c3.prefix}.bias")))
        dz2 = self._bytes(f"{t}dz2", n * 81 * 64 * 4)
        hip.h2_conv(hip.H2_CONV3_DGRAD, dz3, self._wbytes("w3g", 0), P(S_DZ3), W(S_W3G), n, dz2, P(M_DZ2),
                    out_scale=P(S_DZ2), bound_in=P(M_DZ3), bound_w=W(R_W3G), mask_in=saved["m2"])
        net._release([self.c3.prefix])
        # conv2: weight gradient (a1, dz2) beside its data gradient -> dz1 (float32 NHWC for the first layer's backward)
        wws2 = ws.get(self.pfx + "wgrad2", hip.h2_wgrad_workspace(hip.H2_WGRAD_CONV2)).data_ptr()
        net._on_side(lambda: hip.h2_wgrad(hip.H2_WGRAD_CONV2, saved["a1"], dz2, P(S_A1), P(S_DZ2), n, wws2,
                                          g(f"{self.c2.prefix}.weight"), g(f"{self.c2.prefix}.bias")))
        dz1 = net._buf(f"{t}dz1", n * 400, 32)
        hip.h2_conv(hip.H2_CONV2_DGRAD, dz2, self._wbytes("w2g", 0), P(S_DZ2), W(S_W2G), n, dz1.ptr, P(M_DZ1),
                    mask_in=saved["m1"])
        net._release([self.c2.prefix])
        self.first_layer_bwd(n, saved["first"], dz1.ptr, P(M_DZ1))
        net._release([self.ln.prefix, self.c1.prefix])

    def first_layer_bwd(self, n, first, dz_ptr, dz_absmax_ptr):
        """The first layer's weight / bias / LayerNorm-affine gradients from dz (float32 NHWC [n, 20, 20, 32]) with its measured
        range.  Inside the trainer's chunk loop the position sums (Q, R, C) of an executor's chunks add up in this block's
        workspace and the four gradients are formed ONCE, behind the executor's last chunk (srl_conv2d_obs_bwd's `phase`).
        Whether a call continues an accumulation is decided from the block's STATE, never from `n`: a ragged last chunk --
        including one below H2_MIN_ROWS, which `HipNet._chain_bwd` sends here from the layer-by-layer path -- must add to and
        close what the chunks before it opened.  Only OPENING one asks for a full-sized chunk."""
        net = self.net
        src, is_u8, mean, rstd, row_index = first
        desc1 = hip.conv_desc(n, 21, 21, 64, 2, 2, 1, 32, hip.ACT_RELU)
        wsz = hip.conv2d_obs_bwd_workspace(desc1)
        phase = 3
        defer = net._in_update[0] and net.last_chunk is not None and os.environ.get("SRL_OBS_BWD_DEFER", "1")[:1] != "0"
        if self.open or (defer and n >= 4096):
            assert net.last_chunk is not None, "an open first-layer accumulation outside the trainer's chunk loop"
            phase = (0 if self.open else 1) | (2 if net.last_chunk else 0)
        wst = net.ws.get(self.pfx + "conv_obs_bwd", wsz)
        if self.open and wst is not self._open_ws:  # (chunks only shrink towards the tail; a grown buffer would have dropped the sums)
            raise hip.HipError("first-layer accumulation: the workspace was reallocated under an open accumulation")
        if phase != 3:
            self.open = not net.last_chunk
            self._open_ws = wst if self.open else None
        g = net._g
        hip.conv2d_obs_bwd(desc1, src.data_ptr(), is_u8, mean.data_ptr(), rstd.data_ptr(), net._p(f"{self.ln.prefix}.weight"),
                           net._p(f"{self.ln.prefix}.bias"), net._p(f"{self.c1.prefix}.weight"), dz_ptr, g(f"{self.c1.prefix}.weight"),
                           g(f"{self.c1.prefix}.bias"), g(f"{self.ln.prefix}.weight"), g(f"{self.ln.prefix}.bias"),
                           wst.data_ptr(), channels_last=True, row_index=row_index, phase=phase,
                           dz_absmax_ptr=dz_absmax_ptr)

    # The Linear's weight gradient on the pre-split operands both products of the data-gradient chain read anyway (dyh, a3) through
    # csrc/h2tn.h (round 6: DMA in, transposing reads out); SRL_FC_WGRAD_TN=0: round 3's kernel on the float32 dy (A/B).
    FC_WGRAD_TN = os.environ.get("SRL_FC_WGRAD_TN", "1") != "0"

    def _fc_wgrad(self, n, dy, dyh, a3, tag):
        net, g = self.net, self.net._g
        H = self.H
        side = net._side_stream is not None and torch.cuda.current_stream() == net._side_stream
        gb = g(f"{self.fc.prefix}.bias")
        if self.FC_WGRAD_TN:
            wsp = net.ws.get("h2tn_side" if side else "h2tn", hip.h2_wgrad_dense_workspace(n, H, 3136)).data_ptr()
            hip.h2_wgrad_dense(dyh, a3, self._slot(S_DY, tag), self._slot(S_A3, tag), n, H, 3136, wsp, g(f"{self.fc.prefix}.weight"))
            if H > 2048:   # (else: summed by the pack launch, `backward`)
                hip.colsum(dy.ptr, dy.ld, n, H, gb, accumulate=True)
            return
        tiles = ((H + 127) // 128) * ((3136 + 127) // 128)
        from srl_amd.algorithm.hipnet import _split_for
        split = _split_for(n, tiles)
        wsp = net.ws.get("splitk_side" if side else "splitk", split * H * 3136).data_ptr() if split > 1 else None
        fused = hip.gemm_colsum_ok(H, 3136, n, dy.ptr, dy.ld, a3, 3136, 1)
        hip.gemm(H, 3136, n, dy.ptr, dy.ld, 1, a3, 3136, 1, g(f"{self.fc.prefix}.weight"), 3136, accumulate=True, split_k=split,
                 workspace=wsp, a_colsum=gb if fused else None, a_absmax=self._slot(M_DY, tag), b_h2_scale=self._slot(S_A3, tag))
        if not fused:
            hip.colsum(dy.ptr, dy.ld, n, H, gb, accumulate=True)

    def prefixes(self):
        return [self.ln.prefix, self.c1.prefix, self.c2.prefix, self.c3.prefix, self.fc.prefix]
