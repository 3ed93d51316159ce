"""Parity of the pre-split ("h2") kernels of round 4 -- csrc/h2gemm.h, csrc/h2conv.h, the h2 output of the first layer --
through the C ABI (`srl_h2_*`, `srl_conv2d_obs_fwd_h2`), against float64 torch references of the operations they replace:
nn.Conv2d forward / its data and weight gradients (legacy/algorithm/modules/cnn.py:93-135) and nn.Linear forward / data
gradient (modules/utils.py:154-161) as autograd forms them.

Tolerance: a contraction of K terms on two f16 pieces per operand carries every product to 2^-22 relative and accumulates in
float32 (gemm_bf16x3.h); results are compared at 2e-6 of the tensor's largest magnitude, the bar the round-3 two-piece
kernels are held to.  Sizes: small ragged batches in full, the benchmark's 16 384-image chunks on sampled images.  Every
kernel is also run three times on the same inputs and must return the same bits: a scheduling hazard (one was found in this
round's development: a 128-bit buffer store whose data registers were overwritten one instruction later) shows up as a value
that differs from run to run long before it shows up against a tolerance."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _hip():
    from srl_amd import hip
    hip.require_gpu()
    return hip


def _f(*shape, seed=0, relu=False, amp=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = (torch.rand(*shape, device=DEV, generator=g) * 2 - 1) * amp
    return torch.where(x < 0, torch.zeros_like(x), x * x * 3) if relu else x


def _slot(*vals):
    return torch.tensor(list(vals) if vals else [0.0], dtype=torch.float32, device=DEV)


def _absmax(hip, x):
    s = _slot(0.0)
    hip.absmax(x.data_ptr(), x.numel(), s.data_ptr())
    return s


def _close(got, ref, tol=2e-6):
    ref = ref.to(torch.float64)
    err = float((got.to(torch.float64) - ref).abs().max())
    assert err <= tol * max(float(ref.abs().max()), 1e-30), (err, float(ref.abs().max()))


class Packed:
    """An activation in one of the h2 layouts with its scale slot."""

    def __init__(self, hip, x_nhwc, layout):
        n, H, W, C = x_nhwc.shape
        self.buf = torch.empty(n * H * W * C, dtype=torch.float32, device=DEV)
        self.scale = _slot(0.0)
        self.amax = _absmax(hip, x_nhwc)
        hip.h2_pack_image(x_nhwc.data_ptr(), n, H, W, C, layout, self.buf.data_ptr(), absmax=self.amax.data_ptr(),
                          scale_out=self.scale.data_ptr())


def _weights(hip, w, rows, K, mode, desc=None):
    """h2p weights + (scale, row-norm) slots.  w: the float32 tensor as the layer stores it."""
    amax = _absmax(hip, w)
    dst = torch.empty(rows * K, dtype=torch.float32, device=DEV)
    s, r = _slot(0.0), _slot(0.0)
    hip.h2_weights(w.data_ptr(), rows, K, mode, amax.data_ptr(), s.data_ptr(), r.data_ptr(), dst.data_ptr(), desc=desc)
    return dst, s, r


def _mask_h2(act_nhwc):
    """Sign bytes in h2 order of a [n, H, W, C] activation: byte (pixel, block, group), bit j = element j of the group."""
    n, H, W, C = act_nhwc.shape
    a = (act_nhwc > 0).reshape(n * H * W, C // 32, 32)
    idx = torch.tensor([[4 * (g >> 1) + 16 * (g & 1) + (j & 3) + 8 * (j >> 2) for j in range(8)] for g in range(4)], device=DEV)
    bits = a[:, :, idx]                       # [pix, blk, g, j]
    w = (2 ** torch.arange(8, device=DEV)).to(torch.int32)
    return (bits.to(torch.int32) * w).sum(-1).to(torch.uint8).contiguous()


def _mask_natural(act_nhwc):
    n, H, W, C = act_nhwc.shape
    a = (act_nhwc > 0).reshape(-1, 32).to(torch.int64)
    w = 2 ** torch.arange(32, device=DEV, dtype=torch.int64)
    v = (a * w).sum(-1)
    return torch.where(v >= 2**31, v - 2**32, v).to(torch.int32).contiguous()


# ------------------------------------------------------------------------------------------------ formats
def test_pack_unpack_round_trip_every_layout():
    hip = _hip()
    x = _f(37, 64, seed=1, amp=3.0)
    amax = _absmax(hip, x)
    buf, back, s = torch.empty_like(x), torch.empty_like(x), _slot(0.0)
    hip.h2_pack_rows(x.data_ptr(), 64, 37, 64, buf.data_ptr(), absmax=amax.data_ptr(), scale_out=s.data_ptr())
    hip.h2_unpack_rows(buf.data_ptr(), 37, 64, s.data_ptr(), back.data_ptr(), 64)
    scale = float(s.item())
    assert scale == 2.0 ** round(np.log2(scale)) and float(amax.item()) * scale < 2**15  # a power of two, no f16 overflow
    # two f16 pieces: 22 bits of every element that is within 2^11 of the scaled bound, 2^-25 / scale absolute below
    assert float((back - x).abs().max()) <= max(2.0**-22 * float(x.abs().max()), 2.0**-24 / scale)
    img = _f(5, 20, 20, 32, seed=2, relu=True)
    for layout, unpack_layout in ((0, 0), (1, 1)):
        p = Packed(hip, img, layout)
        out = torch.empty_like(img)
        hip.h2_unpack_image(p.buf.data_ptr(), 5, 20, 20, 32, unpack_layout, p.scale.data_ptr(), out.data_ptr())
        assert float((out - img).abs().max()) <= 2.0**-21 * float(img.abs().max())
    # parity-class order = a permutation of the raster rows
    p2, p1 = Packed(hip, img, 2), Packed(hip, img, 1)
    yy, xx = np.meshgrid(np.arange(20), np.arange(20), indexing="ij")
    ent = torch.from_numpy((((yy & 1) * 2 + (xx & 1)) * 100 + (yy >> 1) * 10 + (xx >> 1)).reshape(-1)).to(DEV)
    assert torch.equal(p2.buf.view(5, 400, 32)[:, ent], p1.buf.view(5, 400, 32))


def _pick(n, count=48):
    """All images of a small batch; of a benchmark-size chunk (n = 16 384) 48 sampled ones plus both ends -- the float64
    reference is then computed for those only."""
    if n <= 2048:
        return torch.arange(n, device=DEV)
    idx = torch.from_numpy(np.random.default_rng(n).choice(n, count, replace=False)).to(DEV)
    return torch.cat([idx, torch.tensor([0, 1, n - 2, n - 1], device=DEV)])


BENCH_N = 16384  # images per chunk of the benchmarked update (bench.py --chunk-rows)


# ------------------------------------------------------------------------------------------------ forward convolutions
def _conv_case(kind):
    return dict(c2=(20, 32, 4, 2, 9), c3=(9, 64, 3, 1, 7))[kind]


@pytest.mark.parametrize("kind,n", [("c2", 5), ("c2", 301), ("c3", 7), ("c3", 301), ("c2", BENCH_N), ("c3", BENCH_N)])
def test_conv_forward_vs_float64(kind, n):
    hip = _hip()
    hip.dispatch_tiles(reset=True)
    H, C, k, st, OH = _conv_case(kind)
    x = _f(n, H, H, C, seed=3, relu=True, amp=2.0)
    w = _f(64, k, k, C, seed=4, amp=0.05)     # [Cout, KH, KW, Cin]: the layout the layers keep
    b = _f(64, seed=5, amp=0.1)
    xp = Packed(hip, x, 2 if kind == "c2" else 0)
    wp, sw, rw = _weights(hip, w, 64, k * k * C, 0)
    out = torch.zeros(n * OH * OH * 64, device=DEV)
    mask = torch.zeros(n * OH * OH * 8, dtype=torch.uint8, device=DEV)
    osc, oam, bb = _slot(0.0), _slot(0.0), _absmax(hip, b)
    run = lambda: hip.h2_conv(hip.H2_CONV2_FWD if kind == "c2" else hip.H2_CONV3_FWD, xp.buf.data_ptr(), wp.data_ptr(), xp.scale.data_ptr(),
                              sw.data_ptr(), n, out.data_ptr(), oam.data_ptr(), bias=b.data_ptr(), act=1, out_scale=osc.data_ptr(),
                              bound_in=xp.amax.data_ptr(), bound_w=rw.data_ptr(), bound_b=bb.data_ptr(), mask_out=mask.data_ptr())
    run()
    assert hip.dispatch_tiles(reset=True) == {f"h2:conv:{0 if kind == 'c2' else 1}:s3": 1}
    first = (out.clone(), mask.clone())
    pick = _pick(n)
    ref = F.relu(F.conv2d(x[pick].double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(), stride=st)).permute(0, 2, 3, 1).contiguous()
    got = torch.empty(n, OH, OH, 64, device=DEV)
    if kind == "c2":
        hip.h2_unpack_image(out.data_ptr(), n, OH, OH, 64, 0, osc.data_ptr(), got.data_ptr())
    else:
        hip.h2_unpack_rows(out.data_ptr(), n * OH * OH, 64, osc.data_ptr(), got.data_ptr(), 64)
    _close(got[pick], ref)
    # the bound really bounds, the measured range is the range, the sign bytes are the signs (up to values at rounding distance of 0)
    top = float(got.abs().max())
    assert top * float(osc.item()) < 2**15 and abs(float(oam.item()) - top) <= 1e-5 * top and float(ref.abs().max()) <= top * (1 + 1e-5)
    want_mask = _mask_h2(ref.float())
    assert int((mask.view(n, -1)[pick] != want_mask.reshape(len(pick), -1)).sum()) <= 2
    for _ in range(2):   # bit-reproducible
        run()
        assert torch.equal(out, first[0]) and torch.equal(mask, first[1])


# ------------------------------------------------------------------------------------------------ data gradients
@pytest.mark.parametrize("kind,n", [("c3", 3), ("c3", 300), ("c2", 2), ("c2", 6), ("c2", 300), ("c3", BENCH_N), ("c2", BENCH_N)])
def test_conv_data_gradient_vs_float64(kind, n):
    hip = _hip()
    hip.dispatch_tiles(reset=True)
    H, C, k, st, OH = _conv_case(kind)
    dz = _f(n, OH, OH, 64, seed=6, amp=1e-3)
    w = _f(64, k, k, C, seed=7, amp=0.05)
    act = _f(n, H, H, C, seed=8, relu=True)          # the forward activation whose ReLU derivative gates dx
    desc = hip.conv_desc(1, H, H, C, k, k, st, 64, 1)
    rows, K = st * st * C, (k // st) * (k // st) * 64
    wg, sw, rw = _weights(hip, w, rows, K, 2, desc)
    if kind == "c3":
        zp = torch.empty_like(dz).reshape(-1)
        sz, az = _slot(0.0), _absmax(hip, dz)
        hip.h2_pack_rows(dz.data_ptr(), 64, n * OH * OH, 64, zp.data_ptr(), absmax=az.data_ptr(), scale_out=sz.data_ptr())
        mask = _mask_h2(act)
    else:
        p = Packed(hip, dz, 0)
        zp, sz, az = p.buf, p.scale, p.amax
        mask = _mask_natural(act)
    out = torch.zeros(n * H * H * C, device=DEV)
    osc, oam = _slot(0.0), _slot(0.0)
    run = lambda: hip.h2_conv(hip.H2_CONV3_DGRAD if kind == "c3" else hip.H2_CONV2_DGRAD, zp.data_ptr(), wg.data_ptr(), sz.data_ptr(),
                              sw.data_ptr(), n, out.data_ptr(), oam.data_ptr(), out_scale=osc.data_ptr(), bound_in=az.data_ptr(),
                              bound_w=rw.data_ptr(), mask_in=mask.data_ptr())
    run()
    assert hip.dispatch_tiles(reset=True) == {f"h2:conv:{2 if kind == 'c3' else 3}:s2": 1}
    first = out.clone()
    pick = _pick(n)
    ref = F.conv_transpose2d(dz[pick].double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=st).permute(0, 2, 3, 1) * (act[pick] > 0)
    if kind == "c3":
        got = torch.empty(n, H, H, C, device=DEV)
        hip.h2_unpack_image(out.data_ptr(), n, H, H, C, 0, osc.data_ptr(), got.data_ptr())
    else:
        got = out.view(n, H, H, C)
    _close(got[pick], ref)
    for _ in range(3):
        run()
        assert torch.equal(out, first)


# ------------------------------------------------------------------------------------------------ weight gradients
@pytest.mark.parametrize("kind,n", [("c3", 5), ("c3", 293), ("c2", 5), ("c2", 293), ("c3", BENCH_N), ("c2", BENCH_N)])
def test_conv_weight_gradient_vs_float64(kind, n):
    """(n = 16 384: every slab of the 256-workgroup launch and the reduction over them; 1.3 M / 0.8 M terms per element.)"""
    hip = _hip()
    hip.dispatch_tiles(reset=True)
    H, C, k, st, OH = _conv_case(kind)
    x = _f(n, H, H, C, seed=9, relu=True, amp=2.0)
    dz = _f(n, OH, OH, 64, seed=10, amp=1e-3)
    xp = Packed(hip, x, 2 if kind == "c2" else 0)
    if kind == "c2":
        zp = Packed(hip, dz, 0)
        zbuf, sz = zp.buf, zp.scale
    else:
        zbuf, sz, az = torch.empty_like(dz).reshape(-1), _slot(0.0), _absmax(hip, dz)
        hip.h2_pack_rows(dz.data_ptr(), 64, n * OH * OH, 64, zbuf.data_ptr(), absmax=az.data_ptr(), scale_out=sz.data_ptr())
    code = hip.H2_WGRAD_CONV2 if kind == "c2" else hip.H2_WGRAD_CONV3
    ws = torch.empty(hip.h2_wgrad_workspace(code), device=DEV)
    K = k * k * C
    gw0, gb0 = _f(64, K, seed=11, amp=1e-2), _f(64, seed=12, amp=1e-2)   # the kernels ADD into the gradient buffers
    gw, gb = gw0.clone(), gb0.clone()
    hip.h2_wgrad(code, xp.buf.data_ptr(), zbuf.data_ptr(), xp.scale.data_ptr(), sz.data_ptr(), n, ws.data_ptr(), gw.data_ptr(), gb.data_ptr())
    assert hip.dispatch_tiles(reset=True) == {f"h2:wgrad:{0 if kind == 'c2' else 1}:s{2 if kind == 'c2' else 3}": 1}
    wd = torch.zeros(64, C, k, k, dtype=torch.float64, device=DEV, requires_grad=True)
    bd = torch.zeros(64, dtype=torch.float64, device=DEV, requires_grad=True)
    for i0 in range(0, n, 2048):   # the float64 reference in pieces
        (F.conv2d(x[i0:i0 + 2048].double().permute(0, 3, 1, 2), wd, bd, stride=st) * dz[i0:i0 + 2048].double().permute(0, 3, 1, 2)).sum().backward()
    tol = 1e-6 if n <= 2048 else 4e-6
    _close(gw - gw0, wd.grad.permute(0, 2, 3, 1).reshape(64, K), tol=tol)
    _close(gb - gb0, bd.grad, tol=tol)
    gw2, gb2 = gw0.clone(), gb0.clone()
    hip.h2_wgrad(code, xp.buf.data_ptr(), zbuf.data_ptr(), xp.scale.data_ptr(), sz.data_ptr(), n, ws.data_ptr(), gw2.data_ptr(), gb2.data_ptr())
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)


# ------------------------------------------------------------------------------------------------ the Linear's products
@pytest.mark.parametrize("M", [77, 1000, BENCH_N])
def test_linear_forward_and_data_gradient_vs_float64(M):
    hip = _hip()
    hip.dispatch_tiles(reset=True)
    K, N = 3136, 512
    x = _f(M, K, seed=13, relu=True, amp=2.0)
    w = _f(N, K, seed=14, amp=0.03)
    b = _f(N, seed=15, amp=0.1)
    xbuf, sx, ax = torch.empty_like(x), _slot(0.0), _absmax(hip, x)
    hip.h2_pack_rows(x.data_ptr(), K, M, K, xbuf.data_ptr(), absmax=ax.data_ptr(), scale_out=sx.data_ptr())
    wp, sw, rw = _weights(hip, w, N, K, 0)
    y = torch.zeros(M, N, device=DEV)
    mo = torch.zeros(M * N // 32, dtype=torch.int32, device=DEV)
    hip.h2_gemm(xbuf.data_ptr(), wp.data_ptr(), sx.data_ptr(), sw.data_ptr(), M, N, K, y.data_ptr(), bias=b.data_ptr(), act=1, mask_out=mo.data_ptr())
    assert hip.dispatch_tiles(reset=True) == {"h2:gemm:4:s3": 1}
    ref = F.relu(x.double() @ w.double().t() + b.double())
    _close(y, ref)
    assert int((mo != _mask_natural(ref.float().view(M, 1, 1, N))).sum()) <= 1
    # data gradient: dx = (dy W) * relu'(x), dx as h2p rows, the derivative from x's sign bytes in h2 order
    dy = _f(M, N, seed=16, amp=1e-3)
    dbuf, sd, ad = torch.empty_like(dy), _slot(0.0), _absmax(hip, dy)
    hip.h2_pack_rows(dy.data_ptr(), N, M, N, dbuf.data_ptr(), absmax=ad.data_ptr(), scale_out=sd.data_ptr())
    wt, swt, rwt = _weights(hip, w, K, N, 1)   # transposed: rows = input features
    mask = _mask_h2(x.view(M, 1, 1, K))
    dx = torch.zeros(M * K, device=DEV)
    osc, oam = _slot(0.0), _slot(0.0)
    run = lambda: hip.h2_gemm(dbuf.data_ptr(), wt.data_ptr(), sd.data_ptr(), swt.data_ptr(), M, K, N, dx.data_ptr(), out_h2=True,
                              out_scale=osc.data_ptr(), bound_in=ad.data_ptr(), bound_w=rwt.data_ptr(), out_absmax=oam.data_ptr(),
                              mask_in=mask.data_ptr(), mask_in_h2order=True)
    hip.dispatch_tiles(reset=True)
    run()
    assert hip.dispatch_tiles(reset=True) == {"h2:gemm:8:s2": 1}   # wide output, short reduction: 256 channels per workgroup
    first = dx.clone()
    got = torch.empty(M, K, device=DEV)
    hip.h2_unpack_rows(dx.data_ptr(), M, K, osc.data_ptr(), got.data_ptr(), K)
    _close(got, (dy.double() @ w.double()) * (x > 0))
    run()
    assert torch.equal(dx, first)


@pytest.mark.parametrize("M,ks", [(77, 3), (2048, 8), (4096, 4), (1000, 98)])
def test_linear_forward_split_over_workgroups_and_finished_by_its_consumer(M, ks):
    """srl_h2_gemm_splitk: the reduction of the Linear forward split over `ks` workgroups per tile, raw partial sums in [ks][M][N]
    slabs -- their sum equals the unsplit product -- and srl_ln_heads_fwd's x_slabs finishing them (slabs added, bias, ReLU) while
    it reads: heads' outputs and statistics equal those on the finished product (the inference path of `H2Cnn.forward`)."""
    hip = _hip()
    K, N = 3136, 512
    x = _f(M, K, seed=23, relu=True, amp=2.0)
    w = _f(N, K, seed=24, amp=0.03)
    b = _f(N, seed=25, amp=0.1)
    xbuf, sx, ax = torch.empty_like(x), _slot(0.0), _absmax(hip, x)
    hip.h2_pack_rows(x.data_ptr(), K, M, K, xbuf.data_ptr(), absmax=ax.data_ptr(), scale_out=sx.data_ptr())
    wp, sw, rw = _weights(hip, w, N, K, 0)
    y = torch.zeros(M, N, device=DEV)
    hip.h2_gemm(xbuf.data_ptr(), wp.data_ptr(), sx.data_ptr(), sw.data_ptr(), M, N, K, y.data_ptr(), bias=b.data_ptr(), act=1)
    slabs = torch.full((ks, M, N), float("nan"), device=DEV)
    hip.h2_gemm_splitk(xbuf.data_ptr(), wp.data_ptr(), sx.data_ptr(), sw.data_ptr(), M, N, K, slabs.data_ptr(), ks)
    fin = F.relu(slabs.double().sum(0) + b.double())
    _close(fin.float(), F.relu(x.double() @ w.double().t() + b.double()))
    assert float((fin.float() - y).abs().max()) <= 1e-5 * float(y.abs().max())
    heads = (6, 1)
    g, be = 1 + 0.1 * _f(N, seed=26), 0.1 * _f(N, seed=27)
    W, hb = [_f(a, N, seed=28 + i, amp=0.05) for i, a in enumerate(heads)], [_f(a, seed=30 + i, amp=0.1) for i, a in enumerate(heads)]
    outs = []
    for xin, kw in ((y, {}), (slabs, dict(x_slabs=ks, x_slab_stride=M * N, x_bias=b.data_ptr(), x_act=1))):
        o = [torch.full((M, a), float("nan"), device=DEV) for a in heads]
        mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        hip.ln_heads_fwd(xin.data_ptr(), N, M, N, g.data_ptr(), be.data_ptr(), [t.data_ptr() for t in W], [t.data_ptr() for t in hb],
                         list(heads), [t.data_ptr() for t in o], list(heads), mean.data_ptr(), rstd.data_ptr(), **kw)
        outs.append(o + [mean, rstd])
    for a_, b_ in zip(*outs):
        assert float((a_ - b_).abs().max()) <= 2e-5 * max(float(a_.abs().max()), 1e-6)


@pytest.mark.parametrize("M", [1111, BENCH_N])
def test_linear_weight_gradient_reads_h2p_rows(M):
    """dW = dy^T x with x never written as float32: the round-3 two-piece kernel stages the h2p rows as they are
    (srl_gemm_desc::b_h2_scale).  With the split-K factor, the workspace and the fused bias sum `H2Cnn._fc_wgrad` passes for that
    row count (16 384 rows: the 128 x 128 tiles over 5 k-ranges the benchmark launches)."""
    from srl_amd.algorithm.hipnet import _split_for
    hip = _hip()
    K, N = 3136, 512
    x = _f(M, K, seed=17, relu=True, amp=2.0)
    dy = _f(M, N, seed=18, amp=1e-3)
    xbuf, sx, ax = torch.empty_like(x), _slot(0.0), _absmax(hip, x)
    hip.h2_pack_rows(x.data_ptr(), K, M, K, xbuf.data_ptr(), absmax=ax.data_ptr(), scale_out=sx.data_ptr())
    ad = _absmax(hip, dy)
    gw, gb = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
    split = _split_for(M, ((N + 127) // 128) * ((K + 127) // 128))
    ws = torch.empty(max(split * N * K, 4), device=DEV)
    fused = hip.gemm_colsum_ok(N, K, M, dy.data_ptr(), N, xbuf.data_ptr(), K, 1)
    hip.dispatch_counts(reset=True)
    hip.dispatch_tiles(reset=True)
    hip.gemm(N, K, M, dy.data_ptr(), N, 1, xbuf.data_ptr(), K, 1, gw.data_ptr(), K, accumulate=True, split_k=split,
             workspace=ws.data_ptr() if split > 1 else None, a_colsum=gb.data_ptr() if fused else None, a_absmax=ad.data_ptr(),
             b_h2_scale=sx.data_ptr())
    assert hip.dispatch_counts(reset=True)["gemm2h"] == 1
    tiles = hip.dispatch_tiles(reset=True)
    assert list(tiles) == [f"gemm2h:128x128:k{split}:f3"], tiles
    if M == BENCH_N:
        assert split == 5 and fused
    _close(gw, dy.double().t() @ x.double(), tol=2e-6 if M < 2048 else 4e-6)
    if fused:
        _close(gb, dy.double().sum(0), tol=4e-6)


@pytest.mark.parametrize("M,N,K", [(1111, 512, 3136), (BENCH_N, 512, 3136), (37, 256, 256), (1000, 96, 160), (2085, 320, 1056)])
def test_dense_weight_gradient_on_presplit_operands(M, N, K):
    """dW = dy^T x with BOTH operands as h2p rows (csrc/h2tn.h, round 6: LDS-DMA in, transposing reads out, slabs per row range added
    in a fixed order): what `H2Cnn._fc_wgrad` launches for the benchmarked chunk (16 384 rows: 26 tiles x 9 row ranges), ragged row
    counts (a last k-step of fewer than 16 rows, a last range shorter than the others), channel counts that are not multiples of the
    256-channel tile, accumulate on / off, and bit-reproducibility."""
    hip = _hip()
    x = _f(M, K, seed=17, relu=True, amp=2.0)
    dy = _f(M, N, seed=18, amp=1e-3)
    xbuf, sx, ax = torch.empty_like(x), _slot(0.0), _absmax(hip, x)
    hip.h2_pack_rows(x.data_ptr(), K, M, K, xbuf.data_ptr(), absmax=ax.data_ptr(), scale_out=sx.data_ptr())
    dbuf, sd, ad = torch.empty_like(dy), _slot(0.0), _absmax(hip, dy)
    hip.h2_pack_rows(dy.data_ptr(), N, M, N, dbuf.data_ptr(), absmax=ad.data_ptr(), scale_out=sd.data_ptr())
    ws = torch.full((max(hip.h2_wgrad_dense_workspace(M, N, K), 4),), float("nan"), device=DEV)
    ref = dy.double().t() @ x.double()
    tol = 2e-6 if M < 2048 else 4e-6
    g0 = torch.full((N, K), float("nan"), device=DEV)
    hip.dispatch_tiles(reset=True)
    hip.h2_wgrad_dense(dbuf.data_ptr(), xbuf.data_ptr(), sd.data_ptr(), sx.data_ptr(), M, N, K, ws.data_ptr(), g0.data_ptr(), accumulate=False)
    assert list(hip.dispatch_tiles(reset=True)) == ["h2:tn:8:s4"]
    _close(g0, ref, tol=tol)
    base = _f(N, K, seed=19, amp=float(ref.abs().max()))
    g1 = base.clone()
    hip.h2_wgrad_dense(dbuf.data_ptr(), xbuf.data_ptr(), sd.data_ptr(), sx.data_ptr(), M, N, K, ws.data_ptr(), g1.data_ptr(), accumulate=True)
    _close(g1, ref + base.double(), tol=tol)
    g2 = torch.zeros(N, K, device=DEV)
    ws.fill_(float("nan"))
    hip.h2_wgrad_dense(dbuf.data_ptr(), xbuf.data_ptr(), sd.data_ptr(), sx.data_ptr(), M, N, K, ws.data_ptr(), g2.data_ptr(), accumulate=False)
    assert torch.equal(g0, g2)


def test_persistent_wide_product_every_epilogue():
    """The cases below in a child process with SRL_H2GEMM_P=1 (the library reads the switch once: by default the persistent kernel
    takes only the large products, `srl_h2_gemm`'s `big`; the wide data gradient of the Atari Linear stays on one tile per
    workgroup -- see h2.hip)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests'); import test_gpu_h2 as t\n"
            "for c in [(300, 1024, 64), (1111, 1056, 512), (2000, 3136, 32), (4133, 1280, 1024)]: t._persistent_wide_case(*c)\nprint('ok')") % (root, root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SRL_H2GEMM_P="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]


def test_persistent_kernel_takes_the_large_products_by_default():
    """Default dispatch: M >= 4096, NC >= 2048, K >= 1024 (the football tower's layers) -> csrc/h2gemmp.h; checked against float64 on
    sampled rows."""
    hip = _hip()
    M, N, K = 4200, 2080, 1056
    x = _f(M, K, seed=61, relu=True, amp=2.0)
    w = _f(N, K, seed=62, amp=0.03)
    b = _f(N, seed=63, amp=0.1)
    xbuf, sx, ax = torch.empty_like(x), _slot(0.0), _absmax(hip, x)
    hip.h2_pack_rows(x.data_ptr(), K, M, K, xbuf.data_ptr(), absmax=ax.data_ptr(), scale_out=sx.data_ptr())
    wp, sw, rw = _weights(hip, w, N, K, 0)
    y = torch.full((M, N), float("nan"), device=DEV)
    hip.dispatch_tiles(reset=True)
    hip.h2_gemm(xbuf.data_ptr(), wp.data_ptr(), sx.data_ptr(), sw.data_ptr(), M, N, K, y.data_ptr(), bias=b.data_ptr(), act=1)
    assert hip.dispatch_tiles(reset=True) == {"h2:gemmp:8:s4": 1}
    _close(y, F.relu(x.double() @ w.double().t() + b.double()))


def _persistent_wide_case(M, N, K):
    """csrc/h2gemmp.h (round 6: persistent workgroups, a flat ring across tiles, stores not waited for) through `srl_h2_gemm`'s wide
    case, every epilogue it carries: float32 output with bias + ReLU + sign words out; h2p output with bias + ReLU (the bound's
    bias term); the data-gradient form with the ReLU derivative in natural order; ragged rows, a narrow last channel tile, one
    pair of k-steps per tile (K = 32: a tile boundary at every barrier), tiles with more than one visit per workgroup."""
    hip = _hip()
    x = _f(M, K, seed=41, relu=True, amp=2.0)
    w = _f(N, K, seed=42, amp=0.03)
    b = _f(N, seed=43, amp=0.1)
    xbuf, sx, ax = torch.empty_like(x), _slot(0.0), _absmax(hip, x)
    hip.h2_pack_rows(x.data_ptr(), K, M, K, xbuf.data_ptr(), absmax=ax.data_ptr(), scale_out=sx.data_ptr())
    wp, sw, rw = _weights(hip, w, N, K, 0)
    ref = F.relu(x.double() @ w.double().t() + b.double())
    # float32 out, bias, ReLU, sign words
    y = torch.full((M, N), float("nan"), device=DEV)
    mo = torch.zeros(M * N // 32, dtype=torch.int32, device=DEV)
    oam = _slot(0.0)
    hip.dispatch_tiles(reset=True)
    hip.h2_gemm(xbuf.data_ptr(), wp.data_ptr(), sx.data_ptr(), sw.data_ptr(), M, N, K, y.data_ptr(), bias=b.data_ptr(), act=1,
                mask_out=mo.data_ptr(), out_absmax=oam.data_ptr())
    assert hip.dispatch_tiles(reset=True) == {"h2:gemmp:8:s4": 1}   # (SRL_H2GEMM_P=1, set below: the persistent kernel for wide products too)
    _close(y, ref)
    assert int((mo != _mask_natural(y.view(M, 1, 1, N))).sum()) == 0
    assert abs(float(oam) - float(y.abs().max())) <= 1e-6 * float(y.abs().max())
    # h2p out with the bias in the bound
    yh = torch.zeros(M * N, device=DEV)
    osc, bb = _slot(0.0), _absmax(hip, b)
    hip.h2_gemm(xbuf.data_ptr(), wp.data_ptr(), sx.data_ptr(), sw.data_ptr(), M, N, K, yh.data_ptr(), bias=b.data_ptr(), act=1, out_h2=True,
                out_scale=osc.data_ptr(), bound_in=ax.data_ptr(), bound_w=rw.data_ptr(), bound_b=bb.data_ptr())
    got = torch.empty(M, N, device=DEV)
    hip.h2_unpack_rows(yh.data_ptr(), M, N, osc.data_ptr(), got.data_ptr(), N)
    _close(got, ref, tol=4e-6)
    # data-gradient form, derivative bits in natural order
    act = _f(M, N, seed=44, relu=True)
    mi = _mask_natural(act.view(M, 1, 1, N))
    dx = torch.zeros(M * N, device=DEV)
    oam2 = _slot(0.0)
    hip.h2_gemm(xbuf.data_ptr(), wp.data_ptr(), sx.data_ptr(), sw.data_ptr(), M, N, K, dx.data_ptr(), out_h2=True, out_scale=osc.data_ptr(),
                bound_in=ax.data_ptr(), bound_w=rw.data_ptr(), out_absmax=oam2.data_ptr(), mask_in=mi.data_ptr(), mask_in_h2order=False)
    hip.h2_unpack_rows(dx.data_ptr(), M, N, osc.data_ptr(), got.data_ptr(), N)
    ref2 = (x.double() @ w.double().t()) * (act > 0)
    _close(got, ref2)
    assert abs(float(oam2) - float(got.abs().max())) <= 2e-6 * float(got.abs().max())


@pytest.mark.parametrize("rows,C", [(1, 32), (37, 96), (1111, 512), (BENCH_N, 512), (300, 2048)])
def test_pack_rows_with_column_sums(rows, C):
    """srl_h2_pack_rows_colsum (round 6): the same pieces as srl_h2_pack_rows, bit for bit, and the column sums of the source added
    into / written to the bias gradient (slabs per workgroup added in a fixed order: bit-reproducible)."""
    hip = _hip()
    x = _f(rows, C, seed=51, amp=1e-3)
    ax = _absmax(hip, x)
    a, b, sa, sb = torch.zeros_like(x), torch.zeros_like(x), _slot(0.0), _slot(0.0)
    hip.h2_pack_rows(x.data_ptr(), C, rows, C, a.data_ptr(), absmax=ax.data_ptr(), scale_out=sa.data_ptr())
    ws = torch.full((max(hip.h2_pack_rows_colsum_workspace(rows, C), 4),), float("nan"), device=DEV)
    base = _f(C, seed=52)
    cs = base.clone()
    hip.h2_pack_rows_colsum(x.data_ptr(), C, rows, C, b.data_ptr(), ws.data_ptr(), cs.data_ptr(), absmax=ax.data_ptr(), scale_out=sb.data_ptr())
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and float(sa) == float(sb)
    ref = x.double().sum(0) + base.double()
    assert float((cs.double() - ref).abs().max()) <= 2e-6 * max(float(ref.abs().max()), 1e-30)
    cs2 = torch.full((C,), float("nan"), device=DEV)
    hip.h2_pack_rows_colsum(x.data_ptr(), C, rows, C, b.data_ptr(), ws.data_ptr(), cs2.data_ptr(), absmax=ax.data_ptr(), scale_out=sb.data_ptr(),
                            accumulate=False)
    cs3 = torch.zeros(C, device=DEV)
    hip.h2_pack_rows_colsum(x.data_ptr(), C, rows, C, b.data_ptr(), ws.data_ptr(), cs3.data_ptr(), absmax=ax.data_ptr(), scale_out=sb.data_ptr())
    assert torch.equal(cs2, cs3)


# ------------------------------------------------------------------------------------------------ benchmark-size launches
def test_benchmark_size_chunk_on_sampled_images():
    """One 16 384-image chunk -- the launches bench.py times -- through conv2 forward, its weight gradient and conv2's data
    gradient, checked against float64 on 48 sampled images (forward / data gradient) and in full (weight gradient)."""
    hip = _hip()
    n = 16384
    x = _f(n, 20, 20, 32, seed=19, relu=True, amp=2.0)
    w = _f(64, 4, 4, 32, seed=20, amp=0.05)
    b = _f(64, seed=21, amp=0.1)
    xp = Packed(hip, x, 2)
    wp, sw, rw = _weights(hip, w, 64, 512, 0)
    out = torch.zeros(n * 81 * 64, device=DEV)
    mask = torch.zeros(n * 81 * 8, dtype=torch.uint8, device=DEV)
    osc, oam, bb = _slot(0.0), _slot(0.0), _absmax(hip, b)
    hip.h2_conv(hip.H2_CONV2_FWD, xp.buf.data_ptr(), wp.data_ptr(), xp.scale.data_ptr(), sw.data_ptr(), n, out.data_ptr(), oam.data_ptr(),
                bias=b.data_ptr(), act=1, out_scale=osc.data_ptr(), bound_in=xp.amax.data_ptr(), bound_w=rw.data_ptr(),
                bound_b=bb.data_ptr(), mask_out=mask.data_ptr())
    got = torch.empty(n, 9, 9, 64, device=DEV)
    hip.h2_unpack_image(out.data_ptr(), n, 9, 9, 64, 0, osc.data_ptr(), got.data_ptr())
    pick = torch.from_numpy(np.random.default_rng(0).choice(n, 48, replace=False)).to(DEV)
    pick = torch.cat([pick, torch.tensor([0, 1, n - 2, n - 1], device=DEV)])
    ref = F.relu(F.conv2d(x[pick].double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(), stride=2)).permute(0, 2, 3, 1)
    _close(got[pick], ref)
    # weight gradient over the whole chunk (dz = a smooth function of the forward output keeps the reference cheap)
    dz = (got * 1e-3).contiguous()
    zp = Packed(hip, dz, 0)
    ws = torch.empty(hip.h2_wgrad_workspace(hip.H2_WGRAD_CONV2), device=DEV)
    gw, gb = torch.zeros(64, 512, device=DEV), torch.zeros(64, device=DEV)
    hip.h2_wgrad(hip.H2_WGRAD_CONV2, xp.buf.data_ptr(), zp.buf.data_ptr(), xp.scale.data_ptr(), zp.scale.data_ptr(), n, ws.data_ptr(),
                 gw.data_ptr(), gb.data_ptr())
    wd = torch.zeros(64, 32, 4, 4, dtype=torch.float64, device=DEV, requires_grad=True)
    for i0 in range(0, n, 2048):   # float64 reference in pieces
        (F.conv2d(x[i0:i0 + 2048].double().permute(0, 3, 1, 2), wd, stride=2) * dz[i0:i0 + 2048].double().permute(0, 3, 1, 2)).sum().backward()
    _close(gw, wd.grad.permute(0, 2, 3, 1).reshape(64, 512), tol=4e-6)   # 1.3 M terms per element
    _close(gb, dz.double().sum((0, 1, 2)), tol=4e-6)
    # data gradient
    desc = hip.conv_desc(1, 20, 20, 32, 4, 4, 2, 64, 1)
    wg, swg, rwg = _weights(hip, w, 128, 256, 2, desc)
    m1 = _mask_natural(x)
    dx = torch.zeros(n, 20, 20, 32, device=DEV)
    oam2 = _slot(0.0)
    hip.h2_conv(hip.H2_CONV2_DGRAD, zp.buf.data_ptr(), wg.data_ptr(), zp.scale.data_ptr(), swg.data_ptr(), n, dx.data_ptr(), oam2.data_ptr(),
                mask_in=m1.data_ptr())
    ref = F.conv_transpose2d(dz[pick].double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=2).permute(0, 2, 3, 1) * (x[pick] > 0)
    _close(dx[pick], ref)


# ------------------------------------------------------------------------------------------------ first layer
def test_first_layer_h2_output_equals_its_float32_output():
    """srl_conv2d_obs_fwd_h2 = srl_conv2d_obs_fwd with the result written as h2p rows in parity-class order; after unpacking
    the two agree to the pieces' precision, the sign words except where the value is at rounding distance of zero."""
    hip = _hip()
    n = 300
    g = torch.Generator(device=DEV).manual_seed(22)
    frames = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device=DEV, generator=g)
    s2d, mean, rstd = torch.empty(n, 21, 21, 64, dtype=torch.uint8, device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    hip.obs_space_to_depth(frames.data_ptr(), True, n, 4, 84, 84, 4, s2d.data_ptr(), mean.data_ptr(), rstd.data_ptr())
    desc = hip.conv_desc(n, 21, 21, 64, 2, 2, 1, 32, 1)
    gamma, beta = 1 + _f(21 * 21 * 64, seed=23, amp=0.2), _f(21 * 21 * 64, seed=24, amp=0.2)
    w, b = _f(32, 256, seed=25, amp=0.06), _f(32, seed=26, amp=0.1)
    ws = torch.empty(hip.conv2d_obs_fwd_workspace(desc), device=DEV)
    y = torch.empty(n, 400, 32, device=DEV)
    ym, yam = torch.zeros(n * 400, dtype=torch.int32, device=DEV), _slot(0.0)
    hip.conv2d_obs_fwd(desc, s2d.data_ptr(), True, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(), b.data_ptr(),
                       y.data_ptr(), channels_last=True, y_absmax=yam.data_ptr(), y_mask=ym.data_ptr(), ws_ptr=ws.data_ptr())
    yh = torch.zeros(n * 400 * 32, device=DEV)
    hm, ham, hs = torch.zeros(n * 400, dtype=torch.int32, device=DEV), _slot(0.0), _slot(0.0)
    hip.conv2d_obs_fwd_h2(desc, s2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(), b.data_ptr(),
                          yh.data_ptr(), hs.data_ptr(), ws.data_ptr(), None, ham.data_ptr(), hm.data_ptr(), reuse_folded=False, ent_order=2)
    back = torch.empty(n * 400, 32, device=DEV)
    hip.h2_unpack_rows(yh.data_ptr(), n * 400, 32, hs.data_ptr(), back.data_ptr(), 32)
    yy, xx = np.meshgrid(np.arange(20), np.arange(20), indexing="ij")
    ent = torch.from_numpy((((yy & 1) * 2 + (xx & 1)) * 100 + (yy >> 1) * 10 + (xx >> 1)).reshape(-1)).to(DEV)
    back = back.view(n, 400, 32)[:, ent]
    # (obs_h2.h: the folded weights as two f16 pieces, 2^-22 of a channel's largest weight each, where obs_bf16.h's three
    # bf16 pieces are exact -- both are checked against float64 in test_first_layer_at_benchmark_size)
    assert float((back - y).abs().max()) <= 2.0**-19 * float(y.abs().max())
    flips = (hm != ym)
    assert int(flips.sum()) <= 4 and float(y.view(n * 400, 32)[flips.view(-1)].abs().max() if flips.any() else 0.0) <= 1e-5
    assert abs(float(ham.item()) - float(yam.item())) <= 1e-5 * float(yam.item())
    assert float(y.abs().max()) * float(hs.item()) < 2**15   # the a-priori bound holds


def test_first_layer_two_streams_on_one_set_of_folded_weights():
    """The executors of one update share the folded first-layer weights (srl_conv2d_obs_fold_h2 writes them once per parameter
    version) and run srl_conv2d_obs_fwd_h2 side by side on their own streams: each call then needs its own room for the per-sample
    records (`records`) -- inside the shared workspace the launches overwrite each other's (frame slot, mean, rstd).  Ten rounds of
    two concurrent launches over different frames against the same launches one after the other: bit for bit; and (what does not
    depend on how the GPU happens to schedule the two) a launch with its own records leaves the shared workspace untouched."""
    hip = _hip()
    n, slots = 8192, 2 * 8192 + 64
    g = torch.Generator(device=DEV).manual_seed(41)
    frames = torch.randint(0, 256, (slots, 4, 84, 84), dtype=torch.uint8, device=DEV, generator=g)
    frames[::3] //= 8
    s2d, mean, rstd = torch.empty(slots, 21, 21, 64, dtype=torch.uint8, device=DEV), torch.empty(slots, device=DEV), torch.empty(slots, device=DEV)
    hip.obs_space_to_depth(frames.data_ptr(), True, slots, 4, 84, 84, 4, s2d.data_ptr(), mean.data_ptr(), rstd.data_ptr())
    del frames
    perm = torch.randperm(slots, device=DEV, generator=g).to(torch.int32)
    rows = [perm[:n].contiguous(), perm[n:2 * n].contiguous()]
    desc = hip.conv_desc(n, 21, 21, 64, 2, 2, 1, 32, 1)
    gamma, beta = 1 + _f(21, 21, 64, seed=42, amp=0.2), _f(21, 21, 64, seed=43, amp=0.2)
    w, b = _f(32, 2, 2, 64, seed=44, amp=0.06), _f(32, seed=45, amp=0.1)
    ws = torch.empty(hip.conv2d_obs_fwd_workspace(desc), device=DEV)
    assert hip.conv2d_obs_fold_h2(hip.conv_desc(1, 21, 21, 64, 2, 2, 1, 32, 1), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(), b.data_ptr(),
                                  ws.data_ptr())
    assert not hip.conv2d_obs_fold_h2(hip.conv_desc(1, 20, 20, 64, 4, 4, 2, 32, 1), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(),
                                      b.data_ptr(), ws.data_ptr())   # not the block kernel's layer: nothing written
    out = [[torch.zeros(n * 400 * 32, device=DEV), torch.zeros(n * 400, dtype=torch.int32, device=DEV), _slot(0.0), _slot(0.0),
            torch.empty(4 * (n + 32), device=DEV)] for _ in range(2)]

    def launch(e):
        yh, hm, ham, hs, rec = out[e]
        ham.zero_()
        hip.conv2d_obs_fwd_h2(desc, s2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(),
                              b.data_ptr(), yh.data_ptr(), hs.data_ptr(), ws.data_ptr(), rows[e], ham.data_ptr(), hm.data_ptr(),
                              reuse_folded=True, ent_order=2, records=rec.data_ptr())

    folded = ws.clone()
    for e in range(2):   # one after the other
        launch(e)
    torch.cuda.synchronize()
    assert torch.equal(ws.view(torch.int32), folded.view(torch.int32))   # nothing of a launch is kept in the shared workspace
    want = [(o[0].clone(), o[1].clone(), float(o[2].item()), float(o[3].item())) for o in out]
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    for _ in range(10):
        for o in out:
            o[0].zero_(); o[1].zero_()
        torch.cuda.synchronize()
        for e in range(2):
            with torch.cuda.stream(streams[e]):
                launch(e)
        torch.cuda.synchronize()
        for e in range(2):
            assert torch.equal(out[e][0], want[e][0]) and torch.equal(out[e][1], want[e][1]), e
            assert float(out[e][2].item()) == want[e][2] and float(out[e][3].item()) == want[e][3]


def test_first_layer_at_benchmark_size():
    """Both first-layer kernels on one 16 384-frame chunk exactly as `H2Cnn` calls them: frames read in place from a slot buffer
    through `row_index` (the HBM observation ring), forward with the h2p output in parity-class order + sign words + measured range,
    backward for dW / db / dgamma / dbeta.  Forward against float64 on sampled frames, the four gradients against float64 over
    the whole chunk (autograd in pieces of 1024 frames)."""
    hip = _hip()
    n, slots = BENCH_N, BENCH_N + 64
    g = torch.Generator(device=DEV).manual_seed(31)
    frames = torch.randint(0, 256, (slots, 4, 84, 84), dtype=torch.uint8, device=DEV, generator=g)
    frames[::5] //= 16   # dark frames: small variance, large rstd
    s2d, mean, rstd = torch.empty(slots, 21, 21, 64, dtype=torch.uint8, device=DEV), torch.empty(slots, device=DEV), torch.empty(slots, device=DEV)
    hip.obs_space_to_depth(frames.data_ptr(), True, slots, 4, 84, 84, 4, s2d.data_ptr(), mean.data_ptr(), rstd.data_ptr())
    del frames
    rows = torch.randperm(slots, device=DEV, generator=g)[:n].to(torch.int32)
    desc = hip.conv_desc(n, 21, 21, 64, 2, 2, 1, 32, 1)
    gamma, beta = 1 + _f(21, 21, 64, seed=32, amp=0.2), _f(21, 21, 64, seed=33, amp=0.2)
    w, b = _f(32, 2, 2, 64, seed=34, amp=0.06), _f(32, seed=35, amp=0.1)
    ws = torch.empty(hip.conv2d_obs_fwd_workspace(desc), device=DEV)
    yh = torch.zeros(n * 400 * 32, device=DEV)
    hm, ham, hs = torch.zeros(n * 400, dtype=torch.int32, device=DEV), _slot(0.0), _slot(0.0)
    hip.dispatch_tiles(reset=True)
    hip.conv2d_obs_fwd_h2(desc, s2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(), b.data_ptr(),
                          yh.data_ptr(), hs.data_ptr(), ws.data_ptr(), rows, ham.data_ptr(), hm.data_ptr(), reuse_folded=False, ent_order=2)
    assert hip.dispatch_tiles(reset=True) == {"obs_fwd_bf16:k256:h2blk:split250": 1}   # obs_h2.h: 50 blocks of 2 x 4 positions x 5 ranges of tiles
    y = torch.empty(n * 400, 32, device=DEV)
    hip.h2_unpack_rows(yh.data_ptr(), n * 400, 32, hs.data_ptr(), y.data_ptr(), 32)
    yy, xx = np.meshgrid(np.arange(20), np.arange(20), indexing="ij")
    ent = torch.from_numpy((((yy & 1) * 2 + (xx & 1)) * 100 + (yy >> 1) * 10 + (xx >> 1)).reshape(-1)).to(DEV)
    y = y.view(n, 400, 32)[:, ent].reshape(n, 20, 20, 32)   # raster order

    def pre_activation(idx, gam, bet, wt, bias):   # float64, from the bytes
        xs = s2d[rows[idx].long()].double()
        mu = xs.mean((1, 2, 3), keepdim=True)
        xhat = (xs - mu) / torch.sqrt(xs.var((1, 2, 3), unbiased=False, keepdim=True) + 1e-5)
        return F.conv2d((xhat * gam + bet).permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), bias).permute(0, 2, 3, 1)

    pick = _pick(n)
    ref = F.relu(pre_activation(pick, gamma.double(), beta.double(), w.double(), b.double()))
    _close(y[pick], ref)
    top = float(y.abs().max())
    assert abs(float(ham.item()) - top) <= 1e-5 * top and top * float(hs.item()) < 2**15 and float(ref.abs().max()) <= top * (1 + 1e-5)
    # backward: dz gated by the forward's own ReLU
    dz = (_f(n, 20, 20, 32, seed=36, amp=1e-3) * (y > 0)).contiguous()
    outs = [torch.zeros(32 * 256, device=DEV), torch.zeros(32, device=DEV), torch.zeros(21 * 21 * 64, device=DEV), torch.zeros(21 * 21 * 64, device=DEV)]
    wsb = torch.empty(hip.conv2d_obs_bwd_workspace(desc), device=DEV)
    hip.conv2d_obs_bwd(desc, s2d.data_ptr(), True, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(),
                       dz.data_ptr(), *[o.data_ptr() for o in outs], wsb.data_ptr(), channels_last=True, row_index=rows)
    assert hip.dispatch_tiles(reset=True) == {"obs_bwd_bf16:k256:f32:split8": 1}
    # ... and with the measured bound of |dz| that the trainer's data gradient hands over: the block kernel of obs_h2.h
    outs_blk = [torch.zeros_like(o) for o in outs]
    bound = dz.abs().max().reshape(1).clone()
    hip.conv2d_obs_bwd(desc, s2d.data_ptr(), True, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(),
                       dz.data_ptr(), *[o.data_ptr() for o in outs_blk], wsb.data_ptr(), channels_last=True, row_index=rows,
                       dz_absmax_ptr=bound.data_ptr())
    assert hip.dispatch_tiles(reset=True) == {"obs_bwd_bf16:k256:h2blk:split5": 1}   # 50 blocks of 2 x 4 positions x 5 sample ranges
    leaves = [t.double().requires_grad_(True) for t in (w, b, gamma, beta)]
    for i0 in range(0, n, 1024):
        idx = torch.arange(i0, min(n, i0 + 1024), device=DEV)
        (pre_activation(idx, leaves[2], leaves[3], leaves[0], leaves[1]) * dz[idx].double()).sum().backward()
    for kernel, res in (("bf16", outs), ("block", outs_blk)):
        for got, leaf, name in zip(res, leaves, ("dw", "db", "dgamma", "dbeta")):
            err = float((got.double() - leaf.grad.reshape(-1)).abs().max()) / float(leaf.grad.abs().max())
            assert err <= 1e-5, (kernel, name, err)


@pytest.mark.parametrize("n1,n2,indexed,slack", [(2500, 2077, True, 1.0), (4096, 2048, False, 37.0), (37, 40, True, 1.0), (33, 300, False, 3.0)])
def test_first_layer_block_weight_gradient_ragged_and_accumulated(n1, n2, indexed, slack):
    """obs_h2.h's weight gradient on sample counts that are no multiple of its 16-sample tiles (down to fewer tiles than
    workgroups per block), frames in place (row_index) or
    consecutive, a bound of |dz| that is exact or loose, and the position sums accumulated over two calls (`phase`): against
    obs_bf16.h's kernel (exact bf16 pieces, checked against float64 above) on the same samples in one call."""
    hip = _hip()
    n, slots = n1 + n2, n1 + n2 + 40
    g = torch.Generator(device=DEV).manual_seed(41)
    frames = torch.randint(0, 256, (slots, 4, 84, 84), dtype=torch.uint8, device=DEV, generator=g)
    frames[::7] //= 32
    s2d, mean, rstd = torch.empty(slots, 21, 21, 64, dtype=torch.uint8, device=DEV), torch.empty(slots, device=DEV), torch.empty(slots, device=DEV)
    hip.obs_space_to_depth(frames.data_ptr(), True, slots, 4, 84, 84, 4, s2d.data_ptr(), mean.data_ptr(), rstd.data_ptr())
    rows = torch.randperm(slots, device=DEV, generator=g)[:n].to(torch.int32) if indexed else None
    gamma, beta = 1 + _f(21, 21, 64, seed=42, amp=0.2), _f(21, 21, 64, seed=43, amp=0.2)
    w = _f(32, 2, 2, 64, seed=44, amp=0.06)
    dz = (_f(n, 20, 20, 32, seed=45, amp=1e-3) * (_f(n, 20, 20, 32, seed=46) > 0)).contiguous()
    dz[n1 - 3:n1 + 5] *= 50.0   # the largest rows sit at the seam of the two calls
    bound = (dz.abs().max() * slack).reshape(1).clone()

    def call(i0, cnt, phase, with_bound, outs, ws):
        desc = hip.conv_desc(cnt, 21, 21, 64, 2, 2, 1, 32, 1)
        hip.conv2d_obs_bwd(desc, s2d.data_ptr() + (0 if indexed else i0 * 21 * 21 * 64), True, mean.data_ptr() + (0 if indexed else 4 * i0),
                           rstd.data_ptr() + (0 if indexed else 4 * i0), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(),
                           dz.data_ptr() + 4 * i0 * 400 * 32, *[o.data_ptr() for o in outs], ws.data_ptr(), channels_last=True,
                           row_index=rows[i0:i0 + cnt].contiguous() if indexed else None, phase=phase,
                           dz_absmax_ptr=bound.data_ptr() if with_bound else None)

    shapes = (32 * 256, 32, 21 * 21 * 64, 21 * 21 * 64)
    ref, got = [torch.zeros(k, device=DEV) for k in shapes], [torch.zeros(k, device=DEV) for k in shapes]
    ws = torch.empty(hip.conv2d_obs_bwd_workspace(hip.conv_desc(n, 21, 21, 64, 2, 2, 1, 32, 1)), device=DEV)
    hip.dispatch_tiles(reset=True)
    call(0, n, 3, False, ref, ws)
    assert all(k.startswith("obs_bwd_bf16:k256:f32:split") for k in hip.dispatch_tiles(reset=True))
    call(0, n1, 1, True, got, ws)
    assert all(float(o.abs().max()) == 0.0 for o in got)   # an open accumulation forms no gradients yet
    call(n1, n2, 2, True, got, ws)
    tiles = hip.dispatch_tiles(reset=True)   # split = min(5, tiles of 16 samples): every workgroup a range of its own
    assert sum(tiles.values()) == 2 and all(k.startswith("obs_bwd_bf16:k256:h2blk:split") for k in tiles), tiles
    for a, b, name in zip(got, ref, ("dw", "db", "dgamma", "dbeta")):
        err = float((a - b).abs().max()) / float(b.abs().max())
        assert err <= 3e-6, (name, err)   # float32 accumulation in a different order; the pieces drop nothing at this slack


def test_linear_forward_wide_split_with_finished_rows_written():
    """The training-chunk variant of the split Linear forward (opt-in, `SRL_FC_TRAIN_SPLITK`): 256 channels per workgroup over two
    k-ranges, and `srl_ln_heads_fwd` writing the finished rows (`x_out`) a backward pass will read: they equal the unsplit product."""
    hip = _hip()
    M, K, N, ks = 9000, 3136, 512, 2
    x = _f(M, K, seed=33, relu=True, amp=2.0)
    w = _f(N, K, seed=34, amp=0.03)
    b = _f(N, seed=35, amp=0.1)
    xbuf, sx, ax = torch.empty_like(x), _slot(0.0), _absmax(hip, x)
    hip.h2_pack_rows(x.data_ptr(), K, M, K, xbuf.data_ptr(), absmax=ax.data_ptr(), scale_out=sx.data_ptr())
    wp, sw, rw = _weights(hip, w, N, K, 0)
    y = torch.zeros(M, N, device=DEV)
    hip.h2_gemm(xbuf.data_ptr(), wp.data_ptr(), sx.data_ptr(), sw.data_ptr(), M, N, K, y.data_ptr(), bias=b.data_ptr(), act=1)
    slabs = torch.full((ks, M, N), float("nan"), device=DEV)
    hip.dispatch_tiles(reset=True)
    hip.h2_gemm_splitk(xbuf.data_ptr(), wp.data_ptr(), sx.data_ptr(), sw.data_ptr(), M, N, K, slabs.data_ptr(), ks, wide=True)
    assert hip.dispatch_tiles(reset=True) == {"h2:gemm:8:s2": 1}
    heads = (6, 1)
    g, be = 1 + 0.1 * _f(N, seed=36), 0.1 * _f(N, seed=37)
    W, hb = [_f(a, N, seed=38 + i, amp=0.05) for i, a in enumerate(heads)], [_f(a, seed=40 + i, amp=0.1) for i, a in enumerate(heads)]
    o = [torch.empty(M, a, device=DEV) for a in heads]
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    xo = torch.full((M, N), float("nan"), device=DEV)
    hip.ln_heads_fwd(slabs.data_ptr(), N, M, N, g.data_ptr(), be.data_ptr(), [t.data_ptr() for t in W], [t.data_ptr() for t in hb],
                     list(heads), [t.data_ptr() for t in o], list(heads), mean.data_ptr(), rstd.data_ptr(), x_slabs=ks,
                     x_slab_stride=M * N, x_bias=b.data_ptr(), x_act=1, x_out=xo.data_ptr(), ldxo=N)
    assert float((xo - y).abs().max()) <= 1e-5 * float(y.abs().max())
    _close(xo, F.relu(x.double() @ w.double().t() + b.double()))
