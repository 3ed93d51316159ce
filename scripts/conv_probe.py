"""Per-kernel times of the convolution / FC launches of one 16384-row chunk of the benchmark's network, with operand ranges
(the two-piece f16 kernels) and sign masks as the trainer passes them.  Used with leave-out builds
(scripts/build_variant.sh <tag> -DSRL_GEMM3_DBG=<bits>; SRL_HIP_LIB=...).  usage: python3 scripts/conv_probe.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srl_amd import hip

DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


def amax(t):
    return t.abs().max().reshape(1).float().contiguous()


for (H, Cin, k, s, Cout) in ((20, 32, 4, 2, 64), (9, 64, 3, 1, 64)):
    d = hip.conv_desc(n, H, H, Cin, k, k, s, Cout, act=1)
    OH = (H - k) // s + 1
    x = torch.relu(torch.randn((n, H, H, Cin), device=DEV))
    w = torch.randn((Cout, k, k, Cin), device=DEV) * 0.05
    b = torch.randn(Cout, device=DEV)
    y = torch.empty((n, OH, OH, Cout), device=DEV)
    dz = torch.randn((n, OH, OH, Cout), device=DEV) * (torch.rand((n, OH, OH, Cout), device=DEV) < 0.5)
    gw = torch.zeros_like(w)
    gb = torch.zeros(Cout, device=DEV)
    ws = torch.empty(max(hip.conv2d_wgrad_workspace(d), 1), device=DEV)
    wt = torch.empty(hip.conv2d_dgrad_weight_elems(d), device=DEV)
    dx = torch.empty_like(x)
    hip.conv2d_dgrad_repack(d, w.data_ptr(), wt.data_ptr())
    xr, wr, dzr = amax(x), amax(w), amax(dz)
    yr, dxr = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    ym = torch.zeros(y.numel() // 32, dtype=torch.int32, device=DEV)
    xm = torch.randint(-2 ** 31, 2 ** 31 - 1, (x.numel() // 32,), dtype=torch.int32, device=DEV)
    P = lambda t: t.data_ptr()
    cases = (
             ("fwd", lambda: hip.conv2d_nhwc_fwd(d, P(x), P(w), P(b), P(y), x_absmax=P(xr), w_absmax=P(wr), y_absmax=P(yr))),
             ("fwd+mask", lambda: hip.conv2d_nhwc_fwd(d, P(x), P(w), P(b), P(y), x_absmax=P(xr), w_absmax=P(wr), y_absmax=P(yr), y_mask=P(ym))),
             ("wgrad", lambda: hip.conv2d_nhwc_wgrad(d, P(x), P(dz), P(gw), P(ws), P(gb), x_absmax=P(xr), dz_absmax=P(dzr))),
             ("dgrad floats", lambda: hip.conv2d_nhwc_dgrad(d, P(dz), P(wt), P(x), 1, P(dx), dz_absmax=P(dzr), w_absmax=P(wr), dx_absmax=P(dxr))),
             ("dgrad bits", lambda: hip.conv2d_nhwc_dgrad(d, P(dz), P(wt), None, 1, P(dx), dz_absmax=P(dzr), w_absmax=P(wr), dx_absmax=P(dxr), x_mask=P(xm))),
             ("dgrad none", lambda: hip.conv2d_nhwc_dgrad(d, P(dz), P(wt), None, 0, P(dx), dz_absmax=P(dzr), w_absmax=P(wr), dx_absmax=P(dxr))),
             ("dgrad none, no range out", lambda: hip.conv2d_nhwc_dgrad(d, P(dz), P(wt), None, 0, P(dx), dz_absmax=P(dzr), w_absmax=P(wr))))
    for name, fn in cases:
        print(f"conv H={H} Cin={Cin} k={k} s={s} Cout={Cout} {name:26s}: {timeit(fn):8.1f} us", flush=True)
# FC 3136 -> 512
M, K, N = n, 3136, 512
x = torch.relu(torch.randn((M, K), device=DEV))
w = torch.randn((N, K), device=DEV) * 0.02
b = torch.randn(N, device=DEV)
y = torch.empty((M, N), device=DEV)
dz = torch.randn((M, N), device=DEV)
dx = torch.empty_like(x)
gw, gb = torch.zeros_like(w), torch.zeros(N, device=DEV)
xr, wr, dzr = amax(x), amax(w), amax(dz)
xm = torch.randint(-2 ** 31, 2 ** 31 - 1, (x.numel() // 32,), dtype=torch.int32, device=DEV)
ws = torch.empty(8 * N * K, device=DEV)
P = lambda t: t.data_ptr()
for name, fn in (("fwd", lambda: hip.gemm(M, N, K, P(x), K, 0, P(w), K, 0, P(y), N, bias=P(b), act=1, a_absmax=P(xr), b_absmax=P(wr))),
                 ("dgrad floats", lambda: hip.gemm(M, K, N, P(dz), N, 0, P(w), K, 1, P(dx), K, dact_src=P(x), ld_dact=K, dact=1, a_absmax=P(dzr), b_absmax=P(wr))),
                 ("dgrad bits", lambda: hip.gemm(M, K, N, P(dz), N, 0, P(w), K, 1, P(dx), K, ld_dact=K, dact=1, a_absmax=P(dzr), b_absmax=P(wr), dact_mask=P(xm))),
                 ("dgrad none", lambda: hip.gemm(M, K, N, P(dz), N, 0, P(w), K, 1, P(dx), K, a_absmax=P(dzr), b_absmax=P(wr)))):
    print(f"FC {K}->{N} {name:26s}: {timeit(fn):8.1f} us", flush=True)
