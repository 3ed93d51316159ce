"""The three products of a football-sized dense layer (22528 -> 11264) on the pre-split kernels against float64 on sampled rows /
outputs: srl_h2_pack_rows + srl_h2_gemm (forward, data gradient) + srl_h2_wgrad_dense at the row count given (default 51200)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srl_amd import hip

DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
K, N = 22528, 11264
g = torch.Generator(device=DEV).manual_seed(1)
rnd = lambda *s, amp=1.0: (torch.rand(*s, device=DEV, generator=g) * 2 - 1) * amp
x = rnd(M, K, amp=2.0)
x = torch.where(x < 0, torch.zeros_like(x), x * x)       # relu-like, heavy tail
w = rnd(N, K, amp=0.01)
b = rnd(N, amp=0.1)
slot = lambda: torch.zeros(1, device=DEV)


def pieces(n, width):
    step = max(256, ((0xfff00000 // (4 * width)) // 256) * 256)
    return [(r0, min(n, r0 + step)) for r0 in range(0, n, step)]


# forward
ax, sx = slot(), slot()
hip.absmax(x.data_ptr(), M * K, ax.data_ptr())
xh = torch.empty(M * K, device=DEV)
hip.h2_pack_rows(x.data_ptr(), K, M, K, xh.data_ptr(), absmax=ax.data_ptr(), scale_out=sx.data_ptr())
aw, sw, rw, swt, rwt = slot(), slot(), slot(), slot(), slot()
hip.absmax(w.data_ptr(), N * K, aw.data_ptr())
wh, wth = torch.empty(N * K, device=DEV), torch.empty(N * K, device=DEV)
hip.h2_weights(w.data_ptr(), N, K, 0, aw.data_ptr(), sw.data_ptr(), rw.data_ptr(), wh.data_ptr())
hip.h2_weights(w.data_ptr(), K, N, 1, aw.data_ptr(), swt.data_ptr(), rwt.data_ptr(), wth.data_ptr())
y = torch.full((M, N), float("nan"), device=DEV)
mask = torch.zeros(M * N // 32, dtype=torch.int32, device=DEV)
oam = slot()
for r0, r1 in pieces(M, max(K, N)):
    hip.h2_gemm(xh.data_ptr() + 4 * r0 * K, wh.data_ptr(), sx.data_ptr(), sw.data_ptr(), r1 - r0, N, K, y.data_ptr() + 4 * r0 * N,
                bias=b.data_ptr(), act=1, mask_out=mask.data_ptr() + 4 * (r0 * N // 32), out_absmax=oam.data_ptr())
rows = torch.cat([torch.tensor([0, 1, M - 1, M - 2], device=DEV), torch.randint(0, M, (60,), device=DEV, generator=g)])
if M > 47616:
    rows = torch.cat([rows, torch.tensor([47615, 47616, 47617], device=DEV)])
ref = torch.relu(x[rows].double() @ w.double().t() + b.double())
err = float((y[rows].double() - ref).abs().max() / ref.abs().max())
print(f"forward  M={M}: max err / max |ref| = {err:.3e}   (max |y| {float(oam):.5g} vs {float(y.abs().max()):.5g})", flush=True)
# data gradient: dx = (dz w) * relu'(x)
dz = rnd(M, N, amp=1e-3)
dz = dz * (torch.rand(M, N, device=DEV, generator=g) < 0.5)
az, sz = slot(), slot()
hip.absmax(dz.data_ptr(), M * N, az.data_ptr())
dzh = torch.empty(M * N, device=DEV)
hip.h2_pack_rows(dz.data_ptr(), N, M, N, dzh.data_ptr(), absmax=az.data_ptr(), scale_out=sz.data_ptr())
xmask = torch.zeros(M * K // 32, dtype=torch.int32, device=DEV)
hip.relu_mask(x.data_ptr(), M * K, xmask.data_ptr())
dx = torch.full((M, K), float("nan"), device=DEV)
for r0, r1 in pieces(M, max(K, N)):
    hip.h2_gemm(dzh.data_ptr() + 4 * r0 * N, wth.data_ptr(), sz.data_ptr(), swt.data_ptr(), r1 - r0, K, N, dx.data_ptr() + 4 * r0 * K,
                mask_in=xmask.data_ptr() + 4 * (r0 * K // 32), mask_in_h2order=False)
refd = (dz[rows].double() @ w.double()) * (x[rows] > 0)
errd = float((dx[rows].double() - refd).abs().max() / refd.abs().max())
print(f"dgrad    M={M}: max err / max |ref| = {errd:.3e}", flush=True)
del dx, y
torch.cuda.empty_cache()
# weight gradient: gw = dz^T x
gw = torch.zeros(N, K, device=DEV)
ws = torch.empty(max(hip.h2_wgrad_dense_workspace(M, N, K), 4), device=DEV)
hip.h2_wgrad_dense(dzh.data_ptr(), xh.data_ptr(), sz.data_ptr(), sx.data_ptr(), M, N, K, ws.data_ptr(), gw.data_ptr(), accumulate=True)
os_ = torch.randint(0, N, (48,), device=DEV, generator=g)
refw = dz[:, os_].double().t() @ x.double()
errw = float((gw[os_].double() - refw).abs().max() / refw.abs().max())
print(f"wgrad    M={M}: max err / max |ref| = {errw:.3e}", flush=True)
