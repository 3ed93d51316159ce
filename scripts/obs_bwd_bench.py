"""First layer's weight gradient on one 16 384-frame chunk: obs_bf16.h's per-position kernel (no bound of |dz| given) against
obs_h2.h's block kernel (bound given) -- results against each other, then timing (HIP events; phase 1 = position sums only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srl_amd import hip
DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
slots = n + 64
g = torch.Generator(device=DEV).manual_seed(1)
frames = torch.randint(0, 256, (slots, 4, 84, 84), dtype=torch.uint8, device=DEV, generator=g)
frames[::5] //= 16
s2d, mean, rstd = torch.empty(slots, 21, 21, 64, dtype=torch.uint8, device=DEV), torch.empty(slots, device=DEV), torch.empty(slots, device=DEV)
hip.obs_space_to_depth(frames.data_ptr(), True, slots, 4, 84, 84, 4, s2d.data_ptr(), mean.data_ptr(), rstd.data_ptr())
del frames
rows = torch.randperm(slots, device=DEV, generator=g)[:n].to(torch.int32)
desc = hip.conv_desc(n, 21, 21, 64, 2, 2, 1, 32, 1)
f = lambda *shape: torch.randn(*shape, device=DEV, generator=g)
gamma, beta = 1 + 0.2 * f(21, 21, 64), 0.2 * f(21, 21, 64)
w = 0.06 * f(32, 2, 2, 64)
dz = (1e-3 * f(n, 400, 32) * (f(n, 400, 32) > 0)).contiguous()
amax = dz.abs().max().reshape(1).clone()
wsb = torch.empty(hip.conv2d_obs_bwd_workspace(desc), device=DEV)


def run(bound, ri, phase=3):
    outs = [torch.zeros(32 * 256, device=DEV), torch.zeros(32, device=DEV), torch.zeros(21 * 21 * 64, device=DEV), torch.zeros(21 * 21 * 64, device=DEV)]
    hip.conv2d_obs_bwd(desc, s2d.data_ptr(), True, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(),
                       dz.data_ptr(), *[o.data_ptr() for o in outs], wsb.data_ptr(), channels_last=True, row_index=ri, phase=phase,
                       dz_absmax_ptr=amax.data_ptr() if bound else None)
    return outs


for order in ("index", "span"):
    ri = rows if order == "index" else None
    hip.dispatch_tiles(reset=True)
    ref = run(False, ri)
    got = run(True, ri)
    print(hip.dispatch_tiles(reset=True))
    for a, b, name in zip(got, ref, ("dw", "db", "dgamma", "dbeta")):
        print(f"  {order} {name}: max |block - bf16| / max |bf16| = {float((a - b).abs().max() / b.abs().max()):.3e}", flush=True)
    for bound in (False, True):
        for _ in range(3):
            run(bound, ri, 1)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(20):
            run(bound, ri, 1)
        e.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(e) / 20 * 1e3
        print(f"obs bwd ({order}, {'block' if bound else 'bf16'}): {us:8.1f} us per call (with the slab fold and 4 memsets)", flush=True)
