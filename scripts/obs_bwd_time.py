"""Timing only of the first layer's block weight gradient (obs_h2.h), phase 1 (position sums): for the SRL_OBSB_DBG leave-outs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srl_amd import hip
DEV = "cuda:0"
n = 16384
slots = n + 64
g = torch.Generator(device=DEV).manual_seed(1)
frames = torch.randint(0, 256, (slots, 4, 84, 84), dtype=torch.uint8, device=DEV, generator=g)
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
outs = [torch.zeros(32 * 256, device=DEV), torch.zeros(32, device=DEV), torch.zeros(21 * 21 * 64, device=DEV), torch.zeros(21 * 21 * 64, device=DEV)]
for bound in ([True, False] if os.environ.get("SRL_OBSB_DBG", "0") == "0" else [True]):
    run = lambda: hip.conv2d_obs_bwd(desc, s2d.data_ptr(), True, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(),
                                     dz.data_ptr(), *[o.data_ptr() for o in outs], wsb.data_ptr(), channels_last=True, row_index=rows, phase=1,
                                     dz_absmax_ptr=amax.data_ptr() if bound else None)
    for _ in range(3):
        run()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(20):
        run()
    e.record()
    torch.cuda.synchronize()
    print(f"obs bwd {'block' if bound else 'bf16 '} DBG={os.environ.get('SRL_OBSB_DBG', '0'):>2s}: {a.elapsed_time(e) / 20 * 1e3:8.1f} us per call", flush=True)
