"""Timing of the first layer's forward kernels on one 16 384-frame chunk (HIP events, the launch stream):
SRL_OBS_H2BLOCK=0/1 selects obs_bf16.h's per-position kernel or obs_h2.h's block kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srl_amd import hip
DEV = "cuda:0"
n, slots = 16384, 16384 + 64
g = torch.Generator(device=DEV).manual_seed(1)
frames = torch.randint(0, 256, (slots, 4, 84, 84), dtype=torch.uint8, device=DEV, generator=g)
s2d, mean, rstd = torch.empty(slots, 21, 21, 64, dtype=torch.uint8, device=DEV), torch.empty(slots, device=DEV), torch.empty(slots, device=DEV)
hip.obs_space_to_depth(frames.data_ptr(), True, slots, 4, 84, 84, 4, s2d.data_ptr(), mean.data_ptr(), rstd.data_ptr())
del frames
rows = torch.randperm(slots, device=DEV, generator=g)[:n].to(torch.int32)
desc = hip.conv_desc(n, 21, 21, 64, 2, 2, 1, 32, 1)
f = lambda *shape: torch.randn(*shape, device=DEV, generator=g)
gamma, beta = 1 + 0.2 * f(21, 21, 64), 0.2 * f(21, 21, 64)
w, b = 0.06 * f(32, 2, 2, 64), 0.1 * f(32)
ws = torch.empty(hip.conv2d_obs_fwd_workspace(desc), device=DEV)
yh = torch.zeros(n * 400 * 32, device=DEV)
hm, ham, hs = torch.zeros(n * 400, dtype=torch.int32, device=DEV), torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
for order in ("index", "span"):
    ri = rows if order == "index" else None
    run = lambda fold: hip.conv2d_obs_fwd_h2(desc, s2d.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(),
                                             b.data_ptr(), yh.data_ptr(), hs.data_ptr(), ws.data_ptr(), ri, ham.data_ptr(), hm.data_ptr(),
                                             reuse_folded=not fold, ent_order=2)
    run(True)
    for _ in range(3):
        run(False)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(20):
        run(False)
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / 20 * 1e3
    print(f"obs fwd h2 ({order}, block={os.environ.get('SRL_OBS_H2BLOCK', '1')}): {us:8.1f} us  {2 * n * 400 * 32 * 256 / us / 1e6:7.1f} TFLOP/s algorithmic", flush=True)
