"""First-layer backward on byte frames (obs_bwd_bf16_kernel) in isolation: time per 16 384-sample launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srl_amd import hip
DEV = "cuda:0"; n = 16384
d = hip.conv_desc(n, 21, 21, 64, 2, 2, 1, 32, act=1)
s2d = torch.randint(0, 256, (n, 21, 21, 64), dtype=torch.uint8, device=DEV)
mean = torch.full((n,), 127.0, device=DEV); rstd = torch.full((n,), 0.02, device=DEV)
g = torch.ones(21 * 21 * 64, device=DEV); b = torch.zeros(21 * 21 * 64, device=DEV)
w = torch.randn(32 * 256, device=DEV) * 0.05
dz = torch.randn((n, 20, 20, 32), device=DEV)
outs = [torch.zeros(32 * 256, device=DEV), torch.zeros(32, device=DEV), torch.zeros(21 * 21 * 64, device=DEV), torch.zeros(21 * 21 * 64, device=DEV)]
ws = torch.empty(hip.conv2d_obs_bwd_workspace(d), device=DEV)
f = lambda: hip.conv2d_obs_bwd(d, s2d.data_ptr(), True, mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), b.data_ptr(), w.data_ptr(), dz.data_ptr(), *[o.data_ptr() for o in outs], ws.data_ptr(), channels_last=True)
for _ in range(3): f()
a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(20): f()
bb.record(); torch.cuda.synchronize()
print("obs bwd (kernel + finalisation): %.1f us per launch" % (a.elapsed_time(bb) * 1e3 / 20))
