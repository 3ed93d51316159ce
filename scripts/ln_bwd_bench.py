import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srl_amd import hip
DEV = "cuda:0"
rows, D = 16384, 512
x, dy = torch.relu(torch.randn(rows, D, device=DEV)), torch.randn(rows, D, device=DEV)
g, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
y, mean, rstd = torch.empty_like(x), torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
hip.layernorm_fwd(x.data_ptr(), D, g.data_ptr(), b.data_ptr(), rows, D, y.data_ptr(), D, mean.data_ptr(), rstd.data_ptr())
dx, dg, db, am = torch.empty_like(x), torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(1, device=DEV)
run = lambda: hip.layernorm_bwd(dy.data_ptr(), D, x.data_ptr(), D, g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, D, dx.data_ptr(), D, 1,
                                dg.data_ptr(), db.data_ptr(), dx_absmax=am.data_ptr())
for _ in range(5): run()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(50): run()
e.record(); torch.cuda.synchronize()
print(f"layernorm_bwd {rows}x{D}: {a.elapsed_time(e) / 50 * 1e3:.1f} us")
