"""Fused MLP chain (srl_mlp_fwd + srl_mlp_bwd) on the 2 x 64 nets of BASELINE configs[0] at 65 536 rows:
SRL_MLP_MFMA=0 -> the FMA chain (mlp_small.hip), default -> the float32 matrix-core chain (mlp_mfma.h)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srl_amd import hip
DEV = "cuda:0"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dims, acts = (4, 64, 64, 2), (1, 1, 0)
g = torch.Generator(device=DEV).manual_seed(0)
f = lambda *s: torch.randn(*s, device=DEV, generator=g)
params = []
desc = []
for i in range(len(dims) - 1):
    w, b = f(dims[i + 1], dims[i]) / dims[i] ** 0.5, 0.1 * f(dims[i + 1])
    gw, gb = torch.zeros_like(w), torch.zeros_like(b)
    params += [w, b, gw, gb]
    desc.append((1, dims[i], dims[i + 1], acts[i], w.data_ptr(), b.data_ptr(), gw.data_ptr(), gb.data_ptr()))
arr = hip.mlp_layers(desc)
tld = hip.mlp_tape_floats(arr)
x, dy = f(rows, dims[0]), f(rows, dims[-1])
tape, y = torch.empty(rows, tld, device=DEV), torch.empty(rows, dims[-1], device=DEV)
def step():
    hip.mlp_fwd(arr, x.data_ptr(), dims[0], rows, tape.data_ptr(), tld, y.data_ptr(), dims[-1])
    hip.mlp_bwd(arr, x.data_ptr(), dims[0], rows, tape.data_ptr(), tld, dy.data_ptr(), dims[-1])
for _ in range(5):
    step()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
for _ in range(50):
    step()
e.record()
torch.cuda.synchronize()
us = a.elapsed_time(e) / 50 * 1e3
flops = 6 * rows * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)) - 2 * rows * dims[0] * dims[1]
print(f"rows {rows}: fwd + bwd {us:8.1f} us, {flops / us / 1e6:6.2f} TFLOP/s (float32), SRL_MLP_MFMA={os.environ.get('SRL_MLP_MFMA', 'default')}", flush=True)
