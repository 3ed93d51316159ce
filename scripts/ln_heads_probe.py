"""srl_ln_heads_fwd / srl_ln_heads_bwd alone at the Atari chunk's size (16 384 x 512, heads 6 + 1): launch times by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srl_amd import hip
n, D, heads = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 512, (6, 1)
d = "cuda:0"
g = torch.Generator(device=d).manual_seed(0)
f = lambda *s: torch.randn(*s, device=d, generator=g)
x = torch.relu(f(n, D)); gam, bet = 1 + 0.1 * f(D), 0.1 * f(D)
W, b = [f(a, D) / D ** 0.5 for a in heads], [0.1 * f(a) for a in heads]
y = [torch.empty(n, a, device=d) for a in heads]
mean, rstd = torch.empty(n, device=d), torch.empty(n, device=d)
dy = [f(n, a) for a in heads]
dx = torch.empty(n, D, device=d)
dg, db = torch.zeros(D, device=d), torch.zeros(D, device=d)
dW, dhb = [torch.zeros_like(w) for w in W], [torch.zeros_like(v) for v in b]
P = lambda ts: [t.data_ptr() for t in ts]
fw = lambda: hip.ln_heads_fwd(x.data_ptr(), D, n, D, gam.data_ptr(), bet.data_ptr(), P(W), P(b), list(heads), P(y), list(heads), mean.data_ptr(), rstd.data_ptr())
bw = lambda: hip.ln_heads_bwd(x.data_ptr(), D, n, D, gam.data_ptr(), bet.data_ptr(), mean.data_ptr(), rstd.data_ptr(), P(W), list(heads), P(dy), list(heads), 1,
                              dx.data_ptr(), D, dg.data_ptr(), db.data_ptr(), P(dW), P(dhb))
def timeit(fn, k=50):
    for _ in range(5): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / k * 1e3
print(f"n {n}: fwd {timeit(fw):7.1f} us  bwd {timeit(bw):7.1f} us")
