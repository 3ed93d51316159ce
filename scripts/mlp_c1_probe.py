"""srl_mlp_fwd / srl_mlp_bwd on the C1 tower (LN4, 4->64 relu, LN64, 64->64 relu, 64->64 relu, 64->2) alone:
scripts/mlp_c1_probe.py [rows = 524288]; per-launch times by HIP events, and the output / gradient sums as a fingerprint."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from srl_amd import hip
DEV = "cuda:0"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
g = torch.Generator(device=DEV).manual_seed(0)
f = lambda *s: torch.randn(*s, device=DEV, generator=g)
chain = [(0, 4, 4, 0), (1, 4, 64, 1), (0, 64, 64, 0), (1, 64, 64, 1), (1, 64, 64, 1), (1, 64, 2, 0)]
keep, desc = [], []
for kind, i, o, act in chain:
    if kind == 0:
        w, b = 1 + 0.1 * f(i), 0.1 * f(i)
    else:
        w, b = f(o, i) / i ** 0.5, 0.1 * f(o)
    gw, gb = torch.zeros_like(w), torch.zeros_like(b)
    keep += [w, b, gw, gb]
    desc.append((kind, i, o, act, w.data_ptr(), b.data_ptr(), gw.data_ptr(), gb.data_ptr()))
arr = hip.mlp_layers(desc)
tld = hip.mlp_tape_floats_at(arr, rows)
x, dy = f(rows, 4), f(rows, 2)
tape = torch.empty(rows, max(tld, 1), device=DEV) if tld else None
tp = tape.data_ptr() if tape is not None else 0
y = torch.empty(rows, 2, device=DEV)
fw = lambda: hip.mlp_fwd(arr, x.data_ptr(), 4, rows, tp, tld, y.data_ptr(), 2)
bw = lambda: hip.mlp_bwd(arr, x.data_ptr(), 4, rows, tp, tld, dy.data_ptr(), 2)
def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
fw()
for t in keep[2::4] + keep[3::4]:
    t.zero_()
bw()
torch.cuda.synchronize()
print(f"rows {rows} tape floats {tld}: y sum {float(y.double().sum()):.6f}  grads " + " ".join(f"{float(t.double().sum()):.5f}" for t in keep[2::4]))
print(f"fwd {timeit(fw):8.1f} us   bwd {timeit(bw):8.1f} us   SRL_MLP_DBG={os.environ.get('SRL_MLP_DBG', '0')}", flush=True)
