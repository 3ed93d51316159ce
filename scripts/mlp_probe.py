"""Times of srl_mlp_fwd / srl_mlp_bwd over chain shapes and row counts (csrc/mlp_small.hip)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srl_amd import hip
DEV = "cuda:0"


def timeit(fn, reps=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


def chain(kinds, dims):
    """kinds: string of 'n' (LayerNorm) / 'l' (Linear+relu) / 'h' (Linear, no act); dims: widths along the Linears."""
    desc, keep, d, di = [], [], dims[0], 0
    for ch in kinds:
        if ch == "n":
            g, b = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
            gg, gb = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
            desc.append((0, d, d, 0, g.data_ptr(), b.data_ptr(), gg.data_ptr(), gb.data_ptr()))
        else:
            o = dims[di + 1]; di += 1
            w, b = torch.randn((o, d), device=DEV) * 0.1, torch.zeros(o, device=DEV)
            gg, gb = torch.zeros_like(w), torch.zeros_like(b)
            desc.append((1, d, o, 1 if ch == "l" else 0, w.data_ptr(), b.data_ptr(), gg.data_ptr(), gb.data_ptr()))
            d = o
        keep += [g if ch == "n" else w, b, gg, gb]
    return hip.mlp_layers(desc), keep, d


for rows in (256, 65536):
    for kinds, dims in (("nlnlh", (4, 64, 64, 2)), ("l", (64, 64)), ("ll", (64, 64, 64)), ("n", (64,)), ("nn", (64,)), ("h", (64, 2))):
        arr, keep, dout = chain(kinds, dims)
        tld = hip.mlp_tape_floats(arr)
        x = torch.randn((rows, dims[0]), device=DEV)
        tape = torch.empty((rows, tld), device=DEV)
        y = torch.empty((rows, dout), device=DEV)
        dy = torch.randn((rows, dout), device=DEV)
        f = timeit(lambda: hip.mlp_fwd(arr, x.data_ptr(), dims[0], rows, tape.data_ptr(), tld, y.data_ptr(), dout))
        b = timeit(lambda: hip.mlp_bwd(arr, x.data_ptr(), dims[0], rows, tape.data_ptr(), tld, dy.data_ptr(), dout))
        print(f"rows {rows:6d} chain {kinds:6s} {dims}: fwd {f:7.1f} us  bwd {b:7.1f} us", flush=True)
