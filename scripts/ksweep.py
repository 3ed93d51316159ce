"""Time of a tall plain GEMM (M rows, N in {32, 64}) against K: the intercept is the per-tile fixed cost (prologue +
epilogue), the slope the steady-state k-loop."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srl_amd import hip

hip.require_gpu()
dev = "cuda:0"
for N in (32, 64):
    M = 6553600 if N == 32 else 1327104
    for K in (16, 32, 64, 128, 256, 512, 1024):
        A = torch.randn(M, K, device=dev)
        B = torch.randn(N, K, device=dev)
        C = torch.empty(M, N, device=dev)
        b = torch.zeros(N, device=dev)
        f = lambda: hip.gemm(M, N, K, A.data_ptr(), K, 0, B.data_ptr(), K, 0, C.data_ptr(), N, bias=b.data_ptr(), act=1)
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"N={N} M={M} K={K:5d}: {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TF", flush=True)
        del A, C
