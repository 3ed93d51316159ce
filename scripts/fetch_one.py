"""FETCH_SIZE experiment helper: one dense product shape, a few launches (run under rocprofv3 --pmc FETCH_SIZE)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srl_amd import hip

M, N, K, akm, bkm, split = (int(a) for a in sys.argv[1:7])
A = torch.randn((K, M) if akm else (M, K), device="cuda:0")
B = torch.randn((K, N) if bkm else (N, K), device="cuda:0")
C = torch.empty((M, N), device="cuda:0")
ws = torch.empty(split * M * N, device="cuda:0") if split > 1 else None
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(12):
    if i == 2:
        ev0.record()
    hip.gemm(M, N, K, A.data_ptr(), A.shape[1], akm, B.data_ptr(), B.shape[1], bkm, C.data_ptr(), N, split_k=split,
             workspace=None if ws is None else ws.data_ptr())
ev1.record()
torch.cuda.synchronize()
print(f"{ev0.elapsed_time(ev1) / 10:.4f} ms")
