"""Kernel-level check of the FC weight-gradient product (dz^T x, both operands k-major, a_colsum fused) and the FC data
gradient against float64, for both kernel sets.  usage: python3 scripts/wgrad_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srl_amd import hip

g = torch.Generator(device="cuda:0").manual_seed(0)
for mode in ("bf16x3", "f32"):
    os.environ["SRL_MFMA"] = "f32" if mode == "f32" else ""
    for K in (40, 256, 1024):
        M, N = 512, 3136
        A = torch.randn((K, M), device="cuda:0", generator=g)
        A = A * (torch.rand((K, M), device="cuda:0", generator=g) < 0.5)  # half the entries zero like a ReLU-masked dz
        B = torch.relu(torch.randn((K, N), device="cuda:0", generator=g))
        C = torch.zeros((M, N), device="cuda:0")
        cs = torch.zeros(M, device="cuda:0")
        hip.dispatch_counts(reset=True)
        hip.gemm(M, N, K, A.data_ptr(), M, 1, B.data_ptr(), N, 1, C.data_ptr(), N, accumulate=True, a_colsum=cs.data_ptr())
        ref = A.double().t() @ B.double()
        e = (C.double() - ref)
        rms = ref.pow(2).mean().sqrt()
        ecs = cs.double() - A.double().sum(0)
        print(f"{mode:7s} wgrad K={K:5d}: err rms/rms {float(e.pow(2).mean().sqrt() / rms):.2e} max/rms {float(e.abs().max() / rms):.2e}; "
              f"colsum err max {float(ecs.abs().max()):.2e} (scale {float(A.double().sum(0).abs().mean()):.2e})  {hip.dispatch_counts()}")
        # data gradient: dX = dz W, dz [K rows, M], W [M, N] k-major B, ReLU mask from x
        dz, W = A, torch.randn((M, N), device="cuda:0", generator=g) * 0.02
        X = B
        dX = torch.empty((K, N), device="cuda:0")
        hip.gemm(K, N, M, dz.data_ptr(), M, 0, W.data_ptr(), N, 1, dX.data_ptr(), N, dact_src=X.data_ptr(), ld_dact=N, dact=1)
        ref = (dz.double() @ W.double()) * (X > 0)
        e = dX.double() - ref
        rms = ref.pow(2).mean().sqrt()
        print(f"{mode:7s} dgrad rows={K:5d}: err rms/rms {float(e.pow(2).mean().sqrt() / rms):.2e} max/rms {float(e.abs().max() / rms):.2e}")
