"""Random dense shapes / orientations / split-K counts through srl_gemm against a float64 product (one-off sweep)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from srl_amd import hip
rng = np.random.default_rng(0)
DEV = "cuda:0"
bad = 0
for it in range(60):
    M = int(rng.choice([64, 192, 512, 1000, 4096, 16384, 20000]))
    N = int(rng.choice([64, 128, 320, 576, 1100, 3136]))
    K = int(rng.choice([64, 256, 512, 1000, 3136, 8192, 20000]))
    akm, bkm = [(0, 0), (0, 1), (1, 1), (1, 0)][it % 4]
    split = int(rng.choice([1, 1, 2, 5, 9])) if K >= 1000 else 1
    A = torch.randn((K, M) if akm else (M, K), device=DEV)
    B = torch.randn((K, N) if bkm else (N, K), device=DEV)
    C = torch.full((M, N), float("nan"), device=DEV)
    ws = torch.empty(split * M * N, device=DEV) if split > 1 else None
    hip.gemm(M, N, K, A.data_ptr(), A.shape[1], akm, B.data_ptr(), B.shape[1], bkm, C.data_ptr(), N, split_k=split,
             workspace=None if ws is None else ws.data_ptr())
    ref = (A.double().T if akm else A.double()) @ (B.double() if bkm else B.double().T)
    err = (C.double() - ref).abs().max().item()
    t32 = ((A.T if akm else A) @ (B if bkm else B.T)).double()  # the library float32 product: the error scale of this shape
    terr = (t32 - ref).abs().max().item()
    ok = np.isfinite(err) and err <= 3.0 * max(terr, 1e-6 * np.sqrt(K))
    bad += not ok
    print(f"M={M} N={N} K={K} akm={akm} bkm={bkm} split={split} err={err:.3e} (library float32 GEMM: {terr:.3e}){'' if ok else '  <-- BAD'}")
print("bad", bad)
