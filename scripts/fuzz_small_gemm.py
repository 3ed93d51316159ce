"""Small products (the 64 x 64-tile / 64-deep-step path of srl_gemm) against float64: shapes of the CartPole-sized layers,
every orientation, bias / activation, fused column sums, accumulation, split-K."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from srl_amd import hip
rng = np.random.default_rng(0)
DEV = "cuda:0"
bad = 0
for it in range(160):
    M = int(rng.choice([8, 40, 64, 100, 256, 1000]))
    N = int(rng.choice([4, 8, 64, 128]))
    K = int(rng.choice([8, 12, 32, 64, 100, 256, 512]))
    akm, bkm = [(0, 0), (0, 1), (1, 1), (1, 0)][it % 4]
    if akm and M % 4: M = (M // 4) * 4 or 4
    if bkm and N % 4: N = (N // 4) * 4 or 4
    split = int(rng.choice([1, 1, 2, 4])) if K >= 100 else 1
    A = torch.randn((K, M) if akm else (M, K), device=DEV)
    B = torch.randn((K, N) if bkm else (N, K), device=DEV)
    C0 = torch.randn((M, N), device=DEV)
    C = C0.clone()
    acc = bool(rng.integers(0, 2))
    use_bias = (not acc) and split == 1 and bool(rng.integers(0, 2))
    bias = torch.randn(N, device=DEV)
    cs0 = torch.randn(M, device=DEV)
    cs = cs0.clone()
    colsum = bool(akm and bkm and rng.integers(0, 2) and hip.gemm_colsum_ok(M, N, K, A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], bkm))
    ws = torch.empty(split * M * N, device=DEV) if split > 1 else None
    hip.dispatch_counts(reset=True)
    hip.gemm(M, N, K, A.data_ptr(), A.shape[1], akm, B.data_ptr(), B.shape[1], bkm, C.data_ptr(), N, split_k=split,
             workspace=None if ws is None else ws.data_ptr(), accumulate=acc, bias=bias.data_ptr() if use_bias else None,
             act=1 if use_bias else 0, a_colsum=cs.data_ptr() if colsum else None)
    disp = {k: v for k, v in hip.dispatch_counts().items() if v}
    Ad = (A.double().T if akm else A.double())
    ref = Ad @ (B.double() if bkm else B.double().T)
    if use_bias: ref = torch.relu(ref + bias.double())
    if acc: ref = ref + C0.double()
    err = (C.double() - ref).abs().max().item()
    tol = 4e-6 * np.sqrt(K) * 4
    ok = np.isfinite(err) and err <= tol
    if colsum:
        cerr = (cs.double() - (cs0.double() + Ad.sum(1))).abs().max().item()
        ok = ok and cerr <= 1e-5 * np.sqrt(K) * 4
    bad += not ok
    print(f"M={M} N={N} K={K} akm={akm} bkm={bkm} split={split} acc={int(acc)} bias={int(use_bias)} colsum={int(colsum)} err={err:.2e} {disp}{'' if ok else '  <-- BAD'}")
print("bad", bad)
