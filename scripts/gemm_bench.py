"""Micro-benchmark of srl_gemm / implicit conv entry points (TFLOP/s per shape and orientation)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srl_amd import hip

DEV = "cuda:0"


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def gemm_case(M, N, K, akm, bkm, dact=False, split=1):
    A = torch.randn((K, M) if akm else (M, K), device=DEV)
    B = torch.randn((K, N) if bkm else (N, K), device=DEV)
    C = torch.empty((M, N), device=DEV)
    Y = torch.randn((M, N), device=DEV) if dact else None
    ws = torch.empty(split * M * N, device=DEV) if split > 1 else None
    fn = lambda: hip.gemm(M, N, K, A.data_ptr(), A.shape[1], akm, B.data_ptr(), B.shape[1], bkm, C.data_ptr(), N,
                          dact_src=Y.data_ptr() if dact else None, ld_dact=N, dact=1 if dact else 0, split_k=split,
                          workspace=ws.data_ptr() if ws is not None else None)
    ms = timeit(fn)
    print(f"gemm M={M:8d} N={N:5d} K={K:7d} akm={akm} bkm={bkm} dact={int(dact)} split={split:3d}: {ms:8.3f} ms "
          f"{2.0 * M * N * K / ms / 1e9:7.1f} TF", flush=True)


def conv_cases(n=16384):
    for (H, Cin, k, s, Cout) in ((20, 32, 4, 2, 64), (9, 64, 3, 1, 64)):
        d = hip.conv_desc(n, H, H, Cin, k, k, s, Cout, act=1)
        OH = (H - k) // s + 1
        x = torch.randn((n, H, H, Cin), device=DEV)
        w = torch.randn((Cout, k, k, Cin), device=DEV)
        b = torch.randn(Cout, device=DEV)
        y = torch.empty((n, OH, OH, Cout), device=DEV)
        dz = torch.randn((n, OH, OH, Cout), device=DEV)
        gw = torch.zeros_like(w)
        ws = torch.empty(max(hip.conv2d_wgrad_workspace(d), 1), device=DEV)
        wt = torch.empty(hip.conv2d_dgrad_weight_elems(d), device=DEV)
        dx = torch.empty_like(x)
        hip.conv2d_dgrad_repack(d, w.data_ptr(), wt.data_ptr())
        fl = 2.0 * n * OH * OH * Cout * k * k * Cin
        for name, fn in (("fwd", lambda: hip.conv2d_nhwc_fwd(d, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr())),
                         ("wgrad", lambda: hip.conv2d_nhwc_wgrad(d, x.data_ptr(), dz.data_ptr(), gw.data_ptr(), ws.data_ptr())),
                         ("dgrad", lambda: hip.conv2d_nhwc_dgrad(d, dz.data_ptr(), wt.data_ptr(), x.data_ptr(), 1, dx.data_ptr())),
                         ("dgrad, no activation mask", lambda: hip.conv2d_nhwc_dgrad(d, dz.data_ptr(), wt.data_ptr(), None, 0, dx.data_ptr()))):
            ms = timeit(fn)
            print(f"conv H={H} Cin={Cin} k={k} s={s} Cout={Cout} {name:6s}: {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
    # first layer
    d = hip.conv_desc(n, 84, 84, 4, 8, 8, 4, 32, act=1)
    obs = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device=DEV)
    g, bt = torch.randn((4, 84, 84), device=DEV), torch.randn((4, 84, 84), device=DEV)
    w, b = torch.randn((32, 4, 8, 8), device=DEV), torch.randn(32, device=DEV)
    mean, rstd = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    hip.obs_ln_stats(obs.data_ptr(), True, n, 4 * 84 * 84, mean.data_ptr(), rstd.data_ptr())
    y = torch.empty((n, 20, 20, 32), device=DEV)
    dz = torch.randn((n, 20, 20, 32), device=DEV)
    outs = [torch.zeros_like(w), torch.zeros(32, device=DEV), torch.zeros_like(g), torch.zeros_like(bt)]
    ws = torch.empty(hip.conv2d_obs_bwd_workspace(d), device=DEV)
    fl = 2.0 * n * 400 * 32 * 256
    ms = timeit(lambda: hip.conv2d_obs_fwd(d, obs.data_ptr(), True, mean.data_ptr(), rstd.data_ptr(), g.data_ptr(),
                                           bt.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr()))
    print(f"conv1 obs fwd (planar, direct): {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
    # space-to-depth'd, channels-last variants
    d2 = hip.conv_desc(n, 21, 21, 64, 2, 2, 1, 32, act=1)
    s2d = torch.empty_like(obs)
    ms = timeit(lambda: hip.obs_space_to_depth(obs.data_ptr(), True, n, 4, 84, 84, 4, s2d.data_ptr(), mean.data_ptr(),
                                               rstd.data_ptr()))
    print(f"obs space-to-depth + stats: {ms:8.3f} ms", flush=True)
    fws = torch.empty(hip.conv2d_obs_fwd_workspace(d2), device=DEV)
    for tag, wsp in (("direct", None), ("position-batched", fws.data_ptr())):
        ms = timeit(lambda: hip.conv2d_obs_fwd(d2, s2d.data_ptr(), True, mean.data_ptr(), rstd.data_ptr(), g.data_ptr(),
                                               bt.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), channels_last=True,
                                               ws_ptr=wsp))
        print(f"conv1 s2d fwd ({tag}): {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
    ws2 = torch.empty(hip.conv2d_obs_bwd_workspace(d2), device=DEV)
    ms = timeit(lambda: hip.conv2d_obs_bwd(d2, s2d.data_ptr(), True, mean.data_ptr(), rstd.data_ptr(), g.data_ptr(),
                                           bt.data_ptr(), w.data_ptr(), dz.data_ptr(), *[o.data_ptr() for o in outs],
                                           ws2.data_ptr(), channels_last=True))
    print(f"conv1 s2d bwd: {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
    ms = timeit(lambda: hip.conv2d_obs_bwd(d, obs.data_ptr(), True, mean.data_ptr(), rstd.data_ptr(), g.data_ptr(),
                                           bt.data_ptr(), w.data_ptr(), dz.data_ptr(), *[o.data_ptr() for o in outs],
                                           ws.data_ptr()))
    print(f"conv1 obs bwd: {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "conv"]
    if "gemm" in which:
        for M, N, K in ((4096, 4096, 4096), (16384, 512, 3136), (16384, 3136, 512)):
            for akm, bkm in ((0, 0), (0, 1), (1, 1)):
                gemm_case(M, N, K, akm, bkm)
        gemm_case(16384, 3136, 512, 0, 1, dact=True)
        gemm_case(512, 3136, 16384, 1, 1, split=5)
    if "conv" in which:
        conv_cases()
