"""First convolution on space-to-depth'd uint8 frames in isolation: forward and backward (weight gradient) entry points,
bf16 matrix-core kernels (obs_bf16.h) against the float32 MFMA kernels (SRL_OBS_BF16=0), time and algorithmic TFLOP/s.
usage: python3 scripts/obs_bench.py [n] [fwd|bwd|both] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srl_amd import hip
from gemm_bench import timeit

DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
which = sys.argv[2] if len(sys.argv) > 2 else "both"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
obs = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device=DEV)
g, bt = torch.randn((4, 84, 84), device=DEV), torch.randn((4, 84, 84), device=DEV)
w, b = torch.randn((32, 4, 8, 8), device=DEV), torch.randn(32, device=DEV)
mean, rstd = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
y = torch.empty((n, 20, 20, 32), device=DEV)
dz = torch.randn((n, 20, 20, 32), device=DEV)
outs = [torch.zeros_like(w), torch.zeros(32, device=DEV), torch.zeros_like(g), torch.zeros_like(bt)]
d2 = hip.conv_desc(n, 21, 21, 64, 2, 2, 1, 32, act=1)
s2d = torch.empty_like(obs)
hip.obs_space_to_depth(obs.data_ptr(), True, n, 4, 84, 84, 4, s2d.data_ptr(), mean.data_ptr(), rstd.data_ptr())
fws = torch.empty(hip.conv2d_obs_fwd_workspace(d2), device=DEV)
ws2 = torch.empty(hip.conv2d_obs_bwd_workspace(d2), device=DEV)
fl = 2.0 * n * 400 * 32 * 256
fwd = lambda: hip.conv2d_obs_fwd(d2, s2d.data_ptr(), True, mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), bt.data_ptr(),
                                 w.data_ptr(), b.data_ptr(), y.data_ptr(), channels_last=True, ws_ptr=fws.data_ptr())
bwd = lambda: hip.conv2d_obs_bwd(d2, s2d.data_ptr(), True, mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), bt.data_ptr(),
                                 w.data_ptr(), dz.data_ptr(), *[o.data_ptr() for o in outs], ws2.data_ptr(), channels_last=True)
for mode in os.environ.get("OBS_MODES", "bf16,f32").split(","):
    if mode == "f32":
        os.environ["SRL_OBS_BF16"] = "0"
    else:
        os.environ.pop("SRL_OBS_BF16", None)
    if which in ("fwd", "both"):
        ms = timeit(fwd, reps)
        print(f"conv1 s2d fwd [{mode}] n={n}: {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF  ({(n * 28224 + n * 51200) / ms / 1e6:7.1f} GB/s unique)", flush=True)
    if which in ("bwd", "both"):
        ms = timeit(bwd, reps)
        print(f"conv1 s2d bwd [{mode}] n={n}: {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF  ({(n * 28224 + n * 51200) / ms / 1e6:7.1f} GB/s unique)", flush=True)
