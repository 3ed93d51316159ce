"""GAE scan kernel: time per launch and achieved algorithmic GB/s over a sweep of batch sizes (T = 128)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srl_amd import hip
from srl_amd.runtime import synthetic

DEV = "cuda:0"
T = 128
print(f"{'B':>9} {'us/launch':>10} {'alg GB/s':>10} {'frac of 8 TB/s':>15}")
for B in (512, 4096, 32768, 131072, 524288, 1048576):
    arr = synthetic.make_sample_arrays(seed=1, T=T, B=B, obs_spec={}, action_dims=2, p_done=1 / 800)
    d = {k: torch.from_numpy(v).to(DEV) for k, v in arr.items()}
    adv = torch.zeros((T + 1, B, 1), device=DEV)
    ret = torch.zeros((T + 1, B, 1), device=DEV)
    stats = torch.zeros(3, dtype=torch.float64, device=DEV)
    args = (d["reward"], d["analyzed_result.value"], d["done"], d["truncated"], d["on_reset"], 0.99, 0.97, adv, ret)
    for _ in range(5):
        hip.gae_scan(*args, stats=stats)
    reps = 200 if B <= 32768 else 30
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        hip.gae_scan(*args, stats=stats)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    gbs = (19.0 * T * B + 7.0 * B) / us / 1e3
    print(f"{B:9d} {us:10.2f} {gbs:10.1f} {gbs / 8000:15.4f}", flush=True)
