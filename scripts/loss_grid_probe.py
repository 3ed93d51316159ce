"""ppo_loss_kernel under SRL_LOSS_GRID: scripts/loss_grid_probe.py <grid cap> [rows]; average launch time by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["SRL_LOSS_GRID"] = sys.argv[1]
import torch
from srl_amd import hip
n = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
d = "cuda:0"
g = torch.Generator(device=d).manual_seed(0)
r = lambda: torch.randn(n, device=d, generator=g) * 0.1
new_lp, old_lp, value, old_value, adv, ret, ent = r(), r(), r(), r(), r(), r(), r()
mask = torch.ones(n, dtype=torch.uint8, device=d)
stats = torch.tensor([float(n), float(adv.double().sum()), float((adv.double() ** 2).sum())], dtype=torch.float64, device=d)
ln = torch.tensor([float(n)], dtype=torch.float64, device=d)
d1, d2, d3 = torch.empty(n, device=d), torch.empty(n, device=d), torch.empty(n, device=d)
terms = torch.zeros(hip.LT_COUNT, dtype=torch.float64, device=d)
hp = hip.PpoHparams(0.2, 3.0, 0.2, 1.0, 0.01, 10.0, 1e-5, 0, 1, 1, 0)
def run():
    hip.ppo_loss_fwd_bwd(new_lp, old_lp, value, old_value, adv, ret, ent, mask, hp, stats, ln, d1, d2, d3, terms)
for _ in range(5):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
print(os.environ.get("SRL_LOSS_GRID"), n, "us per call (memset + kernel):", e0.elapsed_time(e1) * 20, terms.tolist()[:3])
