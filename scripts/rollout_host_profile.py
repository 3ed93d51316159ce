"""Host side of a stack-aware rollout tick (2048 requests: one group of the closed loop): cProfile of `rollout_async` (issue only: no
synchronisation inside) and of a whole tick, to see where the Python time of the rollout path goes."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srl_amd
from srl_amd.api import config, policy as policy_api
from srl_amd.namedarray import NamedArray
srl_amd.register_all()
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
pol = policy_api.make(config.Policy("actor-critic", args=bench.POLICY))
pol.attach_obs_ring(pol.make_obs_ring(64 * n, patch_rows=n))
planes = torch.randint(0, 256, (8, n, 1, 84, 84), dtype=torch.uint8).pin_memory()
z = lambda dt: np.zeros((n, 1), dt)
def req(t, prev):
    return policy_api.RolloutRequest(obs=NamedArray(obs=planes[t % 8], ring_prev=prev), is_evaluation=z(np.uint8), on_reset=z(np.uint8),
                                     client_id=z(np.int32), request_id=np.arange(n).reshape(n, 1), received_time=z(np.int64), buffer_index=z(np.int32))
prev = np.zeros((n, 1), np.int64)
for t in range(5):
    prev = pol.rollout(req(t, prev)).analyzed_result.obs_ref
torch.cuda.synchronize()
issue, total = [], []
pr = cProfile.Profile()
for t in range(40):
    r = req(t, prev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable()
    p = pol.rollout_async(r)
    pr.disable()
    t1 = time.perf_counter()
    prev = p.result().analyzed_result.obs_ref
    t2 = time.perf_counter()
    issue.append(t1 - t0); total.append(t2 - t0)
print(f"n={n}: issue (host only) {1e3 * np.median(issue):.3f} ms, whole tick {1e3 * np.median(total):.3f} ms")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[-9500:])
