"""Per-parameter error of srl_mlp_fwd / srl_mlp_bwd on the C1 actor tower against float64 autograd, for the float32 shape kernels
(SRL_MLP_F16=0) and the f16-piece kernels (1), at several row counts: scripts/mlp_f16_check.py [rows ...]
(child processes: the mode is read once per process)."""
import os, sys, json, subprocess
CODE = r'''
import sys, json, numpy as np, torch
from srl_amd import hip
rows = int(sys.argv[1])
chain = [(0, 4, 4, 0), (1, 4, 64, 1), (0, 64, 64, 0), (1, 64, 64, 1), (1, 64, 64, 1), (1, 64, 2, 0)]
import os
rng = np.random.default_rng(rows + 1000003 * int(os.environ.get('SEED', '0')))
host, keep, desc, dev_g = [], [], [], []
for kind, i, o, act in chain:
    if kind == 0:
        w, b = 1 + 0.1 * rng.standard_normal(i), 0.1 * rng.standard_normal(i)
    else:
        w, b = rng.standard_normal((o, i)) / np.sqrt(i), 0.1 * rng.standard_normal(o)
    w, b = torch.from_numpy(w.astype(np.float32)), torch.from_numpy(b.astype(np.float32))
    host += [w, b]
    dw, db = w.cuda(), b.cuda()
    gw, gb = torch.zeros_like(dw), torch.zeros_like(db)
    keep += [dw, db]; dev_g += [gw, gb]
    desc.append((kind, i, o, act, dw.data_ptr(), db.data_ptr(), gw.data_ptr(), gb.data_ptr()))
arr = hip.mlp_layers(desc)
x = torch.from_numpy(rng.standard_normal((rows, 4)).astype(np.float32))
dy = torch.from_numpy(rng.standard_normal((rows, 2)).astype(np.float32))
dx, ddy = x.cuda(), dy.cuda()
y = torch.empty(rows, 2, device="cuda")
hip.mlp_fwd(arr, dx.data_ptr(), 4, rows, 0, 0, y.data_ptr(), 2)
hip.mlp_bwd(arr, dx.data_ptr(), 4, rows, 0, 0, ddy.data_ptr(), 2)
torch.cuda.synchronize()
p64 = [p.double().requires_grad_(True) for p in host]
h = x.double()
for li, (kind, i, o, act) in enumerate(chain):
    w, b = p64[2 * li], p64[2 * li + 1]
    h = torch.nn.functional.layer_norm(h, (i,), w, b, 1e-5) if kind == 0 else h @ w.t() + b
    h = torch.relu(h) if act == 1 else h
h.backward(dy.double())
# gates that an error of 2e-6 of the layer's largest pre-activation could flip (float64 forward)
amb, hh = [], x.double()
with torch.no_grad():
    for li, (kind, i, o, act) in enumerate(chain):
        w, b = p64[2 * li], p64[2 * li + 1]
        hh = torch.nn.functional.layer_norm(hh, (i,), w, b, 1e-5) if kind == 0 else hh @ w.t() + b
        if act == 1:
            amb.append(int((hh.abs() < 2e-6 * hh.abs().max()).sum()))
            hh = torch.relu(hh)
errs = [float((g.cpu().double() - p.grad).abs().max() / (p.grad.abs().max() + 1e-30)) for g, p in zip(dev_g, p64)]
print(json.dumps(dict(err_y=float((y.cpu().double() - h.detach()).abs().max()), errs=errs, amb=amb)))
'''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rows in [int(a) for a in sys.argv[1:]] or [4096, 32768, 131072]:
    for mode in os.environ.get("MODES", "0 1").split():
        r = subprocess.run([sys.executable, "-c", CODE, str(rows)], env=dict(os.environ, SRL_MLP_F16=mode, PYTHONPATH=root),
                           capture_output=True, text=True)
        if r.returncode:
            print(rows, mode, "FAILED", r.stderr[-600:]); continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"rows {rows:7d} F16={mode}  err_y {d['err_y']:.2e}  grads " + " ".join(f"{e:.1e}" for e in d["errs"]) + f"  ambiguous gates {d['amb']}", flush=True)
