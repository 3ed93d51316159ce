"""Forward activations of the NatureCNN trunk on the device against a float64 torch restatement, layer by layer (relative
rms / max error), for the kernel set chosen by SRL_MFMA / SRL_OBS_BF16.  usage: python3 scripts/act_error_probe.py [rows] [f32]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if len(sys.argv) > 2 and sys.argv[2] == "f32":
    os.environ["SRL_MFMA"] = "f32"
    os.environ["SRL_OBS_BF16"] = "0"
import srl_amd
from srl_amd.api import config, policy as policy_api
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_trainer import CNN_POLICY

srl_amd.register_all()
pol = policy_api.make(config.Policy("actor-critic", args=CNN_POLICY))
sd = {k: v.double() for k, v in pol.get_checkpoint()["state_dict"].items()}
rng = np.random.default_rng(0)
frames = rng.integers(0, 256, size=(rows, 4, 84, 84), dtype=np.uint8)
net = pol.net
logits, value = net.forward({"obs": torch.from_numpy(frames).to("cuda:0")}, rows, keep_tape=True)
torch.cuda.synchronize()
P = "obs_modules_dict.obs."
x = torch.from_numpy(frames).double()
x = F.layer_norm(x, (4, 84, 84), sd[P + "0.weight"], sd[P + "0.bias"])
ref = {}
for idx, stride in ((0, 4), (2, 2), (4, 1)):
    x = F.relu(F.conv2d(x, sd[f"{P}1._Convolution__model.{idx}.weight"], sd[f"{P}1._Convolution__model.{idx}.bias"], stride=stride))
    ref[f"a:{P}1._Convolution__model.{idx}.y"] = x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])  # NHWC rows
x = F.relu(F.linear(x.flatten(1), sd[P + "1._Convolution__model.7.0.weight"], sd[P + "1._Convolution__model.7.0.bias"]))
ref[f"a:{P}1._Convolution__model.7.0.y"] = x
x = F.layer_norm(x, (512,), sd[P + "1._Convolution__model.7.2.weight"], sd[P + "1._Convolution__model.7.2.bias"])
ref[f"a:{P}1._Convolution__model.7.2.y"] = x
print(f"rows {rows}  SRL_MFMA={os.environ.get('SRL_MFMA', 'bf16x3')}")
for name, r in ref.items():
    buf = net.ws._bufs.get(name)
    if buf is None:
        print(name, "not found; have", [k for k in net.ws._bufs if k.endswith('.y')])
        continue
    got = buf[:r.numel()].view(r.shape).cpu().double()
    e = got - r
    scale = r.pow(2).mean().sqrt()
    print(f"{name[len('a:' + P):]:40s} rms {float(scale):.3e}  err rms/rms {float(e.pow(2).mean().sqrt() / scale):.2e}  max/rms {float(e.abs().max() / scale):.2e}")
