import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import test_gpu_fullsize as F
from srl_amd import hip
from srl_amd.runtime import synthetic
sample, _ = F.device_sample(7)
for chunk in (16384, 8192):
    tr = F.make(chunk)
    hip.dispatch_tiles(reset=True)
    tr.step(synthetic.to_sample_batch(dict(sample)))
    torch.cuda.synchronize()
    print(chunk, F.T, F.B, hip.dispatch_tiles(reset=True))
