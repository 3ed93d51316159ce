"""The C1-shaped update (bench.mlp_roofline) alone, for rocprofv3 --kernel-trace --stats.  argv: chunk_rows pipelines (optional)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, srl_amd
srl_amd.register_all()
import bench
targs = {}
if len(sys.argv) > 1:
    targs["chunk_rows"] = int(sys.argv[1])
if len(sys.argv) > 2:
    targs["pipelines"] = int(sys.argv[2])
r = bench.mlp_roofline("cuda:0", steps=5, trainer_args=targs)
print(targs, r["ms_per_step"], r["achieved"], r["frac"])
