"""The C1-shaped update (bench.mlp_roofline) alone, for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, srl_amd
srl_amd.register_all()
import bench
r = bench.mlp_roofline("cuda:0", steps=5)
print(r["ms_per_step"], r["achieved"], r["frac"])
