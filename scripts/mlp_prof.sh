#!/bin/bash
# Phase timing of mlp_bwd_mfma_kernel (shader-clock stamps by thread 0 of every workgroup) on a variant build:
#   scripts/mlp_prof.sh   (in the build container: builds srl_amd/csrc/libsrlhip_mlpprof.so; on the GPU box: runs the probe)
cd "$(dirname "$0")/.."
if ! python3 -c "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 1)" 2>/dev/null; then
  make -j8 >/dev/null
  d=srl_amd/csrc/build_mlpprof; mkdir -p $d
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -DSRL_MLP_PROF -c -o $d/mlp_small.o srl_amd/csrc/mlp_small.hip
  objs=$(ls srl_amd/csrc/build/*.o | grep -v -e /mlp_small.o)
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -o srl_amd/csrc/libsrlhip_mlpprof.so $objs $d/mlp_small.o
  echo built srl_amd/csrc/libsrlhip_mlpprof.so; exit 0
fi
SRL_HIP_LIB=$PWD/srl_amd/csrc/libsrlhip_mlpprof.so python3 - "$@" <<'PY'
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
sys.argv = ["x", str(rows)]
exec(open("scripts/mlp_chain_bench.py").read())
from srl_amd import hip
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (256 * 40))()
fn = hip.lib().mlp_prof_dump
fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
assert fn(buf, 256 * 40) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(256, 40).astype(np.float64)
d = lambda i, j: np.median(a[:, j] - a[:, i]) / 100.0   # s_memtime: 100 MHz constant clock -> us
print(f"rows {rows}: stage {d(0, 1):.1f} us | loop {d(1, 2):.1f} us | fold (LDS) {d(2, 3):.1f} us | global adds {d(3, 4):.1f} us | total {d(0, 4):.1f} us")
its = int(np.ceil(rows / 32 / 4 / 256))
for lin in range(3):
    b = 20 + 5 * lin
    print(f"  iteration 1, Linear {lin}: writes+colsum {np.median(a[:, b + 1] - a[:, b]):.0f} | barrier {np.median(a[:, b + 2] - a[:, b + 1]):.0f} | blocks {np.median(a[:, b + 3] - a[:, b + 2]):.0f} | barrier {np.median(a[:, b + 4] - a[:, b + 3]):.0f} cycles")
print("  (layer to layer:", [float(np.median(a[:, 20 + 5 * l] - a[:, 20 + 5 * (l + 1) + 4])) for l in (1, 0)], "cycles from a layer's second barrier to the next layer's first stamp)")
print("iterations:", [round(float(np.median(a[:, 9 + k] - a[:, 8 + k]) / 100.0), 1) for k in range(min(its, 16) - 1)])
PY
