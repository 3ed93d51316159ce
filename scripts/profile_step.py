"""Per-launch timing of one trainer step (HIP events on the launch stream), GEMMs broken down by shape."""
import os
import sys


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import srl_amd
from srl_amd import hip
from srl_amd.api import config, trainer as trainer_api

srl_amd.register_all()
T, B = int(os.environ.get("T", 128)), int(os.environ.get("B", 512))
trainer = trainer_api.make(config.Trainer("mappo", args=dict(bench.TRAINER, chunk_rows=int(os.environ.get("CHUNK", 16384)))),
                           config.Policy("actor-critic", args=bench.POLICY))
sample = bench.device_sample(1, T, B, "cuda:0")
trainer.step(sample)
trainer.step(sample)
_gemm = hip.gemm


def named_gemm(M, N, K, A, lda, akm, Bp, ldb, bkm, C, ldc, **kw):
    """The product path's own call (all arguments forwarded, bias-gradient column sums included) under a per-shape
    scope tag: only the label changes."""
    tag = f"gemm M={M} N={N} K={K} {'T' if akm else 'N'}{'T' if bkm else 'N'} split={kw.get('split_k', 1)}"
    with hip._scope(tag, 2.0 * M * N * K):
        prof, hip._prof = hip._prof, None  # the inner scope would double-count
        try:
            _gemm(M, N, K, A, lda, akm, Bp, ldb, bkm, C, ldc, **kw)
        finally:
            hip._prof = prof


hip.gemm = named_gemm
prof = hip.KernelProfile()
hip.set_profile(prof)
trainer.step(sample)
hip.set_profile(None)
summ = prof.summary()
tot = sum(v["ms"] for v in summ.values())
print(f"total event time {tot:.2f} ms")
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
    tf = v["work"] / (v["ms"] * 1e-3) / 1e12 if v["work"] and (k.startswith("gemm") or k.startswith("conv_")) else 0
    print(f"{v['ms']:9.3f} ms  calls={v['calls']:3d}  {tf:6.1f} TF  {k}")
