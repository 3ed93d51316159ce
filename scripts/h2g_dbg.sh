#!/bin/bash
# leave-out timings of h2gemm_kernel on the benchmark's own launches (SRL_H2G_DBG: 1 no DMA, 2 no fragment reads / MFMAs, 4 no stores)
cd "$GRAFT_REPO_ROOT"
export SRL_PIPELINES=1 SRL_WGRAD_STREAM=0
for w in "$@"; do
  echo "=== SRL_H2G_DBG=$w"
  SRL_H2G_DBG=$w bash scripts/kstats.sh bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-from-host --seeds 0 --no-mlp 2>&1 | grep -i "h2gemm\|error" | cut -c1-150
done
