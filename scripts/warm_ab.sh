#!/bin/bash
# A/B of the warm-ahead ring (h2gemm.h WARM) on the benchmark's own launches, one launch at a time: SRL_H2G_WARM = 0 (plain ring), then distances.
cd "$GRAFT_REPO_ROOT"
export SRL_PIPELINES=1 SRL_WGRAD_STREAM=0
for w in "$@"; do
  echo "=== SRL_H2G_WARM=$w"
  SRL_H2G_WARM=$w bash scripts/kstats.sh bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-from-host --seeds 0 --no-mlp 2>&1 | grep -i "h2gemm\|gemm3\|ms_per_step\|error" | cut -c1-150
done
