#!/bin/bash
# A/B of the XCD start-up skew of the wide h2gemm (Linear data gradient): SRL_H2G_STAGGER in units of ~3.4 us
cd "$GRAFT_REPO_ROOT"
export SRL_PIPELINES=1 SRL_WGRAD_STREAM=0
for w in "$@"; do
  echo "=== SRL_H2G_STAGGER=$w"
  SRL_H2G_STAGGER=$w bash scripts/kstats.sh bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-from-host --seeds 0 --no-mlp 2>&1 | grep -i "h2gemm_kernel<8\|error" | cut -c1-150
done
