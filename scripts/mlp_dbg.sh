#!/bin/bash
# leave-out timings of the fused MLP chain's backward kernel (SRL_MLP_DBG: 1 no LDS adds, 2 no weight-gradient blocks, 4 no data gradient, 8 no global adds)
cd "$GRAFT_REPO_ROOT"
for w in "$@"; do
  echo "=== SRL_MLP_DBG=$w"
  SRL_MLP_DBG=$w bash scripts/kstats.sh scripts/mlp_chain_bench.py 131072 2>&1 | grep -i "mlp_\|rows" | cut -c1-150
done
