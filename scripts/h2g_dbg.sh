#!/bin/bash
# leave-out timings of h2gemm_kernel on the benchmark's own launches (SRL_H2G_DBG: 1 no DMA, 2 no fragment reads / MFMAs, 4 no stores)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export SRL_PIPELINES=1 SRL_WGRAD_STREAM=0
for w in "$@"; do
  echo "=== SRL_H2G_DBG=$w"
  rm -rf /tmp/kst
  SRL_H2G_DBG=$w rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-from-host --seeds 0 --no-mlp > /tmp/kst.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/kst/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "h2gemm" in r["Name"]:
        print(f'{r["Name"][:70]:70s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
done
