#!/bin/bash
# FETCH_SIZE of the first-layer kernels under SRL_OBS_XCD=0 / 1 (resident update, one step): scripts/obs_xcd_traffic.sh
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for x in 0 1; do
  export SRL_OBS_XCD=$x SRL_PIPELINES=1 SRL_WGRAD_STREAM=0
  rm -rf /tmp/fx$x
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fx$x -- python3 bench.py --steps 1 --warmup 1 --seeds 0 --no-cpu-baseline --no-configs --no-from-host --no-profile --no-mlp > /dev/null 2> /tmp/fx$x.err
  python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob("/tmp/fx$x/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "obs_fwd_h2" in k or "obs_bwd_h2" in k:
            tot[k[:40]] += float(r["Counter_Value"]); cnt[k[:40]] += 1
for k in tot: print("SRL_OBS_XCD=$x", k, cnt[k], "dispatches", round(tot[k] / cnt[k] / 1e3, 1), "MB per dispatch (FETCH_SIZE KB/1000, before the guide's corrections)")
PY
done
