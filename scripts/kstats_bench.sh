#!/bin/bash
# Per-kernel durations of the benchmark's update legs (resident + ring-fed, no rollout / closed loop / other configs):
# scripts/kstats_bench.sh [rows to print]   (pipelines as timed: kernels overlap, durations include what they share the chip with)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --seeds 0 --no-cpu-baseline --no-profile --no-closed-loop --no-plain-copy --no-configs --no-mlp > /tmp/kst.log 2>&1
tail -2 /tmp/kst.log | cut -c1-300
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/kst/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:${1:-60}]:
    print(f'{r["Name"][:100]:100s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:9.1f} us {float(r["TotalDurationNs"]) / 1e6:9.2f} ms')
PY
