#!/bin/bash
# Per-kernel durations of one script on the GPU box: scripts/kstats.sh <script.py> [args]  (rocprofv3 kernel trace + stats)
cd /tmp && export TMPDIR=/tmp
S=$1; shift
rm -rf /tmp/kst; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python3 "$GRAFT_REPO_ROOT/$S" "$@" > /tmp/kst.log 2>&1
tail -4 /tmp/kst.log
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/kst/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:9.1f} us {r["Percentage"]:>6s} %')
PY
