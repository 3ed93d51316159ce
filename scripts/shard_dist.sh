#!/bin/bash
# the 512-env shard (one rank of an 8-GPU run) without and with a one-rank RCCL process group, then a kernel trace of the latter
cd "$GRAFT_REPO_ROOT"
A="--global-envs 512 --steps 20 --warmup 5 --seeds 0 --no-cpu-baseline --no-plain-copy --no-closed-loop --no-configs --no-mlp --no-profile"
python3 bench.py $A | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('plain', d['ms_per_step'], d['resident_in_hbm']['ms_per_step'])"
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
python3 bench.py $A --force-dist | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('dist ', d['ms_per_step'], d['resident_in_hbm']['ms_per_step'], d['config']['grad_buckets'])"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --global-envs 512 --steps 3 --warmup 2 --seeds 0 --no-cpu-baseline --no-profile --no-from-host --no-configs --force-dist > /tmp/kt.log 2>&1
tail -2 /tmp/kt.log | cut -c1-300
python3 $GRAFT_REPO_ROOT/scripts/gap_report.py /tmp/kt
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
ends = [i for i, r in enumerate(rows) if "adam" in r[2].lower()]
a, b = ends[-2], ends[-1]
seg = rows[a+1:b+1]
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for s, e, n in seg:
    agg[n[:70]][0] += 1; agg[n[:70]][1] += (e - s) / 1e3
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:70s} {c:4d} {us:9.1f} us")
print("kernels in the update:", len(seg))
PY
