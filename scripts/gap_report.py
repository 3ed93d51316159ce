"""Idle time of the GPU inside the benchmark's timed updates, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 bench.py --steps 3 --warmup 1 --seeds 0 --no-cpu-baseline \
        --no-profile --no-from-host --no-closed-loop --no-plain-copy; python3 scripts/gap_report.py /tmp/kt
Updates are delimited by the optimiser kernel (adam_step); per update: span, union of busy intervals, the largest gaps and the kernels
on both sides of them."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
ends = [i for i, r in enumerate(rows) if "adam" in r[2].lower()]
print(len(rows), "kernels,", len(ends), "optimiser steps")
for a, b in list(zip(ends[:-1], ends[1:])):
    seg = rows[a + 1:b + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    busy, cur_s, cur_e, gaps = 0, seg[0][0], seg[0][1], []
    last_name = seg[0][2]
    for s, e, n in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name[:50], n[:50], (cur_e - t0) / 1e6))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= cur_e:
            last_name = n
    busy += cur_e - cur_s
    print(f"update: span {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
    for g in sorted(gaps, reverse=True)[:8]:
        print(f"   gap {g[0] / 1e3:8.1f} us at +{g[3]:6.2f} ms   after {g[1]}   before {g[2]}")
if len(sys.argv) > 2:   # per-kernel totals of the LAST update (any second argument)
    from collections import defaultdict
    tot, cnt = defaultdict(int), defaultdict(int)
    for s, e, n in rows[ends[-2] + 1:ends[-1] + 1]:
        k = n.split("(")[0][:60]
        tot[k] += e - s
        cnt[k] += 1
    print(f"last update: {sum(cnt.values())} launches, sum of durations {sum(tot.values()) / 1e6:.2f} ms")
    for k in sorted(tot, key=tot.get, reverse=True)[:40]:
        print(f"   {tot[k] / 1e3:9.1f} us  x{cnt[k]:<4d} {k}")
if len(sys.argv) > 3:   # the first launches of the last update, in order (any third argument = how many)
    seg = rows[ends[-2] + 1:ends[-1] + 1]
    for s, e, n in seg[:int(sys.argv[3])] + seg[-12:]:
        print(f"   +{(s - seg[0][0]) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  {n[:110]}")
