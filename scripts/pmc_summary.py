"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel name (mean per dispatch)."""
import csv
import glob
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = max(len(v) for v in cs.values())
    print(f"\n{name[:150]}  dispatches={n}")
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.0f}")
