#!/bin/bash
# SQ counters of one script's kernels on the GPU box (a pass of its own, no trace domain): scripts/kpmc.sh <filter> <script.py> [args]
cd /tmp && export TMPDIR=/tmp
F=$1; S=$2; shift; shift
rm -rf /tmp/kpmc
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/kpmc/a -- python3 "$GRAFT_REPO_ROOT/$S" "$@" > /tmp/kpmc_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d /tmp/kpmc/b -- python3 "$GRAFT_REPO_ROOT/$S" "$@" > /tmp/kpmc_b.log 2>&1
tail -2 /tmp/kpmc_b.log
python3 - "$F" <<'PY'
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob("/tmp/kpmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k, f"cycles {cyc:.0f}")
    for n, v in sorted(m.items()):
        print(f"   {n:28s} {v:14.0f}   per SIMD-cycle {v / (1024 * cyc):8.4f}   per CU-cycle {v / (256 * cyc):8.4f}")
PY
