#!/bin/bash
# SQ counters of the fused MLP chain's launches alone (scripts/mlp_c1_probe.py): scripts/mlp_pmc.sh [rows]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/mlp_pmc; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 scripts/mlp_c1_probe.py "$@" > "$OUT/probe.txt" 2> "$OUT/sq.err"
python3 scripts/pmc_summary.py "$OUT/sq" --steps-in-run 1 --csv "$OUT/sq_counters.csv" > /dev/null
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/sq2" -- python3 scripts/mlp_c1_probe.py "$@" > /dev/null 2> "$OUT/sq2.err"
python3 - <<PY
import csv, glob, collections
for d in ("sq", "sq2"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "mlp_" not in k: continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen.add((k, r["Dispatch_Id"]))
        for k, _ in seen: cnt[k] += 1
    for k in tot:
        print(k, "dispatches", cnt[k])
        for c, v in sorted(tot[k].items()): print(f"   {c:28s} {v / cnt[k]:16.0f} per dispatch")
PY
rm -rf "$OUT/sq" "$OUT/sq2"
