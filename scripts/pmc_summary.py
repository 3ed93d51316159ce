"""Per-kernel SQ counter table from one `rocprofv3 --pmc` pass of bench.py (mean per dispatch) -> profiles/rNN_sq_counters_vK.csv.

The pass (counters in a run of their own, the interpreter directly after `--`; 8 SQ slots + GRBM fit one pass on gfx950):

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES \
        SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d <dir> -- \
        python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-from-host --no-profile
    python3 scripts/pmc_summary.py <dir> --steps-in-run 2 --csv profiles/r03_sq_counters_v1.csv

Normalisation (MI355X: 256 CUs x 4 SIMDs = 1024 SIMDs, 8 XCDs, 32 shader engines; /opt/skills/guides/MI355X_MICROARCH.md):
  * rocprofv3 reports every counter summed over the chip.  GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the
    dispatch was in flight, so `cycles = GRBM_GUI_ACTIVE / 8` is its duration in shader cycles (SQ_BUSY_CYCLES / 32 agrees:
    it is summed over the 32 shader engines).
  * SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's matrix pipe is busy (32 per v_mfma_f32_32x32x16_bf16, 64 per
    v_mfma_f32_32x32x2_f32), summed over SIMDs:  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * cycles).
  * SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles per wave:  valu_busy = 4 * SQ_ACTIVE_INST_VALU /
    (1024 * cycles) (share of SIMD issue time spent on vector-ALU instructions, MFMA issue included), lds_busy likewise per
    CU (256), wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of resident wave time parked in s_waitcnt / barriers).
  * valu_per_mfma = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA (SQ_INSTS_VALU includes the MFMAs).
"""
import argparse
import os
import csv
import glob
import json
import subprocess
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def _digest():
    from srl_amd.provenance import kernel_sources_digest
    return kernel_sources_digest()

from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--csv", default=None, help="write the table here (+ .json with the run's configuration); default: print")
ap.add_argument("--steps-in-run", type=int, default=2)
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--rollout-len", type=int, default=128)
ap.add_argument("--chunk-rows", type=int, default=16384)
args = ap.parse_args()

agg = defaultdict(lambda: defaultdict(list))
for d in args.dirs:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))

SIMDS, CUS = 1024, 256
COLS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY",
        "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
rows = []
for name, cs in agg.items():
    m = {c: (sum(v) / len(v) if v else 0.0) for c, v in cs.items()}
    n = max(len(v) for v in cs.values())
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 or m.get("SQ_BUSY_CYCLES", 0.0) / 32.0
    insts_mfma = m.get("SQ_INSTS_MFMA", 0.0)
    row = dict(kernel=name, dispatches=n, cycles_per_dispatch=round(cyc),
               mfma_busy=round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * cyc), 4) if cyc else "",
               valu_busy=round(4 * m.get("SQ_ACTIVE_INST_VALU", 0.0) / (SIMDS * cyc), 4) if cyc and "SQ_ACTIVE_INST_VALU" in m else "",
               lds_busy=round(4 * m.get("SQ_ACTIVE_INST_LDS", 0.0) / (CUS * cyc), 4) if cyc and "SQ_ACTIVE_INST_LDS" in m else "",
               wait_share=round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 4) if m.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in m else "",
               valu_per_mfma=round((m.get("SQ_INSTS_VALU", 0.0) - insts_mfma) / insts_mfma, 2) if insts_mfma else "")
    row.update({c: round(m[c]) if c in m else "" for c in COLS})
    rows.append(row)
rows.sort(key=lambda r: -(r["cycles_per_dispatch"] * r["dispatches"]))

if args.csv is None:
    for r in rows:
        print(f"\n{r['kernel'][:150]}  dispatches={r['dispatches']}")
        for k, v in r.items():
            if k not in ("kernel", "dispatches"):
                print(f"    {k:28s} {v}")
else:
    with open(args.csv, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()) if rows else ["kernel"])
        w.writeheader()
        w.writerows(rows)
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        commit = None
    with open(args.csv[:-4] + ".json", "w") as fh:
        json.dump(dict(envs=args.envs, rollout_len=args.rollout_len, chunk_rows=args.chunk_rows, steps_in_run=args.steps_in_run,
                       commit=commit, kernel_sources=_digest(), simds=SIMDS, normalisation="mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); "
                       "see scripts/pmc_summary.py",
                       command="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY "
                               "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- python3 bench.py --steps 1 --warmup 1 "
                               "--no-cpu-baseline --no-from-host --no-profile"), fh, indent=1)
    print(f"wrote {args.csv}: {len(rows)} kernels")
