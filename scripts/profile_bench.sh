#!/bin/bash
# The profiling passes of bench.py (run on the GPU box through gpurun): kernel trace + stats of the benchmark's own command
# (ring-fed update, no CPU baseline), then the HBM and SQ counter passes in runs of their own (never combined with a trace
# domain).  usage: scripts/profile_bench.sh <tag> [bench args]
# Leaves gpurun_out/<tag>/{kernel_stats.csv,hbm_traffic.csv,hbm_traffic.json,sq_counters.csv,sq_counters.json,bench.json};
# copy what is to be judged into profiles/.
set -u
TAG=${1:-prof}; shift || true
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
# per-kernel figures: one launch at a time (the timed configuration overlaps two row-chunk pipelines and the weight gradients
# on streams of their own; a kernel's duration then includes what it shares the chip with)
export SRL_PIPELINES=1 SRL_WGRAD_STREAM=0
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --steps 2 --warmup 1 --seeds 0 --no-cpu-baseline --no-configs --no-from-host "$@" > "$OUT/bench.json" 2> "$OUT/kt.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --steps 1 --warmup 1 --seeds 0 --no-cpu-baseline --no-configs --no-from-host --no-profile "$@" > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --steps 1 --warmup 1 --seeds 0 --no-cpu-baseline --no-configs --no-from-host --no-profile "$@" > /dev/null 2> "$OUT/write.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 bench.py --steps 1 --warmup 1 --seeds 0 --no-cpu-baseline --no-configs --no-from-host --no-profile "$@" > /dev/null 2> "$OUT/sq.err"
python3 scripts/hbm_traffic.py "$OUT/fetch" "$OUT/write" --steps-in-run 2 --out "$OUT/hbm_traffic.csv"
python3 scripts/pmc_summary.py "$OUT/sq" --steps-in-run 2 --csv "$OUT/sq_counters.csv"
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
rm -rf "$OUT/kt" "$OUT/fetch" "$OUT/write" "$OUT/sq"
head -12 "$OUT/kernel_stats.csv" | cut -c1-200
