#!/bin/bash
# Same-box A/B of the 512-environment shard (BASELINE configs[1] = one rank of the 8-GPU headline) over ENVIRONMENT settings:
# scripts/ab_shard.sh "VAR=a" "VAR=b" [rounds]   (alternating runs of bench.py --global-envs 512; ms per update, ring-fed and resident)
# DIST=1: with a one-rank RCCL process group (--force-dist), the `scaling_model`'s "with collectives" leg
A=$1; B=$2; R=${3:-2}
X=""
if [ -n "$DIST" ]; then X="--force-dist"; export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541; fi
run() { env $1 python bench.py --global-envs 512 --steps 40 --warmup 8 --seeds 0 --no-cpu-baseline --no-profile --no-closed-loop --no-plain-copy --no-configs --no-mlp $X 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('[$1]', 'ms_per_step', round(d['ms_per_step'], 3), 'resident', round(d.get('resident_in_hbm', {}).get('ms_per_step', 0), 3))
"; }
for i in $(seq $R); do run "$A"; run "$B"; done
