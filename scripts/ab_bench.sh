#!/bin/bash
# Same-box A/B of the ring-fed headline: scripts/ab_bench.sh "<env A>" "<env B>" [rounds]   e.g.  scripts/ab_bench.sh "SRL_LN_HEADS=0" "SRL_LN_HEADS=1" 2
A=$1; B=$2; R=${3:-2}
run() { env $1 python bench.py --steps 10 --warmup 3 --seeds 0 --no-cpu-baseline --no-profile --no-closed-loop --no-plain-copy --no-configs --no-mlp 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', 'ms_per_step', d['ms_per_step'], 'resident', d.get('resident_in_hbm', {}).get('ms_per_step'), 'rollout', d.get('rollout_inference', {}).get('requests_per_s'))
"; }
for i in $(seq $R); do run "$A"; run "$B"; done
