#!/bin/bash
# closed-loop leg of bench.py under different flags / env (same box): scripts/ab_closed.sh "<env or flags A>" "<B>" ...
for cfg in "$@"; do
  envs=""; flags=""
  for w in $cfg; do case $w in --*) flags="$flags $w";; [0-9]*) flags="$flags $w";; *) envs="$envs $w";; esac; done
  env $envs python bench.py --steps 6 --warmup 2 --seeds 0 --no-cpu-baseline --no-profile --no-plain-copy --no-configs --no-mlp $flags 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d.get('closed_loop') or {}; r = d.get('rollout_inference') or {}
        print('[$cfg]', 'update', round(d['ms_per_step'], 2), 'closed', round(c.get('ms_per_iteration', 0), 1), 'ms ->', round(c.get('value', 0) / 1e6, 3), 'M; rollout', round(r.get('value', 0) / 1e6, 3), 'M req/s', r.get('ms_per_call'))
"
done
