#!/bin/bash
# One environment variable over several values on one box: scripts/sweep_env.sh VAR "v1 v2 ..." shard|dist|full [rounds]
# ("-" = unset).  shard: bench.py --global-envs 512; dist: the same with a one-rank RCCL process group; full: the 4096-env headline.
V=$1; VALS=$2; LEG=${3:-shard}; R=${4:-1}
X="--steps 10 --warmup 3"
[ "$LEG" != full ] && X="--global-envs 512 --steps 40 --warmup 8"
if [ "$LEG" = dist ]; then X="$X --force-dist"; export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29543; fi
for i in $(seq $R); do for v in $VALS; do
  if [ "$v" = "-" ]; then unset $V; else export $V=$v; fi
  python bench.py $X --seeds 0 --no-cpu-baseline --no-profile --no-closed-loop --no-plain-copy --no-configs --no-mlp 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('[$LEG $V=$v]', 'ms_per_step', round(d['ms_per_step'], 3), 'resident', round(d.get('resident_in_hbm', {}).get('ms_per_step', 0), 3))
"; done; done
