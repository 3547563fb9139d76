#!/bin/bash
# One library, one box, several values of one environment switch, through bench.py's timed loop and blocking calls:
#   bash tools/ab_env.sh BN256_FOLD2 0 1 3
export TMPDIR=/tmp
O=gpurun_out
VAR=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    export $VAR=$v
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 60 > $O/ab_env_$v.json 2>/dev/null || exit 1
    python3 -c "
import json;d=json.loads(open('$O/ab_env_$v.json').read().strip().splitlines()[-1]);p=d['phases_ms']
print('$VAR=$v', round(d['value']/1e6,1), 'Mpts/s  step', round(d['ms_per_step'],4), ' single', round(d['single_call']['median_ms'],4), ' acc', round(p['accumulate'],3), 'fix', round(p['fix'],3), 'reduce', round(p['reduce'],3))"
  done
done
