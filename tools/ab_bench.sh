#!/bin/bash
# A/B of two builds of the library on ONE box through bench.py's timed loop (three calls in flight) and its
# blocking calls:  bash tools/ab_bench.sh   (crypto_amd/libbn256msm_base.so = the build to compare against;
# environment switches given as arguments, e.g. BN256_FOLD2=0, apply to both)
export TMPDIR=/tmp
O=gpurun_out
for a in "$@"; do export "$a"; done
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export BN256MSM_LIB=$PWD/crypto_amd/libbn256msm_base.so; else unset BN256MSM_LIB; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --steps 60 > $O/ab_bench_$v.json 2>/dev/null || exit 1
    python3 -c "
import json;d=json.loads(open('$O/ab_bench_$v.json').read().strip().splitlines()[-1]);p=d['phases_ms']
print('$v', round(d['value']/1e6,1), 'Mpts/s  step', round(d['ms_per_step'],4), ' single', round(d['single_call']['median_ms'],4), ' acc', round(p['accumulate'],3), 'fix', round(p['fix'],3), 'reduce', round(p['reduce'],3))"
  done
done
