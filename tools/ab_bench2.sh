#!/bin/bash
# two builds under bench.py's default command and for one blocking call, alternating on one box
# (crypto_amd/libbn256msm_base.so = the build to compare against)
set -e -o pipefail
for round in 1 2 3 4; do
for lib in _base ""; do
  export BN256MSM_LIB=$PWD/crypto_amd/libbn256msm$lib.so
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 > /tmp/ab.json
  python3 - "$lib" <<'PY'
import json,sys
j=json.load(open('/tmp/ab.json'))
p=j['phases_ms']
print(f"case lib=[{sys.argv[1]:>5}]: value {j['value']/1e6:7.1f} Mpts/s  ms/step {j['ms_per_step']:.4f}  single {j['single_call']['median_ms']:.4f}  passB {p['sort_pass_b']:.4f} k_acc {p['accumulate']:.4f}  total {p['device_total']:.4f}")
PY
done
done
