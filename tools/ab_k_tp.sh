#!/bin/bash
# chunk length K of k_accumulate (BN256_TUNE_K; 52 = one round of four waves per SIMD at 2^20 points) under bench.py's
# default command (four calls in flight), alternating on one box
set -e -o pipefail
for round in 1 2; do
for k in "" 26 35 64 70 78 104 156; do
  if [ -z "$k" ]; then unset BN256_TUNE_K; else export BN256_TUNE_K=$k; fi
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 > /tmp/ab.json
  python3 - "$k" <<'PY'
import json,sys
j=json.load(open('/tmp/ab.json'))
print(f"case K=[{sys.argv[1]:>4}]: value {j['value']/1e6:7.1f} Mpts/s  ms/step {j['ms_per_step']:.4f}  single {j['single_call']['median_ms']:.4f}  k_acc {j['phases_ms']['accumulate']:.4f} fix {j['phases_ms']['fix']:.4f}")
PY
done
done
