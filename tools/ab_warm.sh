#!/bin/bash
# how much of `value` is the chip's state at the start of the timed region: the driver's command (5 warm-up steps,
# 20 timed) against longer warm-ups, alternating on one box
set -e -o pipefail
for round in 1 2 3; do
for w in 5 8 12 20 50; do
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup $w > /tmp/ab.json
  python3 - "$w" <<'PY'
import json,sys
j=json.load(open('/tmp/ab.json'))
print(f"case warmup={sys.argv[1]:>4}: value {j['value']/1e6:7.1f} Mpts/s  ms/step {j['ms_per_step']:.4f}  single {j['single_call']['median_ms']:.4f}  k_acc {j['phases_ms']['accumulate']:.4f}")
PY
done
done
