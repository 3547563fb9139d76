#!/bin/bash
# wave priorities under bench.py's default command (four calls in flight), for one blocking call and for the prover
# flows, alternating on one box.  A case is "<library suffix>/<BN256_ACC_PRIO_ALWAYS>/<depth>"; library suffixes:
# "" = the product, _none = -DSMALL_PRIO=0 -DSORT_PRIO=0, _fr3 = -DFR_PRIO=3 (crypto_amd/libbn256msm<suffix>.so)
set -e -o pipefail
for round in 1 2 3; do
for case in "_none/0/4" "//4" "/0/4"; do
  IFS=/ read lib always d <<< "$case"
  export BN256MSM_LIB=$PWD/crypto_amd/libbn256msm$lib.so
  if [ -z "$always" ]; then unset BN256_ACC_PRIO_ALWAYS; else export BN256_ACC_PRIO_ALWAYS=$always; fi
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 --depth $d > /tmp/ab.json
  python3 - "$case" <<'PY'
import json,sys
j=json.load(open('/tmp/ab.json'))
print(f"case {sys.argv[1]:>12}: value {j['value']/1e6:7.1f} Mpts/s  ms/step {j['ms_per_step']:.4f}  single {j['single_call']['median_ms']:.4f}  k_acc {j['phases_ms']['accumulate']:.4f}  total {j['phases_ms']['device_total']:.4f}")
PY
done
unset BN256_ACC_PRIO_ALWAYS
for lib in _none "" _fr3; do
  export BN256MSM_LIB=$PWD/crypto_amd/libbn256msm$lib.so
  echo "flows lib=[$lib] $(timeout -k 10 120 python3 tools/flows_ms.py)"
done
done
