#!/bin/bash
# k_accumulate's priority step-down points (BN256_ACC_PRIO = iterations before the end of a chunk), one box
set -e -o pipefail
for round in 1 2; do
for p in "" "39,26,13" "24,12,4" "16,8,2" "12,6,2" "8,4,2" "6,3,1" "30,15,5"; do
  if [ -z "$p" ]; then unset BN256_ACC_PRIO; else export BN256_ACC_PRIO=$p; fi
  echo "prio=[$p] $(timeout -k 10 120 python3 tools/perf_msm.py --configs 20:0 --reps 12 | grep c=)"
done
done
