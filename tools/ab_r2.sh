#!/bin/bash
# Same-box A/B against round 2's libraries (ab/r2/*.so, built from commit 5b680ed with crypto_amd/build.py):
#   bash tools/ab_r2.sh <command...>     runs the command twice per build, alternating
# e.g. bash tools/ab_r2.sh python3 tools/wnla_bench.py
for rep in 1 2; do
  for v in r2 r3; do
    if [ $v = r2 ]; then export BN256MSM_LIB=$PWD/ab/r2/libbn256msm.so BPAMD_LIB=$PWD/ab/r2/libbpamd.so; else unset BN256MSM_LIB BPAMD_LIB; fi
    echo "== $v"
    timeout -k 10 300 "$@" 2>/dev/null | tail -${AB_TAIL:-1}
  done
done
