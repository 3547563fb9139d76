#!/bin/bash
# A/B of two builds on ONE box for the element-wise rows (h1 loop, generator fold), alternating.
#   bash tools/ab_ew.sh   (through gpurun; crypto_amd/libbn256msm_base.so = the build to compare against)
set -e -o pipefail
O=gpurun_out
rm -f $O/abew_base.txt $O/abew_new.txt
for round in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export BN256MSM_LIB=$PWD/crypto_amd/libbn256msm_base.so; else unset BN256MSM_LIB; fi
    timeout -k 10 120 python3 tools/perf_elementwise.py 20 >> $O/abew_$v.txt 2>&1
  done
done
echo base; grep -h "n=2" $O/abew_base.txt; echo new; grep -h "n=2" $O/abew_new.txt
