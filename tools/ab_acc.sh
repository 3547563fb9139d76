#!/bin/bash
# A/B of two builds of the library on ONE box: phase times (three alternating rounds, best of 12 calls each)
# and SQ_INSTS_VALU of k_accumulate.
#   bash tools/ab_acc.sh   (through gpurun; crypto_amd/libbn256msm_base.so = the build to compare against)
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out
rm -f $O/ab_base.txt $O/ab_new.txt
for round in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export BN256MSM_LIB=$PWD/crypto_amd/libbn256msm_base.so; else unset BN256MSM_LIB; fi
    timeout -k 10 120 python3 tools/perf_msm.py --configs 20:0 --reps 12 >> $O/ab_$v.txt 2>&1
  done
done
for v in base new; do
  if [ $v = base ]; then export BN256MSM_LIB=$PWD/crypto_amd/libbn256msm_base.so; else unset BN256MSM_LIB; fi
  rm -rf $O/ab_pmc_$v
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/ab_pmc_$v -- python3 tools/perf_msm.py --configs 20:0 --reps 3 > /dev/null 2> $O/ab_pmc_$v.err
done
echo base; grep -h "c=" $O/ab_base.txt; echo new; grep -h "c=" $O/ab_new.txt
