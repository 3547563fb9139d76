#!/bin/bash
# Kernel traces of bench.py's timed loop (three calls in flight) for two builds of the library on one box:
#   bash tools/overlap_trace.sh    -> gpurun_out/ov_{base,new}/kt_kernel_trace.csv  (tools/overlap_report.py reads them)
export TMPDIR=/tmp
O=gpurun_out
for v in base new; do
  if [ $v = base ]; then export BN256MSM_LIB=$PWD/crypto_amd/libbn256msm_base.so; else unset BN256MSM_LIB; fi
  rm -rf $O/ov_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/ov_$v -o kt -- python3 bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 3 > $O/ov_$v.json 2> $O/ov_$v.err || exit 1
  tail -c 300 $O/ov_$v.json
done
