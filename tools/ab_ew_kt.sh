set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out
for v in base new; do
  if [ $v = base ]; then export BN256MSM_LIB=$PWD/crypto_amd/libbn256msm_base.so; else unset BN256MSM_LIB; fi
  rm -rf $O/abew_kt_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/abew_kt_$v -o kt -- python3 tools/perf_elementwise.py 20 > /dev/null 2> $O/abew_kt_$v.err
  echo $v; cut -d, -f1-4 $O/abew_kt_$v/kt_kernel_stats.csv | cut -c1-60,200- | head -5
done
