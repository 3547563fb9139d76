#!/bin/bash
# Here, after tools/refresh_profiles.sh ran on the GPU box: copies what it collected under gpurun_out/ into
# profiles/<tag>_* and regenerates the number blocks and the round comparison.
#   bash tools/finalize_profiles.sh r03 [r02]
set -e
T=${1:-r03}; P=${2:-r02}; O=gpurun_out
cp $O/bench_2p20.json profiles/${T}_bench_2p20.json
cp $O/kernel_stats.csv profiles/${T}_rocprofv3_kernel_stats_bench_2p20.csv
cp $O/kernel_stats_depth1.csv profiles/${T}_rocprofv3_kernel_stats_bench_2p20_depth1.csv
cp $O/kernel_stats_elementwise.csv profiles/${T}_rocprofv3_kernel_stats_elementwise_2p20.csv
[ -f $O/kernel_stats_wnla.csv ] && cp $O/kernel_stats_wnla.csv profiles/${T}_rocprofv3_kernel_stats_wnla_2p20.csv
python tools/pmc_summary.py profiles/${T}_pmc_2p20.json $O/pmc_fetch $O/pmc_write $O/pmc_grbm $O/pmc_sq1 $O/pmc_sq2
PMC_NOTE="rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- python3 tools/perf_elementwise.py 20 (h1 loop: bn256_scalar_mul_batch over 2^20 points; fold: 2^19 outputs); per-launch means, instances summed per dispatch" \
  python tools/pmc_summary.py profiles/${T}_pmc_elementwise_2p20.json $O/pmc_ew
python tools/doc_numbers.py $T
python tools/compare_rounds.py $P $T > /dev/null
echo "profiles/${T}_* written"
