set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out
B="python3 tools/perf_msm.py --configs 20:0 --reps 3"
rm -rf $O/pmc_occ $O/pmc_occ2
timeout -s KILL 120 rocprofv3 --pmc MeanOccupancyPerCU --output-format csv -d $O/pmc_occ -- $B > /dev/null 2> $O/pmc_occ.err
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $O/pmc_occ2 -- $B > /dev/null 2> $O/pmc_occ2.err
