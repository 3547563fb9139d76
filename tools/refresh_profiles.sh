#!/bin/bash
# Collects what profiles/rNN_* is made of, on the GPU box (run through gpurun from the repo root):
#   bash tools/refresh_profiles.sh
# then, here:  python tools/pmc_summary.py profiles/rNN_pmc_2p20.json gpurun_out/pmc_*
# One rocprofv3 --pmc pass per run (the pool's limits: <= 8 SQ_, <= 4 TCC_ counters per pass).
set -e -o pipefail
export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/prof_kt $O/prof_kt1 $O/pmc_fetch $O/pmc_write $O/pmc_grbm $O/pmc_sq1 $O/pmc_sq2
# the driver's own command (BENCH_rNN.json: `python3 bench.py --gpus 1 --steps 20 --warmup 5`)
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_2p20.json 2> $O/bench_2p20.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -o kt -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 > $O/bench_kt.json 2> $O/bench_kt.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt1 -o kt -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 --depth 1 > $O/bench_kt1.json 2> $O/bench_kt1.err
B="python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --depth 1"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2> $O/pmc_fetch.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > /dev/null 2> $O/pmc_write.err
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm -- $B > /dev/null 2> $O/pmc_grbm.err
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_sq1 -- $B > /dev/null 2> $O/pmc_sq1.err
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq2 -- $B > /dev/null 2> $O/pmc_sq2.err
# element-wise rows (h1 loop, generator fold): kernel trace + VALU instruction counts of the two lane kernels
E="python3 tools/perf_elementwise.py 20"
rm -rf $O/prof_ew $O/pmc_ew
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ew -o kt -- $E > $O/elementwise.txt 2> $O/elementwise.err
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_ew -- $E > /dev/null 2> $O/pmc_ew.err
find $O/prof_ew -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_elementwise.csv \;
# the prover flows on this box (BASELINE.json configs[0], configs[3])
timeout -k 10 120 python3 tools/bp_bench.py > $O/bp_bench.json 2>/dev/null
BP_PROFILE=2 timeout -k 10 120 python3 tools/wnla_bench.py > $O/wnla_bench.txt 2>&1
rm -rf $O/wnla_kt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wnla_kt -o kt -- python3 tools/wnla_bench.py 20 > $O/wnla_kt.txt 2> $O/wnla_kt.err
find $O/wnla_kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_wnla.csv \;
find $O/prof_kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof_kt1 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_depth1.csv \;
cat $O/bench_2p20.json
