set -e
O=gpurun_out/sweep.txt
: > $O
for l in 14 16 18 19 20 21 22; do echo "table log2n=$l" >> $O; timeout -k 10 200 python3 tools/perf_msm.py --log2n $l --configs 0:0 --reps 6 >> $O 2>&1; done
for l in 16 18 20; do echo "plain log2n=$l" >> $O; timeout -k 10 200 python3 tools/perf_msm.py --log2n $l --configs 0:0 --reps 6 --no-precompute >> $O 2>&1; done
timeout -k 10 300 python3 tools/bppp_commit_bench.py >> $O 2>&1
