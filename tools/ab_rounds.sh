#!/bin/bash
# Round 2's tree (ab/r2tree: `git archive 9c7408f`, built with its own crypto_amd/build.py) against this tree on ONE
# box: each round's own default bench.py, alternating, plus its WNLA and go/bp tools.  Lines go to gpurun_out/ab_rounds_*.
# Here, first (ab/ is git-ignored):
#   mkdir -p ab/r2tree && git archive 9c7408f | tar -x -C ab/r2tree
#   (cd ab/r2tree && python -m crypto_amd.build && python -c "import __graft_entry__ as g; g.build()")
# then `gpurun -- bash tools/ab_rounds.sh` and, here, `python tools/ab_rounds_report.py`.
set -e -o pipefail
O=$PWD/gpurun_out
rm -f $O/ab_rounds_r2.jsonl $O/ab_rounds_r3.jsonl $O/ab_rounds_flows.txt
for rep in 1 2; do
  (cd ab/r2tree && timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 2>/dev/null | tail -1 >> $O/ab_rounds_r2.jsonl)
  timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 2>/dev/null | tail -1 >> $O/ab_rounds_r3.jsonl
  (cd ab/r2tree && echo "r2 wnla $(timeout -k 10 120 python3 tools/wnla_bench.py 20 2>/dev/null | tail -1)" >> $O/ab_rounds_flows.txt && echo "r2 bp $(timeout -k 10 120 python3 tools/bp_bench.py 2>/dev/null | tail -1)" >> $O/ab_rounds_flows.txt && echo "r2 ew $(timeout -k 10 120 python3 tools/perf_elementwise.py 20 2>/dev/null | tail -1)" >> $O/ab_rounds_flows.txt)
  echo "r3 wnla $(timeout -k 10 120 python3 tools/wnla_bench.py 20 2>/dev/null | tail -1)" >> $O/ab_rounds_flows.txt
  echo "r3 bp $(timeout -k 10 120 python3 tools/bp_bench.py 2>/dev/null | tail -1)" >> $O/ab_rounds_flows.txt
  echo "r3 ew $(timeout -k 10 120 python3 tools/perf_elementwise.py 20 2>/dev/null | tail -1)" >> $O/ab_rounds_flows.txt
done
cat $O/ab_rounds_flows.txt
