#!/usr/bin/env python3
"""Folds rocprofv3 --pmc counter_collection.csv files (one pass per run, as the pool requires)
into profiles/<name>.json: per kernel, the mean counter value per launch.

  python tools/pmc_summary.py out.json dir1 [dir2 ...]      (dirs searched recursively)"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    # gpurun merges each call's output into the local directory: older passes stay, so only the newest file counts
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True),
                       key=os.path.getmtime)[-1:]:
        with open(path, newline="") as f:
            per_dispatch = defaultdict(float)  # counters split over XCD/SE instances are summed per dispatch
            names = {}
            for row in csv.DictReader(f):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"]
            for (disp, ctr), val in per_dispatch.items():
                k = re.sub(r"^void ", "", re.sub(r"\(.*", "", names[disp]))  # template arguments kept: instantiations differ
                acc[k][ctr].append(val)
res = {}
for k, ctrs in sorted(acc.items()):
    if not k.startswith("k_"):
        continue
    res[k] = {{"FETCH_SIZE": "FETCH_SIZE_KB", "WRITE_SIZE": "WRITE_SIZE_KB"}.get(c, c): sum(v) / len(v)
              for c, v in sorted(ctrs.items())}
    res[k]["launches"] = max(len(v) for v in ctrs.values())
res["_note"] = os.environ.get("PMC_NOTE") or ("rocprofv3 --pmc, one counter block per run (FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE | SQ_*), "
                "python3 bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --depth 1, 2^20 points, default window "
                "choice (c=20 from 2^20 points on); per-launch means, "
                "k_accumulate<true> is the timed workload's instantiation (tables proven free of the point at infinity; "
                "13 windows of 20 bits), k_accumulate<false> the set-up's 16-bit-window calls; "
                "instances summed per dispatch. FETCH/WRITE_SIZE in KB as reported. k_accumulate reads are random "
                "64-byte gathers that fetch their whole 128-byte L2 line, so FETCH_SIZE is reported as measured "
                "(no streaming-read correction).")
with open(out, "w") as f:
    json.dump(res, f, indent=1)
print("wrote", out, "kernels:", len(res))
