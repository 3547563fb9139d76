#!/usr/bin/env python3
"""Prints the kernel timeline of the LAST blocking 2^20-point call found in a rocprofv3 kernel trace
(start offset, duration and the gap before each kernel):
  rocprofv3 --kernel-trace --output-format csv -d out -o kt -- python3 tools/event_cost.py
  python tools/call_timeline.py out/kt_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_accumulate")]
want = int(sys.argv[2]) if len(sys.argv) > 2 else 40
# the call around the `want`-th accumulate launch: from the memset before its k_part to its k_pack_results
i = acc[want]
a = i
while a > 0 and not rows[a]["Kernel_Name"].startswith("void k_part"):
    a -= 1
b = i
while b < len(rows) - 1 and not rows[b]["Kernel_Name"].startswith("k_pack_results"):
    b += 1
t0 = int(rows[a]["Start_Timestamp"])
prev = t0
tot = 0.0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f +%6.1f gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"][:44]))
    tot += (e - s) / 1e3
    prev = e
print("span %.1f us, kernels %.1f us" % ((prev - t0) / 1e3, tot))
