#!/usr/bin/env python3
"""Timeline of bench.py's timed loop (three calls in flight) from a rocprofv3 kernel trace: per kernel the
stream, start, duration and what fraction of it ran beside a kernel of another stream.
  python tools/overlap_report.py gpurun_out/ov_new/kt_kernel_trace.csv [first_row [rows]]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the pipelined region: from the first k_accumulate whose stream differs from the previous k_accumulate's
acc = [i for i, r in enumerate(rows) if "k_accumulate" in r["Kernel_Name"]]
streams = [rows[i]["Stream_Id"] for i in acc]
first = next((acc[j] for j in range(1, len(acc)) if streams[j] != streams[j - 1]), 0)
lo = int(sys.argv[2]) if len(sys.argv) > 2 else first
n = int(sys.argv[3]) if len(sys.argv) > 3 else 90
t0 = int(rows[lo]["Start_Timestamp"])
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"]) for r in rows]
for i in range(lo, min(lo + n, len(rows))):
    s, e, st = iv[i]
    ov = 0
    for j in range(max(0, i - 60), min(len(rows), i + 60)):
        if j != i and iv[j][2] != st:
            ov += max(0, min(e, iv[j][1]) - max(s, iv[j][0]))
    name = rows[i]["Kernel_Name"].replace("void ", "")[:28]
    print(f"{(s - t0) / 1e3:9.1f} us  stream {st:>3s}  {name:28s} {(e - s) / 1e3:8.1f} us   beside another stream's kernel: {min(1.0, ov / max(1, e - s)):.2f}")
