#!/usr/bin/env python3
"""Wall-clock latency of small blocking MSM calls (the shape of a proof's late rounds: a call is launch
latency, not throughput).  Median of --reps calls per size, table-less and with a fixed-base table."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import crypto_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="2,16,64,256,1024,4096,16384")
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--only", default="", help="'plain' or 'table': one kind only (for a kernel trace)")
args = ap.parse_args()

eng = crypto_amd.Engine(0)
nmax = max(int(x) for x in args.sizes.split(","))
scal = torch.empty(nmax * 32, dtype=torch.uint8, device="cuda:0")
eng.random_scalars_into(0x5EED0002, nmax, scal.data_ptr())
for kind, flags in (("plain", 0), ("table", crypto_amd.PRECOMPUTE)):
    if args.only and args.only != kind:
        continue
    for n in (int(x) for x in args.sizes.split(",")):
        bases = eng.random_bases(0x5EED0001, n, flags=flags)
        for _ in range(3):
            eng.msm_dev(bases, scal.data_ptr(), n)
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            eng.msm_dev(bases, scal.data_ptr(), n)
            ts.append((time.perf_counter() - t0) * 1e3)
        dev = eng.last_timing()
        print(f"{kind:5s} n={n:6d}  median {statistics.median(ts):7.3f} ms  min {min(ts):7.3f} ms  device phases "
              + " ".join(f"{v:6.3f}" for v in dev), flush=True)
        bases.free()
eng.close()
