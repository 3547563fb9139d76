#!/usr/bin/env python3
"""Phase sums of a blocking 2^24-point call (15 passes): where the time beyond 15 x one 2^20-sized pass goes.
  python tools/big_phases.py [--log2n 24]"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import crypto_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=24)
args = ap.parse_args()
n = 1 << args.log2n
eng = crypto_amd.Engine(0)
H = eng.random_bases(0xB1601, n, flags=crypto_amd.PRECOMPUTE)
v = eng.fr_random(0x31, n)
for timing in (False, True):
    eng.set_phase_timing(timing)
    eng.msm_dev(H, v.device_ptr(), n)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        eng.msm_dev(H, v.device_ptr(), n)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"2^{args.log2n} blocking, phase events {'on' if timing else 'off'}: median {statistics.median(ts):.3f} ms",
          " ".join(f"{x:.3f}" for x in eng.last_timing()), flush=True)
