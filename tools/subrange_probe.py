#!/usr/bin/env python3
"""Sub-range calls on a large fixed-base table (its window width is fixed when the table is built, so a
short call pays the table's bucket count): wall time per call, median of 10."""
import os, statistics, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import crypto_amd
eng = crypto_amd.Engine(0)
for log2t in (20, 19):
    N = 1 << log2t
    H = eng.random_bases(0xB9901, N, flags=crypto_amd.PRECOMPUTE)
    v = eng.fr_random(0x21, N)
    for n in (1, 64, 5000, 1 << 16, 1 << 17, 1 << 18, 1 << 19):
        if n > N:
            continue
        for _ in range(2):
            eng.msm_dev(H, v.device_ptr(), n)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); eng.msm_dev(H, v.device_ptr(), n); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"table 2^{log2t}: n={n:7d}  {statistics.median(ts):7.3f} ms", flush=True)
    H.free(); v.free()
eng.close()
