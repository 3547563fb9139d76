#!/usr/bin/env python3
"""Chunk length K of k_accumulate for small calls: wall time of a blocking call, median of 30, per K."""
import os, statistics, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import crypto_amd
eng = crypto_amd.Engine(0)
nmax = 1 << 16
scal = torch.empty(nmax * 32, dtype=torch.uint8, device="cuda:0")
eng.random_scalars_into(0x5EED0002, nmax, scal.data_ptr())
for kind, flags in (("plain", 0), ("table", crypto_amd.PRECOMPUTE)):
    for n in (16, 64, 256, 1024, 4096, 16384, 65536):
        bases = eng.random_bases(0x5EED0001, n, flags=flags)
        row = []
        for K in (0, 2, 4, 8, 16, 32):
            eng.set_tuning(0, K)
            for _ in range(3):
                eng.msm_dev(bases, scal.data_ptr(), n)
            ts = []
            for _ in range(30):
                t0 = time.perf_counter(); eng.msm_dev(bases, scal.data_ptr(), n); ts.append((time.perf_counter() - t0) * 1e3)
            row.append(f"K={K}: {statistics.median(ts):.3f}")
        print(f"{kind:5s} n={n:6d}  " + "  ".join(row), flush=True)
        bases.free()
eng.close()
