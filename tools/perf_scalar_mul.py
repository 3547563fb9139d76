#!/usr/bin/env python3
"""Wall time of n independent scalar multiplications with distinct scalars (the h1 loop,
go/bp/main.go:156-159; here through bn256_random_bases = k_i * G0, the same kernels)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import crypto_amd  # noqa: E402

eng = crypto_amd.Engine(0)
eng.random_bases(1, 64).free()
for k in (int(a) for a in (sys.argv[1:] or ["6", "10", "14", "15", "16", "18", "20"])):
    n = 1 << k
    best = 1e9
    for r in range(4):
        t0 = time.perf_counter()
        g = eng.random_bases(0x5EED0001 + r, n)
        best = min(best, time.perf_counter() - t0)
        g.free()
    print(f"n=2^{k}: {best * 1e3:8.3f} ms ({n / best / 1e6:7.2f} Mpts/s)", flush=True)
eng.close()
