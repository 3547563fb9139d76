#!/usr/bin/env python3
"""Wall time of the generator fold s_lo*lo[i] + s_hi*hi[i] (go/bp/main.go:355-356) per size.
BN256_FOLD_QUAD_MAX=<n> moves the switch between the quad and the lane-per-output kernel."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import crypto_amd  # noqa: E402

eng = crypto_amd.Engine(0)
N = 0x8fb501e34aa387f9aa6fecb86184dc212e8d8e12f82b39241a2ef45b57ac7261
s_lo, s_hi = 0x1234567 << 200 | 99, N - 0xabcdef123
for k in (int(a) for a in (sys.argv[1:] or ["13", "14", "15", "16", "17", "18", "19"])):
    n = 1 << k
    g = eng.random_bases(0x5EED0001, 2 * n)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        r = eng.fold(g, 0, g, n, n, s_lo, s_hi)
        best = min(best, time.perf_counter() - t0)
        r.free()
    t0 = time.perf_counter()
    r = eng.vector_point_mul_on_scalar(g, s_hi, 0, n)
    t1 = time.perf_counter() - t0
    r.free()
    print(f"n=2^{k}: fold {best * 1e3:8.3f} ms ({n / best / 1e6:7.2f} Mpts/s)   mul_on_scalar {t1 * 1e3:8.3f} ms", flush=True)
    g.free()
eng.close()
