#!/usr/bin/env python3
"""Wall time of the two throughput-sized element-wise rows: n independent scalar multiplications with distinct
scalars (the h1 loop, go/bp/main.go:156-159; measured through bn256_random_bases = k_i * G0, the same kernel with
device-generated scalars) and the generator fold s_lo*lo[i] + s_hi*hi[i] (go/bp/main.go:355-356).
BN256_SCALAR_MUL_V1=1 / BN256_FOLD_XYZZ=1 select round 2's kernels (A/B on one box)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import crypto_amd  # noqa: E402

eng = crypto_amd.Engine(0)
N = 0x8fb501e34aa387f9aa6fecb86184dc212e8d8e12f82b39241a2ef45b57ac7261
s_lo, s_hi = 0x1234567 << 200 | 99, N - 0xabcdef123
eng.random_bases(1, 64).free()
for k in (int(a) for a in (sys.argv[1:] or ["16", "18", "20"])):
    n = 1 << k
    best = 1e9
    for r in range(4):
        t0 = time.perf_counter()
        g = eng.random_bases(0x5EED0001, n)
        best = min(best, time.perf_counter() - t0)
        if r < 3:
            g.free()
    bf = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        r = eng.fold(g, 0, g, n // 2, n // 2, s_lo, s_hi)
        bf = min(bf, time.perf_counter() - t0)
        r.free()
    print(f"n=2^{k}: scalar_mul_each {best * 1e3:8.3f} ms ({n / best / 1e6:6.2f} M/s)   fold to 2^{k-1} {bf * 1e3:8.3f} ms "
          f"({n / 2 / bf / 1e6:6.2f} M outputs/s)", flush=True)
    g.free()
eng.close()
