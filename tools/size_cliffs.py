#!/usr/bin/env python3
"""Phase times of blocking fixed-base calls at sizes on both sides of the pipeline's switches (tile and pass
boundaries, window-width change at 2^20): ns per point per phase, to spot cliffs such as the padded-tile
guess of k_group_sort_tl.   python tools/size_cliffs.py"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import crypto_amd  # noqa: E402

eng = crypto_amd.Engine(0)
NMAX = (1 << 22) + 4096
v = eng.fr_random(0x41, NMAX)
for table_n in (1 << 19, NMAX):
    H = eng.random_bases(0xC11F, table_n, flags=crypto_amd.PRECOMPUTE)
    sizes = ((1 << 18) - 1, 1 << 18, (1 << 18) + 1, 300000, 1 << 19) if table_n == 1 << 19 else (
        (1 << 19) + 1, 800000, (1 << 20) - 2048, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, (1 << 20) + 2049, (1 << 20) + 70000,
        1150000, 1179648, 1179649, 1300000, 1 << 21, (1 << 21) + 1, 2359296, 2359297, 3000000, 1 << 22, (1 << 22) + 4096)
    for n in sizes:
        eng.set_phase_timing(False)
        eng.msm_dev(H, v.device_ptr(), n)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            eng.msm_dev(H, v.device_ptr(), n)
            ts.append((time.perf_counter() - t0) * 1e3)
        eng.set_phase_timing(True)
        eng.msm_dev(H, v.device_ptr(), n)
        eng.msm_dev(H, v.device_ptr(), n)
        ph = eng.last_timing()
        print(f"table {table_n:8d} n {n:8d}: {statistics.median(ts):7.3f} ms = {statistics.median(ts) * 1e6 / n:6.3f} ns/pt   phases ns/pt "
              + " ".join(f"{x * 1e6 / n:6.3f}" for x in ph[:6]), flush=True)
    H.free()
