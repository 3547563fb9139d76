#!/usr/bin/env python3
"""bn256_msm with the scalars in host memory (pageable / pinned), 2^20 and 2^22 points over a fixed-base table:
median wall ms of the whole call (SURVEY 8d's wording of the metric), result checked against the closed form."""
import os
import statistics
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import crypto_amd  # noqa: E402
from oracle_c import OracleC  # noqa: E402

orc = OracleC()
eng = crypto_amd.Engine(0)
for lg in (20, 22):
    n = 1 << lg
    h = eng.random_bases(0x5EED0001, n, flags=crypto_amd.PRECOMPUTE)
    d = torch.empty(n * 32, dtype=torch.uint8, device="cuda:0")
    eng.random_scalars_into(0x5EED0002, n, d.data_ptr())
    host = d.cpu()
    pinned = host.pin_memory()
    want = orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, n)
    for name, ptr in (("pageable", host.numpy().ctypes.data), ("pinned", pinned.data_ptr())):
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            r = eng.msm_host_ptr(h, ptr, n)
            ts.append(time.perf_counter() - t0)
        print(f"2^{lg} {name:8s} median {statistics.median(ts[2:]) * 1e3:.3f} ms  {'ok' if r == want else 'MISMATCH'}", flush=True)
    h.free()
eng.close()
