#!/usr/bin/env python3
"""One line of prover-flow timings for A/B runs (BN256MSM_LIB picks the library): WNLA 2^20 (median of 7),
go/bp N = 64 prove / verify (median of 9), the BP++ commitment (2^21 terms, blocking)."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import crypto_amd  # noqa: E402
from crypto_amd.bp import WNLA, BulletProofs  # noqa: E402

eng = crypto_amd.Engine(0)
w = WNLA(eng)
w.synthetic(0xABCC, 1 << 20, 1 << 19)
wn = statistics.median(w.synthetic(0xABCD + i, 1 << 20, 1 << 19)["ms_run"] for i in range(7))
bp = BulletProofs(eng)
bp.prove(64, 12345, 1)
pv = []
for seed in range(2, 11):
    t0 = time.perf_counter()
    pr = bp.prove(64, 987654321, seed)
    t1 = time.perf_counter()
    assert bp.verify(64, seed, pr) is None
    t2 = time.perf_counter()
    pv.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
print(f"wnla_2p20 {wn:.2f} ms  bp_n64 prove {statistics.median(p for p, _ in pv):.3f} verify {statistics.median(v for _, v in pv):.3f} ms")
eng.close()
