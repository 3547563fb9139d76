#!/usr/bin/env python3
"""vecCom at full size through the call a Go shim would make (go/bp/util.go:96-98: <G, a> + <H, b>): one
bn256_msm_multi over two fixed-base tables with the scalars in HOST memory (32 MiB each at 2^20)."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import crypto_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=20)
args = ap.parse_args()
n = 1 << args.log2n
eng = crypto_amd.Engine(0)
G = eng.random_bases(0x61, n, flags=crypto_amd.PRECOMPUTE)
H = eng.random_bases(0x62, n, flags=crypto_amd.PRECOMPUTE)
va, vb = eng.fr_random(0x63, n), eng.fr_random(0x64, n)
a = b"".join(x.to_bytes(32, "big") for x in va.download())
b = b"".join(x.to_bytes(32, "big") for x in vb.download())
want = eng.msm_fr_multi([(G, 0, 1, n, va, 0, 1), (H, 0, 1, n, vb, 0, 1)])
ts = []
for i in range(12):
    t0 = time.perf_counter()
    got = eng.msm_multi([(G, 0, a), (H, 0, b)])
    ts.append((time.perf_counter() - t0) * 1e3)
    assert got == want
tr = []
for i in range(12):
    t0 = time.perf_counter()
    eng.msm_fr_multi([(G, 0, 1, n, va, 0, 1), (H, 0, 1, n, vb, 0, 1)])
    tr.append((time.perf_counter() - t0) * 1e3)
print(f"vecCom 2 x 2^{args.log2n}: host scalars {statistics.median(ts[2:]):.3f} ms, resident scalars {statistics.median(tr[2:]):.3f} ms")
