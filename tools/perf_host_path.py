#!/usr/bin/env python3
"""Secondary figures of SURVEY.md 8(d), not the headline:
  * PCIe-inclusive rate: bn256_msm with the scalars in host memory (upload inside the call);
  * "Bulletproofs-like" scalars (go/bp/main.go:84-85): half the vector in {0,1}, half in {0,n-1}.
Prints one line per figure."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import crypto_amd  # noqa: E402

N_ORDER = 0x8FB501E34AA387F9AA6FECB86184DC212E8D8E12F82B39241A2EF45B57AC7261

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
n = 1 << args.log2n

eng = crypto_amd.Engine(0)
eng.set_phase_timing(True)  # last_timing() below
bases = eng.random_bases(0x5EED0001, n, flags=crypto_amd.PRECOMPUTE)
dscal = torch.empty(n * 32, dtype=torch.uint8, device="cuda:0")
eng.random_scalars_into(0x5EED0002, n, dscal.data_ptr())
host = dscal.cpu().numpy().tobytes()


def timed(fn):
    fn()
    best, tot = 1e9, 0.0
    for _ in range(args.reps):
        t0 = time.perf_counter()
        r = fn()
        dt = time.perf_counter() - t0
        best, tot = min(best, dt), tot + dt
    return r, best, tot / args.reps


r_dev, b, m = timed(lambda: eng.msm_dev(bases, dscal.data_ptr(), n))
print(f"uniform, scalars resident in HBM : best {b*1e3:7.3f} ms  mean {m*1e3:7.3f} ms  {n/m/1e6:8.1f} Mpts/s")
r_host, b, m = timed(lambda: eng.msm(bases, host, 0, n))
assert r_host == r_dev
print(f"uniform, scalars from host (PCIe): best {b*1e3:7.3f} ms  mean {m*1e3:7.3f} ms  {n/m/1e6:8.1f} Mpts/s")

# Bulletproofs-like: aL bits in {0,1} over the first half, aR = aL - 1 in {0, n-1} over the second
rng = np.random.default_rng(7)
bits = rng.integers(0, 2, size=n // 2, dtype=np.uint8)
arr = np.zeros((n, 32), dtype=np.uint8)
arr[: n // 2, 31] = bits
nm1 = np.frombuffer((N_ORDER - 1).to_bytes(32, "big"), dtype=np.uint8)
arr[n // 2:][bits == 0] = nm1
dbp = torch.from_numpy(arr.reshape(-1)).to("cuda:0")
r_bp, b, m = timed(lambda: eng.msm_dev(bases, dbp.data_ptr(), n))
t = eng.last_timing()
print(f"bulletproofs-like, resident      : best {b*1e3:7.3f} ms  mean {m*1e3:7.3f} ms  {n/m/1e6:8.1f} Mpts/s"
      f"  (device phases {' '.join(f'{v:.3f}' for v in t)})")
eng.close()
