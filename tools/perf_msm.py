#!/usr/bin/env python3
"""Phase-level timing sweep of the MSM pipeline on one GPU (tuning aid, not the bench)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import crypto_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--configs", default="16:0", help="comma list of window_bits:chunk")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--no-precompute", action="store_true")
args = ap.parse_args()

eng = crypto_amd.Engine(0)
eng.set_phase_timing(True)  # last_timing() below
n = 1 << args.log2n
if os.environ.get("PERF_PREALLOC"):
    dummy = eng.random_bases(0x5EED0001, n, flags=crypto_amd.PRECOMPUTE)
scal = torch.empty(n * 32, dtype=torch.uint8, device="cuda:0")
eng.random_scalars_into(0x5EED0002, n, scal.data_ptr())
names = ["recode", "scan", "scatter", "accum", "fix", "reduce", "total"]
ref = None
for cfg in args.configs.split(","):
    c, k = (int(x) for x in cfg.split(":"))
    eng.set_tuning(c, k)
    bases = eng.random_bases(0x5EED0001, n, flags=0 if args.no_precompute else crypto_amd.PRECOMPUTE)
    best = None
    for _ in range(args.reps):
        r = eng.msm_dev(bases, scal.data_ptr(), n)
        t = eng.last_timing()
        if os.environ.get("PERF_ALL"):
            print("   rep", " ".join(f"{v:7.3f}" for v in t), flush=True)
        if best is None or t[6] < best[6]:
            best = t
    if ref is None:
        ref = r
    assert r == ref, "result changed with tuning!"
    print(f"c={c:2d} K={k:4d} " + " ".join(f"{nm}={v:7.3f}" for nm, v in zip(names, best)) +
          f"  -> {n / best[6] / 1e3:8.1f} Mpts/s", flush=True)
    bases.free()
eng.close()
