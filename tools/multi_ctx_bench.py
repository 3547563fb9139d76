#!/usr/bin/env python3
"""One context over several devices behind the C ABI (bn256_ctx_create_multi): the form of multi-GPU use
that the reference's single-goroutine caller can bind (go/bp/main.go:81-178).  One process, one caller
thread; the library shards the bases by contiguous slice, runs every shard's pipeline on its device's
host thread and adds the <= 8 partial sums on the host.  bench.py starts this as a child process after
its own legs (rank 0 only) and copies the JSON line into `multi_device_ctx`.

  python tools/multi_ctx_bench.py --devices 0,1,2,3 --total-log2n 24 [--expect <hex of the 64-byte sum>]

--devices 0,0 is the one-GPU rehearsal (two shards sharing a device): it exercises the path, its rate
says nothing about scaling."""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

SEED_BASES, SEED_SCALARS = 0x5EED0001, 0x5EED0002


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="0,0")
    ap.add_argument("--total-log2n", type=int, default=24)
    ap.add_argument("--calls", type=int, default=5)
    ap.add_argument("--expect", default="")
    args = ap.parse_args()
    devices = [int(x) for x in args.devices.split(",")]
    import crypto_amd
    n = 1 << args.total_log2n
    eng = crypto_amd.Engine(devices=devices)
    t0 = time.perf_counter()
    bases = eng.random_bases(SEED_BASES, n, flags=crypto_amd.PRECOMPUTE)
    setup_s = time.perf_counter() - t0
    v = eng.fr_random(SEED_SCALARS, n)  # on the context's first device; the call copies each shard its slice
    res = None
    for _ in range(2):
        res = eng.msm_dev(bases, v.device_ptr(), n)
    ts = []
    for _ in range(args.calls):
        t0 = time.perf_counter()
        res = eng.msm_dev(bases, v.device_ptr(), n)
        ts.append(time.perf_counter() - t0)
    med = statistics.median(ts)
    out = {"devices": devices, "points_total": n, "calls": args.calls, "median_ms": med * 1e3,
           "points_per_s": n / med, "setup_seconds_bases": setup_s, "result_xy": res.hex(),
           "scalars": "resident on devices[0]; each call peer-copies the other shards their slices"}
    if args.expect:
        out["same_result_as_rank_path"] = bool(res.hex() == args.expect)
    if len(set(devices)) < len(devices):
        out["note"] = "rehearsal: several shards share one device"
    bases.free()
    v.free()
    eng.close()
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
