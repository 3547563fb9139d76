#!/usr/bin/env python3
"""Throughput of T host threads, each with its own context (own stream and workspaces) on the same
GPU, issuing 2^log2n-point fixed-base MSMs back to back: do the latency-bound phases of one call
(bucket reduction, host tail) hide under the accumulation of another?

  python tools/perf_pipelined.py [--log2n 20] [--threads 2] [--steps 40]"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_amd

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--threads", type=int, default=2)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--delay-us", type=float, default=0.0, help="host busy-wait after every wait (submit/wait part)")
a = ap.parse_args()
n = 1 << a.log2n
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctxs = []
for t in range(a.threads):
    eng = crypto_amd.Engine(0)
    eng.set_phase_timing(True)  # last_timing() below
    bases = eng.random_bases(0x5EED0001, n, flags=crypto_amd.PRECOMPUTE)
    sc = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    eng.random_scalars_into(0x5EED0002, n, sc.data_ptr())
    ctxs.append((eng, bases, sc))
ref = None
for eng, bases, sc in ctxs:
    for _ in range(3):
        r = eng.msm_dev(bases, sc.data_ptr(), n)
    assert ref is None or r == ref
    ref = r
torch.cuda.synchronize()


def run(eng, bases, sc, k, out):
    for _ in range(k):
        out.append(eng.msm_dev(bases, sc.data_ptr(), n))


for T in sorted({1, a.threads}):
    outs = [[] for _ in range(T)]
    th = [threading.Thread(target=run, args=(*ctxs[i], a.steps, outs[i])) for i in range(T)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(o == ref for lst in outs for o in lst)
    print(f"threads={T} msms={T * a.steps} wall={dt * 1e3:.1f} ms  {dt * 1e3 / (T * a.steps):.3f} ms/msm  "
          f"{n * T * a.steps / dt / 1e6:.1f} Mpts/s", flush=True)

# one host thread, bn256_msm_submit / bn256_msm_wait with both lanes in flight
eng, bases, sc = ctxs[0]
for depth in (1, 2):
    for rep in range(2):
        q, got = [], []
        t0 = time.perf_counter()
        for _ in range(a.steps):
            q.append(eng.msm_submit(bases, sc.data_ptr(), n))
            if len(q) == depth:
                got.append(eng.msm_wait(q.pop(0)))
                t_end = time.perf_counter() + a.delay_us * 1e-6
                while time.perf_counter() < t_end:
                    pass
        while q:
            got.append(eng.msm_wait(q.pop(0)))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert all(o == ref for o in got)
    print(f"submit/wait depth={depth} msms={a.steps} {dt * 1e3 / a.steps:.3f} ms/msm  {n * a.steps / dt / 1e6:.1f} Mpts/s  "
          f"last_timing={['%.3f' % x for x in eng.last_timing()]}", flush=True)
