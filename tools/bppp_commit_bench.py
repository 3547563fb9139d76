#!/usr/bin/env python3
"""configs[3] timing: the four BP++ vector commitments Co, Cl, Cr, Cs (go/bppp/docs_test.go:560-564,
588-589,1002-1003) at Nv = Nm = 2^log2n, each <HVec, r||l> + <GVec, n> as one bn256_msm_fr_multi over
two fixed-base tables with resident scalars; blocking, and overlapped on the library's lanes."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import crypto_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
nv = nm = 1 << args.log2n
eng = crypto_amd.Engine(0)
t0 = time.perf_counter()
H = eng.random_bases(0xB9901, 9 + nv, flags=crypto_amd.PRECOMPUTE)
G = eng.random_bases(0xB9902, nm, flags=crypto_amd.PRECOMPUTE)
print(f"set-up of both generator tables: {(time.perf_counter() - t0) * 1e3:.1f} ms")
vec = [(eng.fr_random(0x21 + 2 * i, 9 + nv), eng.fr_random(0x22 + 2 * i, nm)) for i in range(4)]
segs = lambda i: [(H, 0, 1, 9 + nv, vec[i][0], 0, 1), (G, 0, 1, nm, vec[i][1], 0, 1)]  # noqa: E731


def blocking():
    return [eng.msm_fr_multi(segs(i)) for i in range(4)]


def overlapped():
    t = [eng.msm_fr_multi_submit(segs(i)) for i in range(3)]
    out = [eng.msm_wait(t[0])]
    t.append(eng.msm_fr_multi_submit(segs(3)))
    return out + [eng.msm_wait(x) for x in t[1:]]


for name, fn in (("blocking", blocking), ("overlapped", overlapped)):
    fn()
    fn()
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    med = statistics.median(ts)
    print(f"{name:10s}: Co, Cl, Cr, Cs = {med * 1e3:7.3f} ms (median of {args.reps}), {med / 4 * 1e3:6.3f} ms per commitment of "
          f"{9 + nv + nm} terms = {4 * (9 + nv + nm) / med / 1e6:7.1f} Mpts/s")
assert blocking() == overlapped()
