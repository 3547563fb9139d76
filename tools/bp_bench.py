#!/usr/bin/env python3
"""BASELINE.json configs[0]: go/bp range proof (N = 64 by default) prove + verify through the
device library; wall milliseconds, generator set-up included in prove (as in go/bp/main_test.go)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import crypto_amd  # noqa: E402
from crypto_amd.bp import BulletProofs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = crypto_amd.Engine(0)
bp = BulletProofs(eng)
bp.prove(n, 12345, 1)  # warm-up (workspaces, code objects)
res = []
for seed in range(2, 7):
    t0 = time.perf_counter()
    pr = bp.prove(n, 987654321 % (1 << min(n, 62)), seed)
    t1 = time.perf_counter()
    err = bp.verify(n, seed, pr)
    t2 = time.perf_counter()
    assert err is None, err
    res.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
print(json.dumps({"config": f"go/bp range proof N={n}", "prove_ms_min": min(r[0] for r in res),
                  "verify_ms_min": min(r[1] for r in res), "runs": len(res)}))
eng.close()
