#!/usr/bin/env python3
"""BASELINE.json configs[3]: WNLA (go/bppp/docs_test.go:1269-1338) over 2^k-length commitment
vectors, all X / R commitments and generator folds on the device."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import crypto_amd  # noqa: E402
from crypto_amd.bp import WNLA  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = crypto_amd.Engine(0)
w = WNLA(eng)
w.synthetic(0xABCC, 1 << k, 1 << (k - 1))  # warm-up: lanes, workspaces, code objects (as tools/bp_bench.py)
r = w.synthetic(0xABCD, 1 << k, 1 << (k - 1))
r["config"] = f"WNLA |l| = 2^{k}, |n| = 2^{k-1}, synthetic generators"
print(json.dumps(r))
eng.close()
