#!/usr/bin/env python3
"""Runs the randomised differential tests (tests/test_gpu_random_shapes.py, test_gpu_random_elementwise.py) on
seeds beyond the ones pytest collects:   python tools/soak_random.py [first_seed [count]]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_c import OracleC  # noqa: E402
import test_gpu_random_elementwise as te  # noqa: E402
import test_gpu_random_shapes as ts  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
orc = OracleC()
for seed in range(first, first + count):
    ts.test_random_calls_against_the_closed_form.__wrapped__(orc, seed) if hasattr(ts.test_random_calls_against_the_closed_form, "__wrapped__") else ts.test_random_calls_against_the_closed_form(orc, seed)
    if seed % 4 == 0:
        ts.test_random_calls_on_a_multi_device_context(orc, seed)
        te.test_random_elementwise_calls(orc, seed)
    print("seed", seed, "ok", flush=True)
