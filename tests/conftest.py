import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "bn256_g1.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    from oracle_c import OracleC
    return OracleC()


@pytest.fixture(scope="session")
def eng():
    import crypto_amd
    e = crypto_amd.Engine(0)
    yield e
    e.close()
