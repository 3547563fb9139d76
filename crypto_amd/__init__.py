"""crypto_amd: MI355X-native MSM engine for the BN256-G1 vector Pedersen commitments of
olegfomenko/crypto's go/bp and go/bppp provers.  The compute path is the HIP library
libbn256msm.so (crypto_amd/csrc, C ABI in include/bn256msm.h); this package is the thin
Python host side used by tests and bench.py."""
from .engine import ASYNC_LANES, BN256Error, Bases, Engine, FrVec, PRECOMPUTE  # noqa: F401
