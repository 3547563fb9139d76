"""Builds libbn256msm.so (HIP, gfx950) in-tree.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbn256msm.so")
SOURCES = ["bn256msm.hip"]


def _headers() -> list:
    """Everything the translation unit can include: csrc/*.h, *.cuh, *.inc and the public header."""
    import glob
    hs = []
    for pat in ("*.h", "*.cuh", "*.inc"):
        hs += glob.glob(os.path.join(CSRC, pat))
    return hs + [os.path.join(HERE, "..", "include", "bn256msm.h")]



def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in [os.path.join(CSRC, s) for s in SOURCES] + _headers())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB] + [
        os.path.join(CSRC, s) for s in SOURCES
    ]
    if verbose:
        print("[crypto_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


BPLIB = os.path.join(HERE, "libbpamd.so")


def build_bp(force: bool = False, verbose: bool = True) -> str:
    """Host-side go/bp mirror (plain C++, links against libbn256msm.so next to it)."""
    src = os.path.join(CSRC, "bp_host.cpp")
    deps = [src, os.path.join(CSRC, "host_g1.h"), os.path.join(HERE, "..", "include", "bp_amd.h"), LIB]
    if not force and os.path.exists(BPLIB) and all(os.path.getmtime(d) <= os.path.getmtime(BPLIB) for d in deps):
        return BPLIB
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", BPLIB, src,
           "-L" + HERE, "-lbn256msm", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print("[crypto_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return BPLIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_bp(force="--force" in sys.argv)
