"""crypto_amd/csrc/inv29.h -- the modular inversion behind every XYZZ -> affine normalisation on the device
(fe_inv, fe29.cuh) -- is plain C++ that compiles for the host as it is: tests/inv29_host_test.cpp checks
x * inverse(x) == 1 against the library's host field arithmetic for edge values (0, small, 2^k and
neighbours, p - small) and random values of several shapes.  The device build of the same source is checked
by bn256_selftest (k_selftest: b * fe_inv(b) == 1 for 2^16 values) in the GPU tier."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_inverse_on_the_host(tmp_path):
    exe = tmp_path / "inv29_host_test"
    subprocess.run(["g++", "-O2", "-o", str(exe), os.path.join(ROOT, "tests", "inv29_host_test.cpp")], check=True)
    r = subprocess.run([str(exe), "300000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
