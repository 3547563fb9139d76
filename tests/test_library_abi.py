"""CPU tests of the C-ABI boundary: the library loads, exports every symbol that
include/bn256msm.h declares, and its host-only entry points work without a GPU."""
import ctypes as C
import os
import re

import bn256_ref as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "bn256msm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bn256_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from crypto_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names


def test_strerror_and_null_arguments():
    from crypto_amd import _lib
    lib = _lib.load()
    assert lib.bn256_strerror(0) == b"ok"
    assert lib.bn256_strerror(-2) == b"invalid length for scalar mul"     # go/bp/util.go:167
    assert lib.bn256_strerror(-4) == b"bn256: malformed point"
    out = C.create_string_buffer(64)
    assert lib.bn256_msm(None, None, 0, 0, None, out) == -1
    assert lib.bn256_ctx_create(0, None) == -1
    assert lib.bn256_last_timing(None, None, 0) == 0
    lib.bn256_ctx_destroy(None)
    lib.bn256_bases_free(None, None)
    assert lib.bn256_bases_len(None) == 0
    # Fr vectors / batched commitments: argument checks come before any HIP call
    h = C.c_void_p()
    assert lib.bn256_fr_upload(None, None, 0, C.byref(h)) == -1
    assert lib.bn256_fr_random(None, 1, 0, 4, C.byref(h)) == -1
    assert lib.bn256_fr_len(None) == 0 and not lib.bn256_fr_device_ptr(None)
    lib.bn256_fr_free(None, None)
    o32 = C.create_string_buffer(32)
    assert lib.bn256_fr_dot(None, None, 0, 1, 0, None, 0, 1, 0, None, o32) == -1
    assert lib.bn256_fr_fold(None, None, 0, 0, None, 0, 0, 1, o32, o32, C.byref(h)) == -1
    assert lib.bn256_msm_fr(None, None, 0, 1, 0, None, 0, 1, out) == -1
    assert lib.bn256_msm_fr_multi(None, 0, None, None, None, None, None, None, None, None, out) == -1
    assert lib.bn256_com_batch(None, out, out, None, None, 0, C.byref(h)) == -1


def test_host_g1_sum_matches_oracle():
    """bn256_g1_sum is pure host code (combination of per-GPU partial sums)."""
    from crypto_amd import _lib
    lib = _lib.load()
    out = C.create_string_buffer(64)
    ks = [5, 7, ref.N - 12, 0, 12345678901234567890]
    pts = [ref.marshal(ref.mul(k, ref.G0)) for k in ks]
    assert lib.bn256_g1_sum(b"".join(pts), len(pts), out) == 0
    assert out.raw == ref.marshal(ref.mul(sum(ks) % ref.N, ref.G0))
    assert lib.bn256_g1_sum(pts[0] + pts[0], 2, out) == 0 and out.raw == ref.marshal(ref.mul(10, ref.G0))
    assert lib.bn256_g1_sum(pts[0] + ref.marshal(ref.neg(ref.mul(5, ref.G0))), 2, out) == 0 and out.raw == bytes(64)
    assert lib.bn256_g1_sum(b"", 0, out) == 0 and out.raw == bytes(64)
    bad = (1).to_bytes(32, "big") + (3).to_bytes(32, "big")
    assert lib.bn256_g1_sum(bad, 1, out) == -4
    assert lib.bn256_g1_sum(ref.P.to_bytes(32, "big") + bytes(32), 1, out) == -3


def test_host_normalisation_over_many_points():
    """Every result leaves the library through hostg1::marshal, whose one inversion is a binary extended
    Euclid (host_g1.h: inv): 150 random two-point sums against the oracle's affine bytes."""
    import random
    from crypto_amd import _lib
    lib = _lib.load()
    rng = random.Random(0x1A7)
    out = C.create_string_buffer(64)
    for _ in range(150):
        k1, k2 = rng.randrange(1, ref.N), rng.randrange(1, ref.N)
        pts = ref.marshal(ref.mul(k1, ref.G0)) + ref.marshal(ref.mul(k2, ref.G0))
        assert lib.bn256_g1_sum(pts, 2, out) == 0
        assert out.raw == ref.marshal(ref.mul((k1 + k2) % ref.N, ref.G0))


def test_glv_split_host_arithmetic():
    """bn256_glv_split (pure host code behind the generator folds): k = k1 + k2*lambda mod n with
    |k1|, |k2| < 2^129, and lambda*(x, y) = (beta*x, y) on the oracle's curve."""
    import ctypes as C
    import random
    from crypto_amd import _lib
    lib = _lib.load()
    lam = 0x196ac037f07e9f9eecf9176e4f8bfcd513be518b5264ac23d
    beta = 0xcb5601bf83f4fcf8877db38fb92fa161b9ff3815a6c2a92e
    assert (lam * lam + lam + 1) % ref.N == 0 and pow(beta, 3, ref.P) == 1 and beta != 1
    q = ref.mul(0xC0FFEE, ref.G0)
    assert ref.mul(lam, q) == (beta * q[0] % ref.P, q[1])
    rng = random.Random(7)
    ks = [0, 1, 2, lam, lam - 1, ref.N - 1, ref.N, ref.N + 5, 2**256 - 1, 2**255, 2**128, 2**129 - 1]
    ks += [rng.randrange(2**256) for _ in range(2000)] + [rng.randrange(2**k) for k in range(1, 256, 3)]
    k1b, k2b = C.create_string_buffer(32), C.create_string_buffer(32)
    n1, n2 = C.c_int(), C.c_int()
    for k in ks:
        assert lib.bn256_glv_split(k.to_bytes(32, "big"), k1b, C.byref(n1), k2b, C.byref(n2)) == 0
        k1 = int.from_bytes(k1b.raw, "big") * (-1 if n1.value else 1)
        k2 = int.from_bytes(k2b.raw, "big") * (-1 if n2.value else 1)
        assert abs(k1) < 2**129 and abs(k2) < 2**129, hex(k)
        assert (k1 + k2 * lam - k) % ref.N == 0, hex(k)
    assert lib.bn256_glv_split(None, k1b, C.byref(n1), k2b, C.byref(n2)) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from crypto_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_bp_library_exports_and_keccak(golden):
    """libbpamd.so (include/bp_amd.h) loads, exports every declared symbol; Keccak is host-only."""
    import ctypes as C
    from crypto_amd import bp
    lib = bp.load()
    txt = open(os.path.join(ROOT, "include", "bp_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = sorted(set(re.findall(r"\b(bp_[a-z0-9_]+)\s*\(", txt)))
    assert names == sorted(bp.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    out = C.create_string_buffer(32)
    for e in golden["keccak256"]:
        m = bytes.fromhex(e["msg"])
        lib.bp_keccak256(m, len(m), out)
        assert out.raw.hex() == e["digest"]
    assert lib.bp_prove_det(None, 4, None, 0, None, 0, None) == -1
