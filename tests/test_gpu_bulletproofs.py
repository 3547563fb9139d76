"""BASELINE.json configs[0]: go/bp 64-bit range proof prove + verify with every vector
commitment going through the C ABI (plumbing), and the north star's "identical proof
transcripts": with the reader injected, the transcript produced by the C++ host mirror over
the HIP library equals the oracle's byte for byte (tests/golden/bn256_g1.json)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bp(eng):
    from crypto_amd.bp import BulletProofs
    return BulletProofs(eng)


@pytest.mark.parametrize("idx", [0, 1, 2], ids=["n4", "n8", "n64"])
def test_transcript_identical_to_oracle(bp, golden, idx):
    from crypto_amd.bp import parse_proof
    t = golden["bulletproofs"][idx]
    raw = bp.prove(t["n"], t["v"], int(t["seed"], 16))
    got = parse_proof(raw)
    want = t["proof"]
    for k in ("V", "A", "S", "T1", "T2"):
        assert got[k] == want[k], k
    for k in ("tx", "taux", "nu", "a", "b"):
        assert got[k] == int(want[k], 16), k
    assert got["L"] == want["L"] and got["R"] == want["R"]
    assert len(got["L"]) == t["n"].bit_length() - 1
    assert bp.verify(t["n"], int(t["seed"], 16), raw) is None


@pytest.mark.parametrize("idx", [1, 2], ids=["n8", "n64"])
def test_transcript_identical_with_the_tensor_expansion_on_the_device(bp, golden, idx, monkeypatch):
    """The inner-product argument over frozen generators makes its expanded scalar vectors on the host up to 512
    generators and on the device above (bn256_fr_tensor_step_multi / _expand_multi); BP_IPA_HOST_EXPAND=0 sends
    the fixtures' small proofs down the device's way too: the same bytes, and either verifier accepts either."""
    from crypto_amd.bp import parse_proof
    t = golden["bulletproofs"][idx]
    seed = int(t["seed"], 16)
    raw_host = bp.prove(t["n"], t["v"], seed)
    monkeypatch.setenv("BP_IPA_HOST_EXPAND", "0")
    raw_dev = bp.prove(t["n"], t["v"], seed)
    assert raw_dev == raw_host
    assert parse_proof(raw_dev)["L"] == t["proof"]["L"]
    assert bp.verify(t["n"], seed, raw_host) is None
    monkeypatch.delenv("BP_IPA_HOST_EXPAND")
    assert bp.verify(t["n"], seed, raw_dev) is None


def test_verify_rejects_with_reference_errors(bp, golden):
    t = golden["bulletproofs"][1]
    seed = int(t["seed"], 16)
    raw = bytearray(bp.prove(t["n"], t["v"], seed))
    # tx is the first scalar after the five points: tx + 1 -> first check fails (go/bp/main.go:220-222)
    bad = bytearray(raw)
    tx = int.from_bytes(bad[320:352], "big") + 1
    bad[320:352] = tx.to_bytes(32, "big")
    assert bp.verify(t["n"], seed, bytes(bad)) == "failed: tx ?= t0 + t1*x +t2*x^2"
    # a is the second-to-last scalar: inner argument fails (go/bp/main.go:278-280)
    bad = bytearray(raw)
    a = int.from_bytes(bad[-64:-32], "big") + 1
    bad[-64:-32] = a.to_bytes(32, "big")
    assert bp.verify(t["n"], seed, bytes(bad)) == "failed to verify inner argument proof"
    # a different statement (other public parameters) does not verify either
    assert bp.verify(t["n"], seed + 1, bytes(raw)) is not None
    # truncated / off-curve encodings are rejected before any arithmetic
    assert bp.verify(t["n"], seed, bytes(raw[:-1])) == "malformed proof encoding"


def test_range_proof_64_bit_random_values(bp):
    """go/bp/main_test.go:16-36 (TestBulletProof, n = 64, v = 11) plus boundary values."""
    for v, seed in [(11, 0xC0FFEE), (0, 1), ((1 << 64) - 1, 2), (0x8000000000000001, 3)]:
        raw = bp.prove(64, v, seed)
        assert bp.verify(64, seed, raw) is None


def test_range_proof_other_bit_lengths(bp):
    """Prove -> Verify at n = 2, 16, 128 and 256 bits (one and eight inner-product levels; the reference's
    own check is this round trip, go/bp/main_test.go:30-35): accepted, and a value outside the range or
    a proof for another parameter set is rejected."""
    for n, v, seed in [(2, 3, 11), (16, 0xBEEF, 12), (128, (1 << 128) - 5, 13), (256, (1 << 255) + 12345, 14)]:
        raw = bp.prove(n, v, seed)
        assert bp.verify(n, seed, raw) is None, n
        assert bp.verify(n, seed + 1, raw) is not None, n  # other generators
    assert bp.verify(16, 21, bp.prove(16, 1 << 16, 21)) is not None  # 2^16 does not fit 16 bits


def test_inner_product_argument(bp):
    """go/bp/docs_test.go:381-392 inputs (n = 4); the reference's TestInnerProduct (n = 8 with
    length-4 vectors) panics with "invalid length" there, see SURVEY.md section 4."""
    a, b = [4, 5, 10, 1], [2, 1, 2, 10]
    assert bp.inner_product_roundtrip(4, 0x1AA, a, b) is None
    assert bp.inner_product_roundtrip(4, 0x1AA, a, b, tamper=True) == "failed to verify inner argument proof"
    assert bp.inner_product_roundtrip(1, 7, [3], [9]) is None
    assert bp.inner_product_roundtrip(6, 7, list(range(6)), list(range(6))) == "invalid n: should be 2^x"
    import random
    rnd = random.Random(5)
    from cases import N_ORDER
    a = [rnd.randrange(N_ORDER) for _ in range(32)]
    b = [rnd.randrange(N_ORDER) for _ in range(32)]
    assert bp.inner_product_roundtrip(32, 99, a, b) is None


# ------------------------------------------------------------------ go/bppp: WNLA
@pytest.fixture(scope="module")
def wnla(eng):
    from crypto_amd.bp import WNLA
    return WNLA(eng)


@pytest.mark.parametrize("l,n", [
    ([4, 5, 10, 1, 99, 35, 1, 15], [1, 3, 42, 14]),          # TestWNLA, go/bppp/docs_test.go:1257-1258
    ([4, 5, 10, 1, 99, 35, 1], [1, 3, 42, 14, 9]),           # odd lengths: pad / truncate paths
    ([7, 8], [1, 2, 3]),                                     # opened immediately (len(l)+len(n) < 6)
    (list(range(1, 34)), list(range(100, 117))),             # 33 / 17
])
def test_wnla_matches_oracle(wnla, l, n):
    import bn256_ref as ref
    seed = 0x77AA + len(l)
    want = ref.wnla_setup_and_run(seed, l, n)
    got = wnla.run(seed, l, n)
    assert got["ok"] and want["ok"]
    assert got["C"] == ref.marshal(want["C"]).hex()
    assert got["rounds"] == [(ref.marshal(x).hex(), ref.marshal(r).hex()) for x, r in want["rounds"]]
    assert got["l"] == want["l"] and got["n"] == want["n"]


def test_wnla_2_16_commitment_vectors(wnla):
    """Scaled-down BASELINE.json configs[3]: every X / R commitment and generator fold on the
    device; the argument's own final check is the criterion (size-independent property)."""
    r = wnla.synthetic(0xABCD, 1 << 16, 1 << 15)
    assert r["ok"] and r["rounds"] >= 14


def test_wnla_2_20_commitment_vectors(wnla):
    """BASELINE.json configs[3] at its full size (|l| = 2^20, |n| = 2^19): 19 rounds with the X / R
    MSMs on two lanes, the lane-per-output fold kernel in the first rounds and the quad kernels after;
    the argument's final check (go/bppp/docs_test.go:1273-1288) must close, twice in one context."""
    for seed in (0xABCD, 0xABCE):
        r = wnla.synthetic(seed, 1 << 20, 1 << 19)
        assert r["ok"] and r["rounds"] == 19


def test_strided_msm_and_ragged_fold(eng, orc):
    n = 37
    braw = orc.synth_bases(0x51, n)
    sraw = orc.synth_scalars(0x52, n)
    sc = [int.from_bytes(sraw[32 * i:32 * i + 32], "big") for i in range(n)]
    h = eng.upload(braw)
    ev = b"".join(braw[64 * i:64 * i + 64] for i in range(0, n, 2))
    od = b"".join(braw[64 * i:64 * i + 64] for i in range(1, n, 2))
    ne, no = (n + 1) // 2, n // 2
    got = eng.msm_multi([(h, 0, sc[:ne], 2), (h, 1, sc[ne:ne + no], 2)])
    want = orc.add(orc.msm(ev, sraw[:32 * ne]), orc.msm(od, sraw[32 * ne:32 * (ne + no)]))
    assert got == want
    import crypto_amd
    with pytest.raises(crypto_amd.BN256Error):
        eng.msm_multi([(h, 1, sc[:no + 1], 2)])              # one past the end of the odd view
    out = eng.fold2(h, 0, ne, h, 1, no, 3, 5, stride=2).download()
    assert len(out) == ne
    for i in range(ne):
        w = orc.mul(braw[128 * i:128 * i + 64], 3)
        if i < no:
            w = orc.add(w, orc.mul(braw[128 * i + 64:128 * i + 128], 5))
        assert out[i] == w


# ------------------------------------------------------------------ go/bppp: arithmetic-circuit protocol
@pytest.fixture(scope="module")
def ac(eng):
    from crypto_amd.bp import ArithmeticCircuit
    return ArithmeticCircuit(eng)


def _ac_expected(tr):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_ac_golden
    want = make_ac_golden.transcript_json(tr)
    for k in ("vT",):
        want[k] = int(want[k], 16)
    for k in ("lT", "nT", "cT", "l", "n"):
        want[k] = [int(x, 16) for x in want[k]]
    want["rounds"] = [tuple(x) for x in want["rounds"]]
    return want


@pytest.mark.parametrize("idx", range(4))
def test_ac_protocol_transcript_identical_to_fixture(ac, idx):
    """ArithmeticCircuitProtocol (go/bppp/docs_test.go:496-1240) over the device library on the
    reference's own test circuits (TestACProtocol, TestArithmeticCircuit2, TestAC3, TestAC4): every
    commitment (V, Co, Cl, Cr, V_, Cs, PT, C(t)), the opened l(T), n(T), c(T), the three bytes.Equal
    checks and the whole WNLA transcript equal the committed oracle transcript byte for byte."""
    import json
    import os
    import ac_cases
    with open(os.path.join(os.path.dirname(__file__), "golden", "bppp_ac.json")) as f:
        gold = json.load(f)
    name, pre, post, pub, prv = ac_cases.CASES[idx]()
    got = ac.run(gold["seed"], pre, post, pub, prv)
    want = dict(gold["cases"][name])
    want["vT"] = int(want["vT"], 16)
    for k in ("lT", "nT", "cT", "l", "n"):
        want[k] = [int(x, 16) for x in want[k]]
    want["rounds"] = [tuple(x) for x in want["rounds"]]
    assert got == want
    assert got["check_V"] and got["check_CT"] and got["check_45"] and got["wnla_ok"]


def test_ac_protocol_wide_circuit_matches_oracle(ac):
    """The TestAC3 construction with 96 digits (Nv = 208, HVec of 217 points, 7 WNLA rounds): the
    commitments are real multi-segment MSMs; compared with the oracle run live."""
    import bn256_ref as ref
    import ac_cases
    case = ac_cases.wide_lookup()
    name, pre, post, pub, prv = case
    pubg, prvg, rd = ac_cases.with_generators(case, 0xAC77)
    want = _ac_expected(ref.ac_protocol(pubg, prvg, rd))
    got = ac.run(0xAC77, pre, post, pub, prv)
    assert got == want and got["wnla_ok"]


def test_ac_protocol_wrong_witness_fails_like_the_oracle(ac):
    import bn256_ref as ref
    import ac_cases
    case = ac_cases.test_ac_protocol()
    name, pre, post, pub, prv = case
    prv = dict(prv)
    prv["wo"] = [16]
    bad = (name, pre, post, pub, prv)
    pubg, prvg, rd = ac_cases.with_generators(bad, 0xAC00)
    want = ref.ac_protocol(pubg, prvg, rd)
    got = ac.run(0xAC00, pre, post, pub, prv)
    assert got["check_CT"] == want["check_CT"] and got["wnla_ok"] == want["wnla_ok"] and not got["wnla_ok"]
    assert got["CT"] == ref.marshal(want["CT"]).hex() and got["Co"] == ref.marshal(want["Co"]).hex()


def test_product_entry_points_os_randomness_round_trip(eng):
    """bp_setup / bp_prove / bp_verify: generators and blinding from the OS CSPRNG as the reference draws
    them from crypto/rand (go/bp/util.go:51-63,81-90), nothing from a seed.  Proofs are randomised, so the
    bar is prove -> verify: by the library, and by the oracle's restatement of Verify
    (go/bp/main.go:181-249) over the downloaded generators."""
    import secrets
    import bn256_ref as ref
    from crypto_amd import bp
    n, v = 64, 11                                          # go/bp/main_test.go:16-36
    pub = bp.BulletProofPublic(eng, n)
    pub2 = bp.BulletProofPublic(eng, n)
    assert pub.G.download() != pub2.G.download() and pub.U != pub2.U     # fresh generators every time
    assert len(set(pub.G.download() + pub.H.download() + [pub.U, pub.g, pub.h])) == 2 * n + 3
    prv = secrets.randbelow(ref.N)
    p1, p2 = pub.prove(v, prv), pub.prove(v, prv)
    assert p1 != p2 and p1[:64] == p2[:64]                 # fresh blinding; V = com(v, prv) is the same
    assert pub.verify(p1) is None and pub.verify(p2) is None
    assert pub2.verify(p1) is not None                     # other generators
    bad = bytearray(p1)
    bad[64 * 5 + 31] ^= 1                                  # tx
    assert pub.verify(bytes(bad)) == "failed: tx ?= t0 + t1*x +t2*x^2"
    # the oracle's Verify over the same public parameters
    k = int.from_bytes(p1[64 * 5 + 96:64 * 5 + 100], "big")
    pts = lambda off, m: [ref.unmarshal(p1[off + 64 * i:off + 64 * i + 64]) for i in range(m)]  # noqa: E731
    sc = lambda off: int.from_bytes(p1[off:off + 32], "big")  # noqa: E731
    V, A, S, T1, T2 = pts(0, 5)
    o = 320
    pr = {"V": V, "A": A, "S": S, "T1": T1, "T2": T2, "tx": sc(o), "taux": sc(o + 32), "nu": sc(o + 64),
          "L": pts(o + 100, k), "R": pts(o + 100 + 64 * k, k), "a": sc(o + 100 + 128 * k), "b": sc(o + 132 + 128 * k)}
    inner = ref.InnerArgumentPublic(n, [ref.unmarshal(p) for p in pub.G.download()],
                                    [ref.unmarshal(p) for p in pub.H.download()], ref.unmarshal(pub.U))
    opub = ref.BulletProofPublic(inner, n, ref.unmarshal(pub.g), ref.unmarshal(pub.h))
    assert ref.bp_verify(opub, pr) is None
    # a value outside the range does not verify (v = 2^64 needs 65 bits)
    assert pub.verify(pub.prove(1 << 64, prv)) is not None
    # the caller's own generators (uploaded handles) instead of bp_setup's
    pub3 = bp.BulletProofPublic(eng, n, G=pub.G, H=pub.H, U=pub.U, g=pub.g, h=pub.h)
    assert pub3.verify(p1) is None
    with pytest.raises(RuntimeError, match="invalid length"):
        bp.BulletProofPublic(eng, 32, G=pub.G, H=pub.H, U=pub.U, g=pub.g, h=pub.h).prove(v, prv)
