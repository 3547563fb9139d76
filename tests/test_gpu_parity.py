"""GPU parity tests: the HIP path through the C ABI against the oracle and golden fixtures.
Bit-exact bar (integer / byte work).  Run on the GPU box: pytest -m gpu."""
import hashlib

import pytest

import crypto_amd

import bn256_ref as ref
from cases import CASES, N_ORDER
from oracle_c import build_case

pytestmark = pytest.mark.gpu

INF = bytes(64)


def test_field_selftest(eng):
    assert eng.selftest() == 0


def test_upload_download_roundtrip(eng, golden):
    pts = [bytes.fromhex(e["point"]) for e in golden["kg"]]
    h = eng.upload(pts)
    assert len(h) == len(pts)
    assert h.download() == pts          # includes infinity (k = 0, n)
    assert h.download(3, 2) == pts[3:5]
    h.free()


def test_upload_rejects_bad_points(eng):
    import crypto_amd
    g0 = ref.marshal(ref.G0)
    bad_curve = g0[:63] + b"\x00"
    with pytest.raises(crypto_amd.BN256Error) as e:
        eng.upload([g0, bad_curve])
    assert e.value.code == -4
    big = ref.P.to_bytes(32, "big") + (2).to_bytes(32, "big")
    with pytest.raises(crypto_amd.BN256Error) as e:
        eng.upload([big])
    assert e.value.code == -3


def test_random_bases_match_oracle(eng, orc, golden):
    n = 1024
    h = eng.random_bases(0x5EED0001, n)
    got = b"".join(h.download())
    assert got == orc.synth_bases(0x5EED0001, n)
    assert hashlib.sha256(got).hexdigest() == golden["synth"]["bases_sha256_1024"]
    h2 = eng.random_bases(0x5EED0001, 16, first=100)
    assert b"".join(h2.download()) == got[64 * 100:64 * 116]


def test_random_scalars_match_oracle(eng, orc, golden):
    import torch
    n = 1024
    buf = torch.empty(n * 32, dtype=torch.uint8, device="cuda:0")
    eng.random_scalars_into(0x5EED0002, n, buf.data_ptr())
    got = bytes(buf.cpu().numpy().tobytes())
    assert got == orc.synth_scalars(0x5EED0002, n)
    assert hashlib.sha256(got).hexdigest() == golden["synth"]["scalars_sha256_1024"]


@pytest.mark.parametrize("case", [c for c in CASES if c[1] <= 1024], ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["plain", "precomp", "c4", "c7_k3", "c13", "c18_fallback", "c18_precomp"])
def test_msm_cases(eng, orc, golden, case, mode):
    import crypto_amd
    want = bytes.fromhex(next(m for m in golden["msm"] if m["name"] == case[0])["result"])
    bases, scalars = build_case(orc, case)
    try:
        if mode == "c4":
            eng.set_tuning(4, 0)
        elif mode == "c7_k3":
            eng.set_tuning(7, 3)
        elif mode == "c13":
            eng.set_tuning(13, 5)
        elif mode.startswith("c18"):      # bucket array too large for LDS: global-atomic sort fallback
            eng.set_tuning(18, 0)
        h = eng.upload(bases, crypto_amd.PRECOMPUTE if "precomp" in mode else 0)
        got = eng.msm(h, scalars)
        h.free()
    finally:
        eng.set_tuning(0, 0)
    assert got == want


def test_msm_65536_fixture_and_oracle(eng, orc, golden):
    """BASELINE.json configs[1]: 2^16 random points, bit-exact vs the CPU restatement."""
    import crypto_amd
    case = next(c for c in CASES if c[0] == "n65536")
    want = bytes.fromhex(next(m for m in golden["msm"] if m["name"] == "n65536")["result"])
    n = case[1]
    braw = orc.synth_bases(case[2], n)
    sraw = orc.synth_scalars(case[3], n)
    assert orc.msm(braw, sraw, mt=True) == want            # C restatement == Python fixture
    h = eng.upload(braw)
    assert eng.msm(h, sraw) == want
    hp = eng.upload(braw, crypto_amd.PRECOMPUTE)
    assert eng.msm(hp, sraw) == want
    hd = eng.random_bases(case[2], n, flags=crypto_amd.PRECOMPUTE)   # device-generated bases
    assert eng.msm(hd, sraw) == want
    # sub-range and offsets
    k = 1000
    assert eng.msm(h, sraw[32 * k:32 * (k + 777)], off=k) == orc.msm(braw[64 * k:64 * (k + 777)], sraw[32 * k:32 * (k + 777)], mt=True)
    assert eng.msm(hp, sraw[32 * k:32 * (k + 777)], off=k) == orc.msm(braw[64 * k:64 * (k + 777)], sraw[32 * k:32 * (k + 777)], mt=True)


def test_msm_empty_and_length_errors(eng):
    import crypto_amd
    g = [ref.marshal(ref.mul(k, ref.G0)) for k in (1, 2, 3)]
    h = eng.upload(g)
    assert eng.msm(h, []) == INF                                   # BP++ empty case
    with pytest.raises(ValueError, match="invalid length for scalar mul"):
        eng.vector_point_scalar_mul(h, [1, 2])                     # go/bp panic
    assert eng.vector_point_scalar_mul_bppp(h, [5]) == ref.marshal(ref.mul(5, ref.G0))        # zero-padded
    assert eng.vector_point_scalar_mul_bppp(h, [1, 1, 1, 9]) == ref.marshal(ref.mul(6, ref.G0))  # truncated
    with pytest.raises(crypto_amd.BN256Error) as e:
        eng.msm(h, [1, 2, 3, 4])
    assert e.value.code == -2
    e0 = eng.upload([])
    assert eng.vector_point_scalar_mul_bppp(e0, [1, 2]) == INF


def test_msm_multi_matches_vec_com(eng, orc):
    """vecCom / productCom shape: <a,G> + <b,H> + c*U as one call (go/bp/util.go:96-104)."""
    n = 64
    graw = orc.synth_bases(0x77, 2 * n + 1)
    sraw = orc.synth_scalars(0x78, 2 * n + 1)
    sc = [int.from_bytes(sraw[32 * i:32 * i + 32], "big") for i in range(2 * n + 1)]
    G = eng.upload(graw[:64 * n])
    H = eng.upload(graw[64 * n:64 * 2 * n])
    U = eng.upload(graw[64 * 2 * n:])
    got = eng.msm_multi([(G, 0, sc[:n]), (H, 0, sc[n:2 * n]), (U, 0, sc[2 * n:])])
    assert got == orc.msm(graw, sraw)
    # IPA "L" shape: G[n1:], H[:n1]
    n1 = n // 2
    got = eng.msm_multi([(G, n1, sc[:n1]), (H, 0, sc[n1:n])])
    want = orc.msm(graw[64 * n1:64 * n] + graw[64 * n:64 * (n + n1)], sraw[:32 * n])
    assert got == want


def test_msm_multi_submit_matches_blocking(eng, orc):
    """bn256_msm_multi_submit / bn256_msm_wait: L, R, P of an inner-product level (go/bp/main.go:340-347)
    in flight together give the bytes of the blocking bn256_msm_multi; single segments and segments over
    a fixed-base table go through the same tickets."""
    n = 64
    braw = orc.synth_bases(0x51, 2 * n)
    sraw = orc.synth_scalars(0x52, 2 * n)
    a = [int.from_bytes(sraw[32 * i:32 * i + 32], "big") for i in range(2 * n)]
    G, H = eng.upload(braw[:64 * n]), eng.upload(braw[64 * n:])
    h = n // 2
    shapes = [[(G, h, a[:h]), (H, 0, a[n + h:])], [(G, 0, a[h:n]), (H, h, a[n:n + h])], [(G, 0, a[:n]), (H, 0, a[n:])]]
    want = [eng.msm_multi(sg) for sg in shapes]
    assert want[2] == orc.msm(braw, sraw)
    tickets = [eng.msm_multi_submit(sg) for sg in shapes]
    spare = [eng.msm_multi_submit([]) for _ in range(crypto_amd.ASYNC_LANES - 3)]
    with pytest.raises(Exception):
        eng.msm_multi_submit(shapes[0])  # BN256_EBUSY
    assert [eng.msm_wait(t) for t in reversed(tickets)] == list(reversed(want))
    assert all(eng.msm_wait(t) == bytes(64) for t in spare)
    # one segment; no segment; several segments over fixed-base tables of >= 2^15 points
    t1, t0 = eng.msm_multi_submit([(G, 3, a[:17])]), eng.msm_multi_submit([])
    assert eng.msm_wait(t1) == eng.msm_multi([(G, 3, a[:17])]) and eng.msm_wait(t0) == INF
    m = 1 << 15
    T1, T2 = eng.random_bases(0x53, m, flags=crypto_amd.PRECOMPUTE), eng.random_bases(0x54, m, flags=crypto_amd.PRECOMPUTE)
    big = [int.from_bytes(hashlib.sha256(b"%d" % i).digest(), "big") % N_ORDER for i in range(m)]
    tb = eng.msm_multi_submit([(T1, 0, big), (T2, 0, big[::-1])])
    assert eng.msm_wait(tb) == eng.msm_multi([(T1, 0, big), (T2, 0, big[::-1])])


def test_linearity_and_permutation_at_2_20(eng):
    """Size-independent properties at the headline size (2^20): MSM(a) + MSM(b) == MSM(a+b),
    MSM over a permuted-range split equals the whole, MSM(k * ones-vector) consistency."""
    import torch
    import crypto_amd
    n = 1 << 20
    h = eng.random_bases(0x5EED0001, n, flags=crypto_amd.PRECOMPUTE)
    a = torch.empty(n * 32, dtype=torch.uint8, device="cuda:0")
    b = torch.empty(n * 32, dtype=torch.uint8, device="cuda:0")
    eng.random_scalars_into(0x5EED0002, n, a.data_ptr())
    eng.random_scalars_into(0x5EED0003, n, b.data_ptr())
    ra = eng.msm_dev(h, a.data_ptr(), n)
    rb = eng.msm_dev(h, b.data_ptr(), n)
    # a + b mod n on the host (numpy big-int via python ints, vectorised through object arrays is slow;
    # use the split property instead): MSM over [0,n/2) + MSM over [n/2,n) == MSM over [0,n)
    half = n // 2
    r0 = eng.msm_dev(h, a.data_ptr(), half, off=0)
    r1 = eng.msm_dev(h, a.data_ptr() + 32 * half, half, off=half)
    assert eng.g1_sum([r0, r1]) == ra
    # n * P == infinity for the outputs (they are in the prime-order group)
    hr = eng.upload([ra, rb])
    assert eng.msm(hr, [N_ORDER - 1, 0]) == ref.marshal(ref.neg(ref.unmarshal(ra)))
    # non-precomputed path agrees with the fixed-base path on a 2^18 prefix
    m = 1 << 18
    hp = eng.random_bases(0x5EED0001, m)
    assert eng.msm_dev(hp, a.data_ptr(), m) == eng.msm_dev(h, a.data_ptr(), m)
    assert ra != rb and ra != INF


def test_elementwise_helpers(eng, orc, golden):
    ew = golden["elementwise"]
    braw = orc.synth_bases(0x5EED0001, 8)
    sraw = orc.synth_scalars(0x5EED0002, 9)
    a = [int.from_bytes(sraw[32 * i:32 * i + 32], "big") for i in range(8)]
    x = int(ew["x"], 16)
    g = eng.upload(braw)
    assert [p.hex() for p in eng.vector_point_mul_on_scalar(g, x).download()] == ew["mul_on_scalar"]
    assert [p.hex() for p in eng.scalar_mul_each(g, a).download()] == ew["mul_each"]
    lo, hi = eng.upload(braw[:256]), eng.upload(braw[256:])
    assert [p.hex() for p in eng.hadamard_point_mul(lo, hi).download()] == ew["hadamard_0_4"]
    xinv = pow(x, -1, N_ORDER)
    assert [p.hex() for p in eng.fold(g, 0, g, 4, 4, xinv, x).download()] == ew["fold_x_xinv"]
    # WNLA-style even/odd fold with stride 2 (go/bppp/docs_test.go:1325-1326)
    got = eng.fold(g, 0, g, 1, 4, 3, 5, stride=2).download()
    for i in range(4):
        want = orc.add(orc.mul(braw[128 * i:128 * i + 64], 3), orc.mul(braw[128 * i + 64:128 * i + 128], 5))
        assert got[i] == want


def test_elementwise_edge_cases(eng, orc):
    g0 = ref.marshal(ref.G0)
    pts = [g0, INF, ref.marshal(ref.mul(5, ref.G0)), ref.marshal(ref.neg(ref.mul(5, ref.G0)))]
    h = eng.upload(pts)
    # scalar 0, n, 1, n-1
    got = eng.scalar_mul_each(h, [0, 7, N_ORDER, N_ORDER - 1]).download()
    assert got == [INF, INF, INF, ref.marshal(ref.mul(5, ref.G0))]
    # P + inf, inf + inf, P + P, P + (-P)
    a = eng.upload([g0, INF, pts[2], pts[2]])
    b = eng.upload([INF, INF, pts[2], pts[3]])
    assert eng.hadamard_point_mul(a, b).download() == [g0, INF, ref.marshal(ref.mul(10, ref.G0)), INF]


def test_scalar_mul_batch_1024_vs_oracle(eng, orc):
    n = 1024
    braw = orc.synth_bases(0x99, n)
    sraw = orc.synth_scalars(0x9A, n)
    h = eng.upload(braw)
    sc = [int.from_bytes(sraw[32 * i:32 * i + 32], "big") for i in range(n)]
    assert b"".join(eng.scalar_mul_each(h, sc).download()) == orc.scalar_mul_each(braw, sraw)
    assert b"".join(eng.vector_point_mul_on_scalar(h, sc[0]).download()) == orc.scalar_mul_each(braw, sraw[:32], broadcast=True)


def test_fold_submit_wait(eng, orc):
    """bn256_fold2_submit / bn256_fold_wait: both generator folds of a level (go/bp/main.go:355-356) in
    flight, a blocking fold and an MSM ticket beside them; ticket kinds do not mix."""
    n = 96
    g = eng.random_bases(0x61, 2 * n)
    x = 0x1234567890abcdef << 150 | 0xfeed
    xinv = pow(x, -1, N_ORDER)
    want_g = eng.fold(g, 0, g, n, n, xinv, x).download()
    want_h = eng.fold(g, 0, g, n, n, x, xinv).download()
    want_r = eng.fold2(g, 0, 5, g, 7, 3, x, xinv, stride=2).download()
    hg, tg = eng.fold2_submit(g, 0, n, g, n, n, xinv, x)
    hh, th = eng.fold2_submit(g, 0, n, g, n, n, x, xinv)
    hr, tr = eng.fold2_submit(g, 0, 5, g, 7, 3, x, xinv, stride=2)
    assert len({tg, th, tr}) == 3
    spare = [eng.msm_multi_submit([]) for _ in range(crypto_amd.ASYNC_LANES - 3)]
    with pytest.raises(Exception):
        eng.fold2_submit(g, 0, n, g, n, n, x, xinv)  # BN256_EBUSY
    assert all(eng.msm_wait(t) == bytes(64) for t in spare)
    with pytest.raises(Exception):
        eng.msm_wait(tg)  # a fold ticket
    assert eng.fold(g, 0, g, n, n, 3, 5).download()[0] == orc.add(orc.mul(g.download(0, 1)[0], 3),
                                                                   orc.mul(g.download(n, 1)[0], 5))
    eng.fold_wait(th)
    tm = eng.msm_multi_submit([(g, 0, [1, 2, 3])])
    with pytest.raises(Exception):
        eng.fold_wait(tm)  # an MSM ticket
    eng.fold_wait(tr)
    eng.fold_wait(tg)
    assert eng.msm_wait(tm) == eng.msm_multi([(g, 0, [1, 2, 3])])
    assert hg.download() == want_g and hh.download() == want_h and hr.download() == want_r
    e, te = eng.fold2_submit(g, 0, 0, g, 0, 0, x, xinv)
    eng.fold_wait(te)
    assert len(e) == 0


def test_glv_paths_randomised_against_oracle(eng, orc):
    """The endomorphism-split kernels (per-point scalars, shared-scalar folds, 2-term commitments) on
    seeded random sizes and scalars of every bit length -- including 0, 1, n-1, values >= n and
    scalars whose GLV halves are tiny or change sign -- against the oracle's double-and-add."""
    import random
    rng = random.Random(0x61C)
    lam = 0x196ac037f07e9f9eecf9176e4f8bfcd513be518b5264ac23d
    special = [0, 1, 2, N_ORDER - 1, N_ORDER, N_ORDER + 1, 2**256 - 1, lam, lam + 1, N_ORDER - lam, (N_ORDER - 1) // 2,
               (N_ORDER + 1) // 2, 2**128, 2**129 - 1, 2**64, 3 * lam % N_ORDER]

    def rscalar():
        r = rng.random()
        if r < 0.25:
            return rng.choice(special)
        return rng.randrange(2 ** rng.randrange(1, 257))

    for trial in range(10):
        n = rng.choice([1, 2, 3, 5, 17, 64, 129, 255])
        braw = eng.random_bases(0x700 + trial, 2 * n).download()
        if trial % 3 == 0:
            braw[rng.randrange(2 * n)] = INF
        g = eng.upload(braw)
        ks = [rscalar() for _ in range(n)]
        want = orc.scalar_mul_each(b"".join(braw[:n]), b"".join((k % 2**256).to_bytes(32, "big") for k in ks))
        assert b"".join(eng.scalar_mul_each(g, ks).download()) == want, trial
        a, b = rscalar(), rscalar()
        got = eng.fold(g, 0, g, n, n, a, b).download()
        wa = orc.scalar_mul_each(b"".join(braw[:n]), a.to_bytes(32, "big"), broadcast=True)
        wb = orc.scalar_mul_each(b"".join(braw[n:]), b.to_bytes(32, "big"), broadcast=True)
        assert b"".join(got) == orc.add_each(wa, wb), trial
        assert b"".join(eng.vector_point_mul_on_scalar(g, a, 0, n).download()) == wa, trial
        p0, p1 = braw[0], braw[n]
        if p0 != INF and p1 != INF:
            aa, bb = [rscalar() for _ in range(n)], [rscalar() for _ in range(n)]
            gotc = eng.com_batch(p0, p1, aa, bb).download()
            w0 = orc.scalar_mul_each(p0 * n, b"".join(x.to_bytes(32, "big") for x in aa))
            w1 = orc.scalar_mul_each(p1 * n, b"".join(x.to_bytes(32, "big") for x in bb))
            assert b"".join(gotc) == orc.add_each(w0, w1), trial


def test_fold_lane_per_output_kernel(eng, orc):
    """Generator folds of more than 24576 outputs run one lane per output (k_fold_points), up to 2^13
    outputs two GLV half-chains of one quad per output, in between one quad over the full chain: the
    three against one another (the whole result against <= 2^13-point pieces), edge cases
    and a sample against the oracle (go/bp/main.go:355-356, vectorPointsAdd padding go/bppp/util.go:279-293)."""
    n = (1 << 15) + 77
    lo_raw = eng.random_bases(0xF01D, n).download()
    hi_raw = eng.random_bases(0xF01E, n).download()
    neg = lambda p: p[:32] + ((ref.P - int.from_bytes(p[32:], "big")) % ref.P).to_bytes(32, "big")  # noqa: E731
    hi_raw[3] = lo_raw[3]        # P + P
    hi_raw[5] = neg(lo_raw[5])   # P - P
    lo_raw[7] = INF
    hi_raw[9] = INF
    lo_raw[11] = hi_raw[11] = INF
    lo, hi = eng.upload(lo_raw), eng.upload(hi_raw)
    s_lo = 0x1d0c0ffee << 190 | 12345
    s_hi = N_ORDER - 0x77777777777
    for a, b in ((s_lo, s_hi), (1, s_hi), (s_hi, s_hi), (s_lo, N_ORDER - s_lo)):
        got = eng.fold(lo, 0, hi, 0, n, a, b).download()
        want = []
        for off in range(0, n, 1 << 13):
            want += eng.fold(lo, off, hi, off, min(1 << 13, n - off), a, b).download()
        assert got == want
        # 2^13 < outputs <= 24576: one quad per output over the full-length chain
        assert eng.fold(lo, 0, hi, 0, 1 << 14, a, b).download() == want[:1 << 14]
        for i in (0, 3, 5, 7, 9, 11, 64, 255, 256, 8191, 16384, 32768, n - 1):
            assert got[i] == orc.add(orc.mul(lo_raw[i], a), orc.mul(hi_raw[i], b)), i
    # ragged: the missing side counts as infinity
    got = eng.fold2(lo, 0, n, hi, 0, n - 5, s_lo, s_hi).download()
    assert got[:n - 5] == eng.fold(lo, 0, hi, 0, n, s_lo, s_hi).download()[:n - 5]
    assert got[n - 5:] == eng.vector_point_mul_on_scalar(lo, s_lo, n - 5, 5).download()
    # vectorPointMulOnScalar (go/bp/util.go:177-183) through the same kernel
    got = eng.vector_point_mul_on_scalar(lo, s_hi).download()
    want = []
    for off in range(0, n, 1 << 13):
        want += eng.vector_point_mul_on_scalar(lo, s_hi, off, min(1 << 13, n - off)).download()
    assert got == want
    assert got[100] == orc.mul(lo_raw[100], s_hi) and got[7] == INF


@pytest.mark.parametrize("mode", ["plain", "precomp"])
def test_skewed_scalar_distributions_65536(eng, orc, mode):
    """Bulletproofs-like scalars (go/bp/main.go:84-85: al in {0,1}, ar in {0,n-1}) and an
    all-equal vector put most terms into one or two buckets; result must still be bit-exact."""
    import crypto_amd
    n = 1 << 16
    braw = orc.synth_bases(0xBB01, n)
    h = eng.upload(braw, crypto_amd.PRECOMPUTE if mode == "precomp" else 0)
    half = n // 2
    sc = [(i * 2654435761 >> 7) & 1 for i in range(half)] + [(N_ORDER - 1) if (i * 40503 >> 5) & 1 else 0 for i in range(half)]
    sraw = b"".join(s.to_bytes(32, "big") for s in sc)
    assert eng.msm(h, sraw) == orc.msm(braw, sraw, mt=True)
    sraw = (5).to_bytes(32, "big") * n
    assert eng.msm(h, sraw) == orc.msm(braw, sraw, mt=True)
    sraw = ((1 << 240) + 3).to_bytes(32, "big") * n
    assert eng.msm(h, sraw) == orc.msm(braw, sraw, mt=True)


def test_two_contexts_are_independent(orc):
    """Distinct contexts share nothing (include/bn256msm.h conventions)."""
    import crypto_amd
    e1, e2 = crypto_amd.Engine(0), crypto_amd.Engine(0)
    try:
        n = 300
        braw = orc.synth_bases(0xE1, n)
        sraw = orc.synth_scalars(0xE2, n)
        h1, h2 = e1.upload(braw), e2.upload(braw[:64 * 100], crypto_amd.PRECOMPUTE)
        want1 = orc.msm(braw, sraw)
        want2 = orc.msm(braw[:64 * 100], sraw[:32 * 100])
        for _ in range(3):
            assert e1.msm(h1, sraw) == want1
            assert e2.msm(h2, sraw[:32 * 100]) == want2
        h1.free()
        h2.free()
    finally:
        e1.close()
        e2.close()


def test_sub_range_passes_match_single_pass(orc, monkeypatch):
    """Large fixed-base calls run as several passes whose bucket arrays are added before the
    reduction (MSM_CHUNK); forced here at 2^10 points per pass, ragged last pass included."""
    import crypto_amd
    monkeypatch.setenv("BN256_MSM_CHUNK_LOG2", "10")
    e = crypto_amd.Engine(0)
    try:
        for n in (1024, 1025, 5000):
            bases = e.random_bases(0xC0 + n, n, flags=crypto_amd.PRECOMPUTE)
            v = e.fr_random(0xD0 + n, n)
            exp = orc.msm(orc.synth_bases(0xC0 + n, n), orc.synth_scalars(0xD0 + n, n))
            assert e.msm_fr(bases, 0, 1, n, v) == exp
            # a sub-range that starts inside a pass
            exp2 = orc.msm(orc.synth_bases(0xC0 + n, n)[64 * 7:], orc.synth_scalars(0xD0 + n, n)[32 * 3:32 * (n - 4)])
            assert e.msm_fr(bases, 7, 1, n - 7, v, 3, 1) == exp2
            bases.free()
    finally:
        e.close()


def test_com_batch_matches_oracle(eng):
    """k = 128 two-term commitments (go/ve-ca/main.go:37-40 shape) against com(), go/bp/util.go:92-94."""
    import random
    rng = random.Random(3)
    g, h = ref.base_mul(rng.randrange(1, N_ORDER)), ref.base_mul(rng.randrange(1, N_ORDER))
    a = [rng.randrange(N_ORDER) for _ in range(128)]
    b = [rng.randrange(N_ORDER) for _ in range(128)]
    a[:6] = [0, 1, 0, N_ORDER - 1, 2**256 - 1, 5]
    b[:6] = [0, 0, 1, 1, 7, N_ORDER - 5]
    got = eng.com_batch(ref.marshal(g), ref.marshal(h), a, b).download()
    exp = [ref.marshal(ref.com(g, h, x, y)) for x, y in zip(a, b)]
    assert got == exp
    # h = g and h = -g exercise the doubling / cancellation paths of g + h
    got = eng.com_batch(ref.marshal(g), ref.marshal(g), a[:8], b[:8]).download()
    assert got == [ref.marshal(ref.mul((x + y) % N_ORDER, g)) for x, y in zip(a[:8], b[:8])]
    got = eng.com_batch(ref.marshal(g), ref.marshal(ref.neg(g)), a[:8], b[:8]).download()
    assert got == [ref.marshal(ref.mul((x - y) % N_ORDER, g)) for x, y in zip(a[:8], b[:8])]
    assert eng.com_batch(ref.marshal(g), ref.marshal(h), [], []).download() == []


def test_two_pass_sort_matches_single_pass(monkeypatch):
    """The two-pass sort (>= 2^18 points with a table) against the single-pass sort on the same
    inputs: uniform, Bulletproofs-like skew (one group holds nearly every entry), a ragged
    chunked call whose last pass falls back, and one that does not."""
    import numpy as np
    import torch
    import crypto_amd
    monkeypatch.setenv("BN256_SORT1", "1")
    e1 = crypto_amd.Engine(0)
    monkeypatch.delenv("BN256_SORT1")
    e2 = crypto_amd.Engine(0)
    try:
        for n, flags in (((1 << 18), 1), ((1 << 18) + 12345, 1), ((1 << 21) + 5000, 1), ((1 << 21) + 300000, 1),
                         ((1 << 18), 0), ((1 << 19) + 777, 0), ((1 << 20) + 10, 0)):
            b1 = e1.random_bases(0xE0, n, flags=flags)
            b2 = e2.random_bases(0xE0, n, flags=flags)
            v1, v2 = e1.fr_random(0xE1, n), e2.fr_random(0xE1, n)
            assert e1.msm_fr(b1, 0, 1, n, v1) == e2.msm_fr(b2, 0, 1, n, v2)
            # stride-2 views of points and scalars (reducePoints / reduceVector halves), odd offsets
            m = (n - 1) // 2
            assert e1.msm_fr(b1, 1, 2, m, v1, 0, 2) == e2.msm_fr(b2, 1, 2, m, v2, 0, 2)
            if n == 1 << 18:  # with and without a table
                bits = np.random.default_rng(1).integers(0, 2, size=n // 2, dtype=np.uint8)
                arr = np.zeros((n, 32), dtype=np.uint8)
                arr[: n // 2, 31] = bits
                arr[n // 2:][bits == 0] = np.frombuffer((N_ORDER - 1).to_bytes(32, "big"), dtype=np.uint8)
                raw = arr.tobytes()
                s1, s2 = e1.fr_upload(raw), e2.fr_upload(raw)
                r1 = e1.msm_fr(b1, 0, 1, n, s1)
                assert r1 == e2.msm_fr(b2, 0, 1, n, s2) and r1 != INF
                # degenerate vectors: no entry at all, and every entry in one bucket
                for val in (0, 1, N_ORDER - 1, N_ORDER, (1 << 255) + 12345):
                    raw = (val % (1 << 256)).to_bytes(32, "big") * n
                    s1, s2 = e1.fr_upload(raw), e2.fr_upload(raw)
                    r1, r2 = e1.msm_fr(b1, 0, 1, n, s1), e2.msm_fr(b2, 0, 1, n, s2)
                    assert r1 == r2
                    assert (r1 == INF) == (val % N_ORDER == 0)
                    s1.free()
                    s2.free()
            b1.free()
            b2.free()
    finally:
        e1.close()
        e2.close()


@pytest.mark.parametrize("log2n,flags", [(20, 1), (20, 0), (22, 1), (24, 1)])
def test_full_size_exact_answer(eng, orc, log2n, flags):
    """BASELINE.json sizes (2^20, 2^24 points): the device result against the closed form of the
    synthetic workload, (sum_i s_i k_i mod n) * G0 -- an exact, bit-for-bit check at sizes where
    the reference's per-term loop would take hours on the CPU."""
    n = 1 << log2n
    bases = eng.random_bases(0x5EED0001, n, flags=flags)
    v = eng.fr_random(0x5EED0002, n)
    try:
        got = eng.msm_fr(bases, 0, 1, n, v)
        assert got == orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, n)
        # a sub-range that starts and ends inside passes / tiles
        off, m = 12345, n - 54321
        assert eng.msm_fr(bases, off, 1, m, v, off, 1) == orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, m, first=off)
    finally:
        bases.free()
        v.free()


def test_submit_wait_matches_blocking_calls(eng, orc):
    """bn256_msm_submit / bn256_msm_wait (include/bn256msm.h): several commitments in flight at once
    (A and S of go/bp/main.go:89,95 are independent) return exactly what two blocking calls return;
    sizes: the headline 2^20 with a table (closed form as the exact answer), a chunked 2^22 call,
    a small table-less one against the oracle's term loop, and the empty call."""
    import torch
    import crypto_amd
    from crypto_amd.engine import BN256Error
    n = 1 << 20
    h = eng.random_bases(0x5EED0001, n, flags=crypto_amd.PRECOMPUTE)
    a = torch.empty(n * 32, dtype=torch.uint8, device="cuda:0")
    b = torch.empty(n * 32, dtype=torch.uint8, device="cuda:0")
    eng.random_scalars_into(0x5EED0002, n, a.data_ptr())
    eng.random_scalars_into(0x5EED0003, n, b.data_ptr())
    want_a = orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, n)
    want_b = eng.msm_dev(h, b.data_ptr(), n)
    for _ in range(3):  # lanes are reused; the second and third rounds run on warm workspaces
        ta = eng.msm_submit(h, a.data_ptr(), n)
        tb = eng.msm_submit(h, b.data_ptr(), n)
        tc = eng.msm_submit(h, a.data_ptr(), n)
        assert len({ta, tb, tc}) == 3
        spare = [eng.msm_submit(h, a.data_ptr(), 0) for _ in range(crypto_amd.ASYNC_LANES - 3)]
        with pytest.raises(BN256Error) as ei:  # every lane taken
            eng.msm_submit(h, a.data_ptr(), n)
        assert ei.value.code == -8
        assert all(eng.msm_wait(t) == bytes(64) for t in spare)
        assert eng.msm_wait(tb) == want_b  # any order
        assert eng.msm_wait(tc) == want_a
        assert eng.msm_wait(ta) == want_a
    with pytest.raises(BN256Error):  # a ticket is good for one wait
        eng.msm_wait(ta)
    # sub-range + a blocking call while a submitted one is in flight
    off, m = 4097, (1 << 19) + 3
    t = eng.msm_submit(h, a.data_ptr() + 32 * off, m, off=off)
    assert eng.msm_dev(h, b.data_ptr(), n) == want_b
    assert eng.msm_wait(t) == orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, m, first=off)
    # empty call and range errors
    t = eng.msm_submit(h, a.data_ptr(), 0)
    assert eng.msm_wait(t) == INF
    with pytest.raises(BN256Error) as ei:
        eng.msm_submit(h, a.data_ptr(), n, off=1)
    assert ei.value.code == -2
    h.free()
    # table-less small call against the term loop
    k = 1000
    braw, sraw = orc.synth_bases(0xBB07, k), orc.synth_scalars(0xCC07, k)
    hs = eng.upload(braw)
    sc = torch.frombuffer(bytearray(sraw), dtype=torch.uint8).to("cuda:0")
    t1 = eng.msm_submit(hs, sc.data_ptr(), k)
    t2 = eng.msm_submit(hs, sc.data_ptr(), k - 1)
    assert eng.msm_wait(t1) == orc.msm(braw, sraw)
    assert eng.msm_wait(t2) == orc.msm(braw[:64 * (k - 1)], sraw[:32 * (k - 1)])
    hs.free()


def test_submit_wait_chunked_2_22(eng, orc):
    """A submitted call of several passes (2^22 points = two 2^21-point passes) never blocks in submit
    and gives the closed-form answer."""
    import crypto_amd
    n = 1 << 22
    h = eng.random_bases(0x5EED0001, n, flags=crypto_amd.PRECOMPUTE)
    v = eng.fr_random(0x5EED0002, n)
    try:
        t1 = eng.msm_submit(h, v.device_ptr(), n)
        t2 = eng.msm_submit(h, v.device_ptr() + 32 * 5, n - 5, off=5)
        assert eng.msm_wait(t1) == orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, n)
        assert eng.msm_wait(t2) == orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, n - 5, first=5)
    finally:
        h.free()
        v.free()


def test_inner_product_h_walkthrough(eng, orc):
    """InnerProductH (go/bp/util.go:15-27) on the reference's own inputs, TestInnerArgumentProof
    (go/bp/docs_test.go:381-430): a = [4,5,10,1], b = [2,1,2,10]; L, R, P and pcheck against the oracle's
    restatement, and the identity the walkthrough checks with bytes.Equal: x^2 L + P + x^-2 R == pcheck."""
    import bn256_ref as ref
    n, n1 = 4, 2
    graw = orc.synth_bases(0xA6, 2 * n + 1)
    pts = [graw[64 * i:64 * i + 64] for i in range(2 * n + 1)]
    g_raw, h_raw, u_raw = pts[:n], pts[n:2 * n], pts[2 * n]
    g_ref, h_ref, u_ref = [ref.unmarshal(p) for p in g_raw], [ref.unmarshal(p) for p in h_raw], ref.unmarshal(u_raw)
    G, H = eng.upload(g_raw), eng.upload(h_raw)
    a, b = [4, 5, 10, 1], [2, 1, 2, 10]
    z = [0] * n1
    N = ref.N
    vm = lambda x, y: sum(p * q for p, q in zip(x, y)) % N  # noqa: E731  vectorMul
    c = vm(a, b)
    assert c == 43  # the reference's comment says 39 (go/bp/docs_test.go:392); vectorMul gives 8 + 5 + 20 + 10

    def both(*args):
        got = eng.inner_product_h(n, G, H, u_raw, *args)
        assert got == ref.marshal(ref.inner_product_h(n, g_ref, h_ref, u_ref, *args))
        return got

    L = both(z, a[:n1], b[n1:], z, vm(a[:n1], b[n1:]))
    R = both(a[n1:], z, z, b[:n1], vm(a[n1:], b[:n1]))
    P = both(a[:n1], a[n1:], b[:n1], b[n1:], c)
    x = int.from_bytes(orc.synth_scalars(0xA7, 1), "big")
    xinv = pow(x, -1, N)
    a1 = [(p * x + q * xinv) % N for p, q in zip(a[:n1], a[n1:])]
    b1 = [(p * x + q * xinv) % N for p, q in zip(b[n1:], b[:n1])]
    sm = lambda v, k: [(e * k) % N for e in v]  # noqa: E731  vectorMulOnScalar
    pcheck = both(sm(a1, xinv), sm(a1, x), sm(b1, x), sm(b1, xinv), vm(a1, b1))
    x2 = x * x % N
    U1 = eng.upload([L, P, R])  # x^2 L + P + x^-2 R through the device as well
    assert eng.msm(U1, [x2, 1, pow(x2, -1, N)]) == pcheck
    # the reference's panics
    with pytest.raises(ValueError, match="invalid n"):
        eng.inner_product_h(3, G, H, u_raw, z, z, z, z, 0)
    with pytest.raises(ValueError, match="invalid length for scalar mul"):
        eng.inner_product_h(n, G, H, u_raw, a, a[n1:], b[:n1], b[n1:], c)
    # a larger even n, random scalars, U as a resident handle
    n = 64
    graw = orc.synth_bases(0xA8, 2 * n + 1)
    sraw = orc.synth_scalars(0xA9, 2 * n + 1)
    sc = [int.from_bytes(sraw[32 * i:32 * i + 32], "big") for i in range(2 * n + 1)]
    G2, H2, U2 = eng.upload(graw[:64 * n]), eng.upload(graw[64 * n:128 * n]), eng.upload(graw[128 * n:])
    got = eng.inner_product_h(n, G2, H2, U2, sc[:32], sc[32:64], sc[64:96], sc[96:128], sc[128])
    assert got == orc.msm(graw, sraw)
