"""GPU parity tests of the device Fr vector kernels (include/bn256msm.h, bn256_fr_*) against
the oracle's restatement of go/bp/util.go:106-163 and go/bppp/util.go:143-176,198-211,245-260.
Integer work: bit-exact."""
import random

import pytest

import crypto_amd

import bn256_ref as ref

pytestmark = pytest.mark.gpu

N = ref.N


def _vals(rng, n, edge=True):
    v = [rng.randrange(N) for _ in range(n)]
    if edge and n >= 6:
        v[0], v[1], v[2], v[3] = 0, 1, N - 1, (N - 1) // 2
    return v


def test_upload_download_and_reduction(eng):
    rng = random.Random(1)
    v = _vals(rng, 37)
    h = eng.fr_upload(v)
    assert len(h) == 37
    assert h.download() == v
    assert h.download(5, 3) == v[5:8]
    h.free()
    e = eng.fr_upload([])
    assert len(e) == 0 and e.download() == []


def test_fr_random_is_the_synthetic_stream(eng, orc):
    h = eng.fr_random(0x5EED0002, 300, first=7)
    raw = b"".join(x.to_bytes(32, "big") for x in h.download())
    assert raw == orc.synth_scalars(0x5EED0002, 307)[32 * 7:]


# Small literals checkable by hand.  The reference pins no vectorMul / weightVectorMul output of
# its own (go/bppp/util_test.go pins only the poly* helpers); go/bp/docs_test.go:389-392 is the
# one literal inner product it mentions.
def test_small_literal_values(eng):
    a = eng.fr_upload([1, 2, 3, 4])
    b = eng.fr_upload([4, 3, 2, 1])
    assert eng.fr_dot(a, 0, 1, 4, b, 0, 1, 4) == 20 == ref.vector_mul_bppp([1, 2, 3, 4], [4, 3, 2, 1])
    assert eng.fr_dot(a, 0, 1, 4, b, 0, 1, 4, mu=2) == ref.weight_vector_mul([1, 2, 3, 4], [4, 3, 2, 1], 2)
    # go/bp/docs_test.go:389-392 (the comment there says 39; vectorMul returns 43, see test_oracle.py)
    x = eng.fr_upload([1, 2, 3, 4, 5])
    y = eng.fr_upload([5, 4, 3, 2, 4])
    assert eng.fr_dot(x, 0, 1, 5, y, 0, 1, 5) == ref.vector_mul([1, 2, 3, 4, 5], [5, 4, 3, 2, 4])


@pytest.mark.parametrize("n", [1, 2, 7, 256, 257, 5000, 70001])
def test_dot_matches_oracle(eng, n):
    rng = random.Random(n)
    a, b = _vals(rng, n), _vals(rng, n)
    mu = rng.randrange(1, N)
    ha, hb = eng.fr_upload(a), eng.fr_upload(b)
    assert eng.fr_dot(ha, 0, 1, n, hb, 0, 1, n) == ref.vector_mul(a, b)
    assert eng.fr_dot(ha, 0, 1, n, hb, 0, 1, n, mu=mu) == ref.weight_vector_mul(a, b, mu)
    assert eng.fr_dot(ha, 0, 1, n, ha, 0, 1, n, mu=N - 1) == ref.weight_vector_mul(a, a, N - 1)
    assert eng.fr_dot(ha, 0, 1, n, hb, 0, 1, n, mu=0) == 0
    assert eng.fr_dot(ha, 0, 1, n, hb, 0, 1, n, mu=1) == ref.vector_mul(a, b)


@pytest.mark.parametrize("n", [2, 3, 10, 11, 4097])
def test_even_odd_views(eng, n):
    """reduceVector halves (go/bppp/util.go:198-211) as stride-2 views, ragged for odd n."""
    rng = random.Random(100 + n)
    a, b = _vals(rng, n, edge=False), _vals(rng, n, edge=False)
    ha, hb = eng.fr_upload(a), eng.fr_upload(b)
    a0, a1 = ref.reduce_vector(a)
    b0, b1 = ref.reduce_vector(b)
    mu = rng.randrange(N)
    n0, n1 = len(a0), len(a1)
    assert eng.fr_dot(ha, 0, 2, n0, hb, 1, 2, n1) == ref.vector_mul_bppp(a0, b1)
    assert eng.fr_dot(ha, 1, 2, n1, hb, 0, 2, n0) == ref.vector_mul_bppp(a1, b0)
    assert eng.fr_dot(ha, 0, 2, n0, hb, 1, 2, n1, mu=mu) == ref.weight_vector_mul(a0, b1, mu)
    assert eng.fr_dot(ha, 1, 2, n1, ha, 1, 2, n1, mu=mu) == ref.weight_vector_mul(a1, a1, mu)
    y, r = rng.randrange(N), rng.randrange(N)
    got = eng.fr_fold(ha, 0, n0, ha, 1, n1, r, y, stride=2).download()
    exp = ref.vector_add_bppp(ref.vector_mul_on_scalar(a0, r), ref.vector_mul_on_scalar(a1, y))
    assert got == exp
    got = eng.fr_fold(ha, 0, n0, ha, 1, n1, 1, y, stride=2).download()
    assert got == ref.vector_add_bppp(a0, ref.vector_mul_on_scalar(a1, y))


def test_ipa_halves_fold(eng):
    """a' = x a_lo + x^-1 a_hi, go/bp/main.go:357-358."""
    rng = random.Random(5)
    n = 64
    a = _vals(rng, n)
    x = rng.randrange(1, N)
    xi = ref.finv(x)
    ha = eng.fr_upload(a)
    got = eng.fr_fold(ha, 0, n // 2, ha, n // 2, n // 2, x, xi).download()
    exp = ref.vector_add(ref.vector_mul_on_scalar(a[:n // 2], x), ref.vector_mul_on_scalar(a[n // 2:], xi))
    assert got == exp
    # scalars >= n are reduced, not rejected
    got2 = eng.fr_fold(ha, 0, n // 2, ha, n // 2, n // 2, x + N if x + N < 2**256 else x, xi).download()
    assert got2 == exp


def test_range_errors(eng):
    import crypto_amd
    h = eng.fr_upload([1, 2, 3])
    with pytest.raises(crypto_amd.BN256Error) as e:
        eng.fr_dot(h, 0, 1, 4, h, 0, 1, 3)
    assert e.value.code == -2
    with pytest.raises(crypto_amd.BN256Error) as e:
        eng.fr_fold(h, 1, 2, h, 0, 3, 1, 1, stride=2)   # needs h[1], h[3]
    assert e.value.code == -2
    with pytest.raises(crypto_amd.BN256Error):
        eng.fr_dot(h, 0, 0, 1, h, 0, 1, 1)              # zero stride


@pytest.mark.parametrize("n,flags", [(1, 0), (33, 0), (1000, 0), (1000, 1), (20000, 1)])
def test_msm_over_device_scalars(eng, orc, n, flags):
    seed_b, seed_s = 0xB0 + n, 0x50 + n
    bases = eng.random_bases(seed_b, n, flags=flags)
    v = eng.fr_random(seed_s, n)
    exp = orc.msm(orc.synth_bases(seed_b, n), orc.synth_scalars(seed_s, n))
    assert eng.msm_fr(bases, 0, 1, n, v) == exp
    # stride-2 views on both sides: <v_odd, G_even>
    if n >= 4 and not flags:
        pts = bases.download()
        sc = v.download()
        g0 = pts[0::2]
        s1 = sc[1::2]
        m = min(len(g0), len(s1))
        exp2 = orc.msm(b"".join(g0[:m]), b"".join(s.to_bytes(32, "big") for s in s1[:m]))
        assert eng.msm_fr(bases, 0, 2, m, v, 1, 2) == exp2


def test_large_dot_linearity(eng):
    """2^20 elements: <a, b> with weights against the oracle on a sparse b, and bilinearity."""
    n = 1 << 20
    a = eng.fr_random(11, n)
    rng = random.Random(9)
    idx = sorted(rng.sample(range(n), 50))
    bvals = {i: rng.randrange(N) for i in idx}
    braw = bytearray(32 * n)
    for i, x in bvals.items():
        braw[32 * i:32 * i + 32] = x.to_bytes(32, "big")
    b = eng.fr_upload(bytes(braw))
    mu = rng.randrange(N)
    av = {i: a.download(i, 1)[0] for i in idx}
    exp = sum(av[i] * bvals[i] for i in idx) % N
    assert eng.fr_dot(a, 0, 1, n, b, 0, 1, n) == exp
    expw = sum(av[i] * bvals[i] * pow(mu, i + 1, N) for i in idx) % N
    assert eng.fr_dot(a, 0, 1, n, b, 0, 1, n, mu=mu) == expw
    # <a, s*a + t*a> = (s + t) <a, a>
    s, t = rng.randrange(N), rng.randrange(N)
    c = eng.fr_fold(a, 0, n, a, 0, n, s, t)
    aa = eng.fr_dot(a, 0, 1, n, a, 0, 1, n)
    assert eng.fr_dot(a, 0, 1, n, c, 0, 1, n) == (s + t) * aa % N


def test_fused_multi_segment_msm(eng, orc):
    """WNLA X commitment shape: <l1,H0> + <l0,H1> + rho<n1,G0> + rho^-1<n0,G1> as one MSM."""
    rng = random.Random(77)
    nh, ng = 37, 20
    H = eng.random_bases(0xA1, nh)
    G = eng.random_bases(0xA2, ng)
    l = eng.fr_random(0xA3, nh)
    n = eng.fr_random(0xA4, ng)
    Hp, Gp, lv, nv = H.download(), G.download(), l.download(), n.download()
    ro = rng.randrange(1, N)
    roinv = ref.finv(ro)
    h0, h1, g0, g1 = (nh + 1) // 2, nh // 2, (ng + 1) // 2, ng // 2
    segs = [(H, 0, 2, min(h0, h1), l, 1, 2), (H, 1, 2, h1, l, 0, 2),
            (G, 0, 2, min(g0, g1), n, 1, 2, ro), (G, 1, 2, g1, n, 0, 2, roinv)]
    got = eng.msm_fr_multi(segs)
    pts, scs = [], []
    for (pp, po, ps, cnt, sv, so, ss, *sc) in [(Hp, 0, 2, min(h0, h1), lv, 1, 2), (Hp, 1, 2, h1, lv, 0, 2),
                                                (Gp, 0, 2, min(g0, g1), nv, 1, 2, ro), (Gp, 1, 2, g1, nv, 0, 2, roinv)]:
        k = sc[0] if sc else 1
        for i in range(cnt):
            pts.append(pp[po + i * ps])
            scs.append(sv[so + i * ss] * k % N)
    exp = orc.msm(b"".join(pts), b"".join(x.to_bytes(32, "big") for x in scs))
    assert got == exp
    assert eng.msm_fr_multi([]) == bytes(64)
    assert eng.msm_fr_multi([(H, 0, 1, 0, l, 0, 1)]) == bytes(64)
    # the submit / wait form: X and R of a round in flight together, a blocking call in between
    rsegs = [(H, 1, 2, h1, l, 1, 2), (G, 1, 2, g1, n, 1, 2)]
    want_r = eng.msm_fr_multi(rsegs)
    tx, tr, te = eng.msm_fr_multi_submit(segs), eng.msm_fr_multi_submit(rsegs), eng.msm_fr_multi_submit([])
    assert len({tx, tr, te}) == 3
    spare = [eng.msm_fr_multi_submit([]) for _ in range(crypto_amd.ASYNC_LANES - 3)]
    with pytest.raises(Exception):
        eng.msm_fr_multi_submit(rsegs)  # BN256_EBUSY: every lane holds a call
    assert eng.fr_dot(l, 0, 1, nh, l, 0, 1, nh) == sum(x * x for x in lv) % N
    assert eng.msm_wait(tr) == want_r and eng.msm_wait(tx) == exp and eng.msm_wait(te) == bytes(64)
    assert all(eng.msm_wait(t) == bytes(64) for t in spare)
    # larger, table-less two-pass sizes through a lane
    big = 1 << 15
    Hb, lb = eng.random_bases(0xB1, big), eng.fr_random(0xB2, big)
    seg = [(Hb, 0, 2, big // 2, lb, 1, 2), (Hb, 1, 2, big // 2, lb, 0, 2, ro)]
    t1, t2 = eng.msm_fr_multi_submit(seg), eng.msm_fr_multi_submit(seg[:1])
    assert eng.msm_wait(t1) == eng.msm_fr_multi(seg) and eng.msm_wait(t2) == eng.msm_fr_multi(seg[:1])


def test_multi_dot_and_multi_fold_match_the_single_calls(eng):
    """bn256_fr_dot_multi / bn256_fr_fold_multi: the five inner products and three folds of a WNLA round
    (go/bppp/docs_test.go:1297-1311,1327-1334) as one call each -- entry by entry the single calls'
    results, including ragged even / odd views, an empty product and plain next to weighted ones."""
    rng = random.Random(77)
    for nl, nn in ((9, 5), (8, 4), (1001, 500), (2, 1)):
        l, n, c = _vals(rng, nl), _vals(rng, nn), _vals(rng, nl)
        mu2, y, roinv = rng.randrange(1, N), rng.randrange(1, N), rng.randrange(1, N)
        hl, hn, hc = eng.fr_upload(l), eng.fr_upload(n), eng.fr_upload(c)
        ev = lambda m: (0, 2, (m + 1) // 2)              # noqa: E731  reduceVector halves as (off, stride, len)
        od = lambda m: (1 if m > 1 else 0, 2, m // 2)   # noqa: E731
        items = [(hn, *ev(nn), hn, *od(nn), mu2), (hc, *ev(nl), hl, *od(nl)), (hc, *od(nl), hl, *ev(nl)),
                 (hn, *od(nn), hn, *od(nn), mu2), (hc, *od(nl), hl, *od(nl), None), (hl, 0, 1, 0, hl, 0, 1, 0)]
        got = eng.fr_dot_multi(items)
        assert got == [eng.fr_dot(*it[:8], mu=(it[8] if len(it) > 8 else None)) for it in items]
        l0, l1, n0, n1, c0, c1 = l[0::2], l[1::2], n[0::2], n[1::2], c[0::2], c[1::2]
        assert got[0] == ref.weight_vector_mul(n0, n1, mu2) and got[1] == ref.vector_mul_bppp(c0, l1)
        assert got[5] == 0
        folds = [(hc, ev(nl)[0], ev(nl)[2], hc, od(nl)[0], od(nl)[2], 1, y, 2),
                 (hl, ev(nl)[0], ev(nl)[2], hl, od(nl)[0], od(nl)[2], 1, y, 2),
                 (hn, ev(nn)[0], ev(nn)[2], hn, od(nn)[0], od(nn)[2], roinv, y, 2)]
        outs = eng.fr_fold_multi(folds)
        for f, o in zip(folds, outs):
            single = eng.fr_fold(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], stride=2)
            assert o.download() == single.download()
        want_n = ref.vector_add_bppp(ref.vector_mul_on_scalar(n0, roinv), ref.vector_mul_on_scalar(n1, y))
        assert outs[2].download() == want_n
    assert eng.fr_dot_multi([]) == [] and eng.fr_fold_multi([]) == []


def test_wl_times_w_of_test_ac3_on_the_device_matches_the_reference_held_vector(eng):
    """go/bppp/docs_test.go:365 records the 32 values of matrixMulOnVector(w, Wl) for TestAC3 (e = 2): here every
    row is a device inner product (bn256_fr_dot_multi over views of one resident matrix), compared with the
    literal vector (tests/golden/ref_wlw_ac3.json) -- the one place the device's Fr kernels meet a known answer
    that the reference itself holds."""
    import json
    import os
    import ac_cases
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_wlw_ac3.json")) as f:
        want = [int(x) for x in json.load(f)["wl_times_w"]]
    _, _, _, pub, prv = ac_cases.test_ac3()
    w = prv["wl"] + prv["wr"] + prv["wo"]
    nw, nl = pub["Nw"], pub["Nl"]
    hw = eng.fr_upload(w)
    hm = eng.fr_upload([x for row in pub["Wl"] for x in row])
    got = []
    for r0 in range(0, nl, 8):  # up to 8 inner products per device call
        got += eng.fr_dot_multi([(hw, 0, 1, nw, hm, r * nw, 1, nw) for r in range(r0, min(r0 + 8, nl))])
    assert got == want
    hw.free()
    hm.free()


def test_tensor_expand_is_the_folded_generators_msm(eng, orc):
    """bn256_fr_tensor_step_multi / _expand_multi: the scalars W (x) a of an MSM over generators that were not folded.
    Checked (i) element by element against the formula in Python, and (ii) as group elements: t rounds of the WNLA
    fold G' = rho*G0 + y*G1 (go/bppp/docs_test.go:1325-1326) on the device, then <G_t, a>, against ONE MSM over
    the unfolded G with the expanded scalars -- including a ragged length and both pairings (swap, odd half only)."""
    rng = random.Random(77)
    n0, t = 45, 3                                     # 45 -> 23 -> 12 -> 6 virtual generators
    los, his = [rng.randrange(1, N) for _ in range(t)], [rng.randrange(1, N) for _ in range(t)]
    nv = (n0 + (1 << t) - 1) >> t
    a = _vals(rng, nv)
    s_even, s_odd = rng.randrange(N), rng.randrange(N)
    hv = eng.fr_upload(a)
    w, ws = None, []
    for s in range(t):
        w = eng.fr_tensor_step([(w, los[s], his[s])])[0]
        ws.append(w)
    W = [1]
    for s in range(t):
        W = [x * los[s] % N for x in W] + [x * his[s] % N for x in W]
    assert w.download() == W
    outs = eng.fr_tensor_expand([(hv, n0, w, 0, 0, 1, 1), (hv, n0, w, 0, 1, s_even, s_odd), (hv, n0, w, 0, 0, 0, 1),
                                 (hv, n0 - 7, w, 0, 1, 1, 0), (hv, nv, None, 0, 1, 1, 1)])

    def want(n_out, swap, se, so):
        r = []
        for m in range(n_out):
            I, src = m >> t, (m >> t) ^ swap
            r.append(W[m & ((1 << t) - 1)] * (so if I & 1 else se) * (a[src] if src < nv else 0) % N)
        return r

    got = [o.download() for o in outs]
    assert got[0] == want(n0, 0, 1, 1)
    assert got[1] == want(n0, 1, s_even, s_odd)
    assert got[2] == want(n0, 0, 0, 1)
    assert got[3] == want(n0 - 7, 1, 1, 0)
    assert got[4] == [a[i ^ 1] if (i ^ 1) < nv else 0 for i in range(nv)]      # no weights: t = 0
    # (ii) the folds themselves
    g = eng.random_bases(0x7E50, n0)
    cur, owned = g, []
    for s in range(t):
        n = len(cur)
        cur = eng.fold2(cur, 0, (n + 1) // 2, cur, 1, n // 2, los[s], his[s], stride=2)
        owned.append(cur)
    assert len(cur) == nv
    assert eng.msm(cur, a) == eng.msm_fr(g, 0, 1, n0, outs[0])
    # <G_t odd half, a odd half> (the R commitment's shape) and the swapped pairing (X's): as MSMs over strided views
    assert eng.msm_multi([(cur, 1, a[1::2], 2)]) == eng.msm_fr(g, 0, 1, n0, outs[2])
    ev_a, od_a = a[0::2], a[1::2]
    lhs = eng.msm_multi([(cur, 0, [x * s_even % N for x in od_a], 2), (cur, 1, [x * s_odd % N for x in ev_a], 2)])
    assert lhs == eng.msm_fr(g, 0, 1, n0, outs[1])
    for o in outs + owned + ws + [g, hv]:
        o.free()


def test_tensor_expand_half_folds_of_the_inner_product_argument(eng):
    """layout 1: G'[i] = lo*G[i] + hi*G[i + n/2] (go/bp/main.go:355-356).  After t rounds <G_t, a> and the L / R
    shapes <G_t[n1:], a[:n1]>, <G_t[:n1], a[n1:]> (main.go:340-341) against MSMs over the unfolded G."""
    rng = random.Random(78)
    n0, t = 64, 3
    los, his = [rng.randrange(1, N) for _ in range(t)], [rng.randrange(1, N) for _ in range(t)]
    nt = n0 >> t
    a = _vals(rng, nt)
    hv = eng.fr_upload(a)
    w, ws = None, []
    for s in range(t):
        w = eng.fr_tensor_step([(w, los[s], his[s])], layout=1)[0]
        ws.append(w)
    sel = nt.bit_length() - 2                          # top bit of the virtual index
    outs = eng.fr_tensor_expand([(hv, n0, w, sel, 0, 1, 1), (hv, n0, w, sel, 1, 0, 1), (hv, n0, w, sel, 1, 1, 0)], layout=1)
    g = eng.random_bases(0x7E51, n0)
    cur, owned = g, []
    for s in range(t):
        h = len(cur) // 2
        cur = eng.fold(cur, 0, cur, h, h, los[s], his[s])
        owned.append(cur)
    assert len(cur) == nt
    n1 = nt // 2
    assert eng.msm(cur, a) == eng.msm_fr(g, 0, 1, n0, outs[0])
    assert eng.msm(cur, a[:n1], off=n1) == eng.msm_fr(g, 0, 1, n0, outs[1])       # upper generators, lower scalars
    assert eng.msm(cur, a[n1:], off=0) == eng.msm_fr(g, 0, 1, n0, outs[2])        # lower generators, upper scalars
    for o in outs + owned + ws + [g, hv]:
        o.free()


def test_dot_results_left_on_the_device_and_lanes_ordered_behind_deferred_kernels(eng):
    """bn256_fr_dot_multi_dev: the WNLA round's five inner products (go/bppp/docs_test.go:1297-1311) left in HBM --
    the same values as bn256_fr_dot_multi, with and without bn256_ctx_fr_defer_sync; and the ordering the deferred
    mode promises: an MSM SUBMITTED to a lane right behind un-waited Fr kernels (the expanded scalar vectors of a
    round over frozen generators) reads their finished output -- three times as many rounds in a row as there are lanes, each result against the
    blocking call made after a wait."""
    rng = random.Random(99)
    nl, nn = 1001, 500
    l, n, c = _vals(rng, nl), _vals(rng, nn), _vals(rng, nl)
    mu2 = rng.randrange(1, N)
    hl, hn, hc = eng.fr_upload(l), eng.fr_upload(n), eng.fr_upload(c)
    ev = lambda m: (0, 2, (m + 1) // 2)              # noqa: E731
    od = lambda m: (1 if m > 1 else 0, 2, m // 2)   # noqa: E731
    items = [(hn, *ev(nn), hn, *od(nn), mu2), (hc, *ev(nl), hl, *od(nl)), (hc, *od(nl), hl, *ev(nl)),
             (hn, *od(nn), hn, *od(nn), mu2), (hc, *od(nl), hl, *od(nl), None), (hl, 0, 1, 0, hl, 0, 1, 0)]
    want = eng.fr_dot_multi(items)
    d = eng.fr_dot_multi_dev(items)
    assert len(d) == len(items) and d.download() == want
    d.free()
    with pytest.raises(crypto_amd.BN256Error):
        eng.fr_dot_multi_dev([])
    n0 = 4096
    g = eng.random_bases(0x7E51, n0)
    a = eng.fr_upload(_vals(rng, n0 >> 3))
    w = None
    for s in range(3):
        w = eng.fr_tensor_step([(w, rng.randrange(1, N), rng.randrange(1, N))])[0]
    eng.fr_defer_sync(True)
    try:
        keep, tickets, dots = [], [], []
        for r in range(3 * crypto_amd.ASYNC_LANES):
            se, so = rng.randrange(N), rng.randrange(N)
            e = eng.fr_tensor_expand([(a, n0, w, 0, r & 1, se, so)])[0]        # enqueued, not waited for
            tickets.append(eng.msm_fr_multi_submit([(g, 0, 1, n0, e, 0, 1)]))  # the lane waits on the device
            dots.append(eng.fr_dot_multi_dev(items))                           # behind it on the context's stream
            keep.append(e)
            if len(tickets) == crypto_amd.ASYNC_LANES:
                got = [eng.msm_wait(t) for t in tickets]
                assert [x.download() for x in dots] == [want] * len(dots)
                assert got == [eng.msm_fr(g, 0, 1, n0, x) for x in keep[-len(got):]]
                tickets, dots = [], []
        assert not tickets
    finally:
        eng.fr_defer_sync(False)
    for o in keep + [g, a, w, hl, hn, hc]:
        o.free()
