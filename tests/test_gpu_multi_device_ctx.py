"""bn256_ctx_create_multi: one context over several devices behind the C ABI, for
the reference's single-goroutine caller (go/bp/main.go:81-178).  A one-GPU box lists its device more
than once: the shards then share the GPU but are otherwise what they are on a node -- own contexts,
streams, workspaces and host threads.  Bar: the bytes of the single-device context, the oracle's
term loop on fixture-sized inputs, the closed form at 2^20."""
import pytest

pytestmark = pytest.mark.gpu

SEED_B, SEED_S = 0x5EED0001, 0x5EED0002


@pytest.fixture(scope="module")
def multi2():
    import crypto_amd
    e = crypto_amd.Engine(devices=[0, 0])
    yield e
    e.close()


def test_uploaded_points_host_scalars_vs_oracle(multi2, eng, orc, golden):
    n = 1001  # 501 + 500
    braw = bytearray(orc.synth_bases(SEED_B, n))
    braw[64 * 7:64 * 8] = bytes(64)            # an infinity base in shard 0
    braw[64 * 900:64 * 901] = braw[64 * 3:64 * 4]  # a duplicate across the shard boundary
    braw = bytes(braw)
    sraw = bytearray(orc.synth_scalars(SEED_S, n))
    sraw[32 * 500:32 * 501] = bytes(32)        # zero scalar at the boundary
    sraw[32 * 501:32 * 502] = (2**256 - 1).to_bytes(32, "big")  # >= n: reduced like the reference's Mod(Order)
    sraw = bytes(sraw)
    want = orc.msm(braw, sraw, mt=True)
    for flags in (0, 1):
        h = multi2.upload(braw, flags)
        try:
            assert len(h) == n
            assert multi2.msm(h, sraw) == want
            assert b"".join(h.download()) == braw
            assert b"".join(h.download(499, 5)) == braw[64 * 499:64 * 504]
            # sub-ranges: inside shard 0, inside shard 1, across the boundary, empty
            for off, m in ((3, 100), (600, 401), (450, 123), (501, 0), (1001, 0)):
                got = multi2.msm(h, sraw[32 * off:32 * (off + m)], off=off, n=m)
                assert got == (orc.msm(braw[64 * off:64 * (off + m)], sraw[32 * off:32 * (off + m)]) if m else bytes(64))
        finally:
            h.free()


def test_2_20_closed_form_and_single_device_bytes(multi2, eng, orc):
    import crypto_amd
    n = 1 << 20
    h = multi2.random_bases(SEED_B, n, flags=crypto_amd.PRECOMPUTE)
    h1 = eng.random_bases(SEED_B, n, flags=crypto_amd.PRECOMPUTE)
    v = multi2.fr_random(SEED_S, n)           # lives on the context's first device
    try:
        want = orc.synth_msm_closed_form(SEED_B, SEED_S, n)
        assert multi2.msm_dev(h, v.device_ptr(), n) == want           # device pointer
        assert multi2.msm_fr(h, 0, 1, n, v) == want                    # resident Fr vector
        sraw = b"".join(x.to_bytes(32, "big") for x in v.download(0, 70000))
        multi2.set_phase_timing(True)
        part = multi2.msm(h, sraw, off=0, n=70000)                     # host scalars
        assert part == eng.msm_dev(h1, v.device_ptr(), 70000)
        assert eng.msm_dev(h1, v.device_ptr(), n) == want
        t = multi2.last_timing()
        assert len(t) == 7 and t[3] > 0
        multi2.set_phase_timing(False)
        multi2.msm(h, sraw, off=0, n=70000)
        assert multi2.last_timing() == [0.0] * 7                       # off by default: no events, no times
    finally:
        h.free()
        h1.free()
        v.free()


def test_three_shards_more_shards_than_points_and_unsupported_calls(orc):
    import crypto_amd
    from crypto_amd.engine import BN256Error
    e = crypto_amd.Engine(devices=[0, 0, 0])
    try:
        braw, sraw = orc.synth_bases(SEED_B, 2), orc.synth_scalars(SEED_S, 2)
        h = e.upload(braw)                     # shards of 1, 1 and 0 points
        assert e.msm(h, sraw) == orc.msm(braw, sraw)
        assert e.msm(h, b"", n=0) == bytes(64)
        with pytest.raises(BN256Error) as ei:
            e.com_batch(braw[:64], braw[64:], [1], [2])   # no sharded form: BN256_ENOTSUP
        assert ei.value.code == -10
        got = e.vector_point_mul_on_scalar(h, 5)          # element-wise calls shard (slices of 1, 1, 0 points)
        assert b"".join(got.download()) == orc.scalar_mul_each(braw, (5).to_bytes(32, "big"), broadcast=True)
        got.free()
        with pytest.raises(BN256Error) as ei:
            e.msm(h, sraw + sraw)              # longer than the handle: the reference's length panic
        assert ei.value.code == -2
        h.free()
        n = 5000
        hb = e.random_bases(SEED_B, n)
        assert e.msm(hb, orc.synth_scalars(SEED_S, n)) == orc.synth_msm_closed_form(SEED_B, SEED_S, n)
        hb.free()
        assert e.selftest() == 0
    finally:
        e.close()
    with pytest.raises(BN256Error):
        crypto_amd.Engine(devices=[0, 99])
    with pytest.raises(BN256Error):
        crypto_amd.Engine(devices=[])


def _fold_oracle(orc, lo: bytes, hi: bytes, s_lo: int, s_hi: int) -> bytes:
    """[s_lo * lo_i + s_hi * hi_i], a missing side = infinity (vectorPointsAdd's padding, go/bppp/util.go:279-293)."""
    n = max(len(lo), len(hi)) // 64
    lo, hi = lo + bytes(64 * n - len(lo)), hi + bytes(64 * n - len(hi))
    a = orc.scalar_mul_each(lo, s_lo.to_bytes(32, "big"), broadcast=True)
    b = orc.scalar_mul_each(hi, s_hi.to_bytes(32, "big"), broadcast=True)
    return orc.add_each(a, b)


def _elementwise_round(e, orc, n):
    """vectorPointMulOnScalar, the h1 loop, hadamardPointMul and both generator folds (go/bp/util.go:177-195,
    go/bp/main.go:156-159,355-356, go/bppp/docs_test.go:1325-1326) on a sharded handle, against the oracle's
    per-point loops; then an MSM over a result handle (the next IPA level's commitment)."""
    import ctypes as C
    braw = bytearray(orc.synth_bases(SEED_B, n))
    if n > 16:
        braw[64 * 5:64 * 6] = bytes(64)                       # an infinity point
        braw[64 * (n - 2):64 * (n - 1)] = braw[64 * 1:64 * 2]  # duplicates across the shard boundary
    braw = bytes(braw)
    sraw = orc.synth_scalars(SEED_S, n)
    k, k2 = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF, (1 << 255) + 12345
    h = e.upload(braw)
    outs = []
    try:
        r = e.vector_point_mul_on_scalar(h, k)
        outs.append(r)
        assert b"".join(r.download()) == orc.scalar_mul_each(braw, k.to_bytes(32, "big"), broadcast=True)
        off, m = (3, n - 7) if n > 16 else (0, n)             # a sub-range whose slices differ from the handle's
        r = e.vector_point_mul_on_scalar(h, k2, off=off, n=m)
        outs.append(r)
        assert b"".join(r.download()) == orc.scalar_mul_each(braw[64 * off:64 * (off + m)], k2.to_bytes(32, "big"), broadcast=True)
        r = e.scalar_mul_each(h, [int.from_bytes(sraw[32 * i:32 * i + 32], "big") for i in range(n)])
        outs.append(r)
        want_each = orc.scalar_mul_each(braw, sraw)
        assert b"".join(r.download()) == want_each
        # hadamardPointMul with offsets that put the operands on different shards
        m2, ao, bo = n // 2, 1, max(0, n // 2 - 3)
        hd = C.c_void_p()
        e._chk(e.lib.bn256_add_batch(e.ctx, h.h, ao, r.h, bo, m2, C.byref(hd)))
        from crypto_amd.engine import Bases
        rb = Bases(e, hd)
        outs.append(rb)
        assert b"".join(rb.download()) == orc.add_each(braw[64 * ao:64 * (ao + m2)], want_each[64 * bo:64 * (bo + m2)])
        # the IPA fold: g' = x^-1 G_lo + x G_hi over the two halves of one vector (they live on different shards)
        n1 = n // 2
        f = e.fold(h, 0, h, n1, n1, k, k2)
        outs.append(f)
        want_f = _fold_oracle(orc, braw[:64 * n1], braw[64 * n1:64 * 2 * n1], k, k2)
        assert b"".join(f.download()) == want_f
        # the WNLA fold over even / odd positions, ragged (odd n: one more even than odd)
        n_even, n_odd = (n + 1) // 2, n // 2
        f2 = e.fold2(h, 0, n_even, h, 1, n_odd, k2, k, stride=2)
        outs.append(f2)
        ev = b"".join(braw[64 * i:64 * i + 64] for i in range(0, n, 2))
        od = b"".join(braw[64 * i:64 * i + 64] for i in range(1, n, 2))
        assert b"".join(f2.download()) == _fold_oracle(orc, ev, od, k2, k)
        # the next level's commitment over a folded (sharded) handle
        assert e.msm(f, sraw[:32 * n1]) == orc.msm(want_f, sraw[:32 * n1], mt=True)
    finally:
        for o in outs:
            o.free()
        h.free()


def test_elementwise_calls_shard_with_no_collective(multi2, orc):
    _elementwise_round(multi2, orc, 1001)
    _elementwise_round(multi2, orc, 2)


def test_forced_peer_copies_msm_and_elementwise(multi2, orc, monkeypatch):
    """BN256_FORCE_PEER_COPY=1: every cross-shard operand goes through hipMemcpyPeerAsync although both shards
    are device 0 -- the branch an 8-GPU node takes (multi_msm's device pointer on another GPU, shard_view's
    pieces), which `src_dev == c->device` otherwise skips on a one-GPU box."""
    monkeypatch.setenv("BN256_FORCE_PEER_COPY", "1")
    _elementwise_round(multi2, orc, 777)
    n = 5000
    h = multi2.random_bases(SEED_B, n)
    v = multi2.fr_random(SEED_S, n)
    try:
        assert multi2.msm_dev(h, v.device_ptr(), n) == orc.synth_msm_closed_form(SEED_B, SEED_S, n)
        off, m = 2400, 300                                     # a sub-range across the shard boundary (2500)
        assert multi2.msm_dev(h, v.device_ptr() + 32 * off, m, off=off) == orc.synth_msm_closed_form(SEED_B, SEED_S, m, first=off)
    finally:
        h.free()
        v.free()
    monkeypatch.delenv("BN256_FORCE_PEER_COPY")
    _elementwise_round(multi2, orc, 64)


def test_elementwise_lane_kernels_per_shard_match_the_single_device_bytes(multi2, eng):
    """2^17 outputs: 2^16 per shard, above the quad / lane switch (24 576): the same bytes as one device."""
    n = 1 << 17
    k, k2 = (1 << 254) - 99, 0xFEDCBA9876543210FEDCBA9876543210
    h, h1 = multi2.random_bases(SEED_B, n), eng.random_bases(SEED_B, n)
    outs = []
    try:
        a, b = multi2.vector_point_mul_on_scalar(h, k), eng.vector_point_mul_on_scalar(h1, k)
        outs += [a, b]
        assert a.download() == b.download()
        a, b = multi2.fold(h, 0, h, n // 2, n // 2, k, k2), eng.fold(h1, 0, h1, n // 2, n // 2, k, k2)
        outs += [a, b]
        assert a.download() == b.download()
        a, b = multi2.fold(h, 0, h, 1, n // 2, k2, k, stride=2), eng.fold(h1, 0, h1, 1, n // 2, k2, k, stride=2)
        outs += [a, b]
        assert a.download() == b.download()
    finally:
        for o in outs:
            o.free()
        h.free()
        h1.free()
