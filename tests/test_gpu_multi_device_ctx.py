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
            e.vector_point_mul_on_scalar(h, 5)
        assert ei.value.code == -10
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
