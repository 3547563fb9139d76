"""The wide-window fixed-base path (c = 20: 13 windows, 2^19 buckets, tile-local sort records, top table
level divided by top_scale, bucket sums carried across passes, fix-up by chunk boundaries, lane-per-
addition fold levels).  Results never depend on the window width: every case is checked against the
oracle's closed form over the synthetic bases (k_i * G0 with known k_i) and against a c = 16 context."""
import os

import pytest

import crypto_amd

pytestmark = pytest.mark.gpu
N = 0x8FB501E34AA387F9AA6FECB86184DC212E8D8E12F82B39241A2EF45B57AC7261
SEED_B = 0xC20


def _be(vals):
    return b"".join(int(v).to_bytes(32, "big") for v in vals)


@pytest.fixture(scope="module")
def wide():
    e = crypto_amd.Engine(0)
    e.set_tuning(20, 0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def table(wide):
    n = (1 << 18) + 37
    h = wide.random_bases(SEED_B, n, flags=crypto_amd.PRECOMPUTE)
    yield h, n
    h.free()


def _check(wide, orc, h, scalars_be, off=0, first=0):
    n = len(scalars_be) // 32
    v = wide.fr_upload(scalars_be)
    try:
        got = wide.msm_dev(h, v.device_ptr(), n, off=off)
    finally:
        v.free()
    assert got == orc.synth_msm_closed_form_scalars(SEED_B, scalars_be, first=first + off)
    return got


def test_uniform_subranges_and_edges(wide, orc, table, eng):
    h, n = table
    s = orc.synth_scalars(0xC21, n)
    got = _check(wide, orc, h, s)
    h16 = eng.random_bases(SEED_B, n, flags=crypto_amd.PRECOMPUTE)     # c = 16 table, same points
    assert eng.msm(h16, s) == got
    h16.free()
    # a sub-range that starts and ends inside sort tiles
    _check(wide, orc, h, s[32 * 1000:32 * (n - 777)], off=1000)
    # short calls on the wide table (global-atomic sort over 2^19 buckets, top digit scaled there too)
    _check(wide, orc, h, s[:32 * 1000])
    _check(wide, orc, h, s[:32], off=5)
    assert wide.msm(h, b"", n=0) == bytes(64)
    # edge scalars: 0, 1, n-1, >= n, the largest top-window digit ((n-1)/2 and its neighbour), 2^256-1
    edge = [0, 1, N - 1, N, N + 5, (N - 1) // 2, (N + 1) // 2, (1 << 256) - 1, 1 << 240, (1 << 240) - 1, 1 << 239, 1 << 19, (1 << 19) + 1]
    mixed = bytearray(s)
    for j, e in enumerate(edge * 50):
        mixed[32 * (7 * j):32 * (7 * j) + 32] = e.to_bytes(32, "big")
    _check(wide, orc, h, bytes(mixed))
    # strided view of the table and of the scalars
    v = wide.fr_upload(s)
    m = (1 << 17) + 3
    got = wide.msm_fr(h, 1, 2, m, v, 0, 2)
    ks = orc.synth_base_scalars(SEED_B, n)
    want = orc.g0_mul(orc.fr_dot(b"".join(s[64 * i:64 * i + 32] for i in range(m)),
                                 b"".join(ks[32 * (1 + 2 * i):32 * (2 + 2 * i)] for i in range(m))))
    assert got == want
    v.free()


def test_skewed_scalars_take_the_split_path_and_recover(wide, orc, table):
    import numpy as np
    h, n = table
    rng = np.random.default_rng(20)
    uniform = orc.synth_scalars(0xC22, n)
    # Bulletproofs-like bit vectors (go/bp/main.go:84-85): every non-zero digit is 1 -> one bucket, one group
    bits = rng.integers(0, 2, size=n, dtype=np.uint8)
    arr = np.zeros((n, 32), dtype=np.uint8)
    arr[:, 31] = bits
    arr[n // 2:][bits[n // 2:] == 0] = np.frombuffer((N - 1).to_bytes(32, "big"), dtype=np.uint8)
    _check(wide, orc, h, arr.tobytes())
    _check(wide, orc, h, uniform)                     # the split kernels stay on for this call ...
    _check(wide, orc, h, uniform)                     # ... and are dropped again for this one
    # digits confined to the lowest 1/16 of every window's range: 32 groups carry everything (all oversized)
    small = np.zeros((n, 32), dtype=np.uint8)
    vals = rng.integers(0, 1 << 15, size=(n, 12), dtype=np.int64)
    for i in range(12):  # window i holds bits 20 i .. 20 i + 14
        for byte in range(3):
            bitpos = 20 * i + 8 * byte
            chunk = ((vals[:, i] >> (8 * byte)) & 0xFF).astype(np.uint16)
            sh = bitpos % 8
            small[:, 31 - bitpos // 8] |= ((chunk << sh) & 0xFF).astype(np.uint8)
            small[:, 31 - bitpos // 8 - 1] |= ((chunk << sh) >> 8).astype(np.uint8)
    _check(wide, orc, h, small.tobytes())
    # a handful of distinct scalars (heavy buckets, with signs)
    few = [3, N - 3, 1 << 100, (1 << 200) + 12345, 0]
    _check(wide, orc, h, _be(few[i % 5] for i in range(n)))
    _check(wide, orc, h, uniform)


def test_uneven_tiles_take_pass_b_search_path(wide, orc, table):
    """Pass B (k_group_sort_tl) gets a record's tile back from six bits stored with it and a guess from its place
    in the group, e * tiles / m -- valid while every run of the group lies within 31 tiles of the guess, which it
    checks per run; otherwise every record of the group takes the binary search.  Zero scalars leave no records:
    a call whose first 70 % are zero has its first record ~90 tiles away from the guess, and one with a sparse
    first half and a dense second half drifts away from it gradually."""
    import numpy as np
    h, n = table
    s = bytearray(orc.synth_scalars(0xC2A, n))
    cut = (7 * n) // 10
    s[:32 * cut] = bytes(32 * cut)
    _check(wide, orc, h, bytes(s))
    arr = np.frombuffer(orc.synth_scalars(0xC2B, n), dtype=np.uint8).reshape(n, 32).copy()
    keep = np.ones(n, dtype=bool)
    keep[: n // 2] = (np.arange(n // 2) % 5) == 0
    arr[~keep] = 0
    _check(wide, orc, h, arr.tobytes())
    _check(wide, orc, h, orc.synth_scalars(0xC2C, n))  # and the tagged path again on the same context


def test_repeated_scalars_chains_and_heavy_queue(wide, orc, table):
    """k_fix_boundary joins up to four partial sums of a bucket per lane; buckets that span 5..32 chunks go
    to the medium queue (serial chains by k_heavy's quads), longer ones to the heavy queue.  Scalars drawn
    from d distinct values put 13 n / (13 d) entries into each of <= 13 d buckets: 1-2, ~11, ~55 and ~440
    chunks of 16 entries."""
    import numpy as np
    h, n = table
    rng = np.random.default_rng(21)
    for d in (40000, 1500, 300, 37):
        pool = [int.from_bytes(rng.bytes(32), "big") % N for _ in range(d)]
        pick = rng.integers(0, d, size=n)
        _check(wide, orc, h, _be(pool[i] for i in pick))
    # chained, heavy and ordinary buckets in one call
    pool = [int.from_bytes(rng.bytes(32), "big") % N for _ in range(1500)]
    s = bytearray(orc.synth_scalars(0xC28, n))
    for i in range(0, n // 2):
        s[32 * i:32 * i + 32] = pool[(i * 7) % (1500 if i % 3 else 20)].to_bytes(32, "big")
    _check(wide, orc, h, bytes(s))


def test_passes_share_one_bucket_array(orc, monkeypatch):
    """A call of several passes: later passes start every bucket from its current sum (k_accumulate's
    add_prev) and leave buckets without entries alone.  Passes of 2^18 points; the third pass alone is
    skewed, so the oversized-group flag comes up in the middle of a call and the whole call is repeated."""
    import numpy as np
    monkeypatch.setenv("BN256_MSM_CHUNK_LOG2", "18")
    e = crypto_amd.Engine(0)
    monkeypatch.delenv("BN256_MSM_CHUNK_LOG2")
    try:
        for c in (20, 16):
            e.set_tuning(c, 0)
            n = (1 << 20) + 5
            h = e.random_bases(SEED_B, n, flags=crypto_amd.PRECOMPUTE)
            s = bytearray(orc.synth_scalars(0xC23, n))
            v = e.fr_upload(bytes(s))
            assert e.msm_dev(h, v.device_ptr(), n) == orc.synth_msm_closed_form_scalars(SEED_B, bytes(s))
            v.free()
            lo, hi = 2 << 18, 3 << 18
            s[32 * lo:32 * hi] = _be([1] * (hi - lo))        # third pass: one bucket
            s[32 * (3 << 18):32 * (4 << 18)] = bytes(32 * (1 << 18))  # fourth pass: no entries at all
            v = e.fr_upload(bytes(s))
            want = orc.synth_msm_closed_form_scalars(SEED_B, bytes(s))
            assert e.msm_dev(h, v.device_ptr(), n) == want
            assert e.msm_dev(h, v.device_ptr(), n) == want
            t = e.msm_submit(h, v.device_ptr(), n)
            assert e.msm_wait(t) == want
            v.free()
            h.free()
    finally:
        e.close()


def test_uploaded_points_with_infinity_and_two_tables_in_one_call(wide, orc):
    n = (1 << 18) + 5
    ks = bytearray(orc.synth_base_scalars(SEED_B, n))
    braw = bytearray(orc.synth_bases(SEED_B, n))
    for i in (0, 77, 2048, n - 1):                    # points at infinity: their scalars must not matter
        braw[64 * i:64 * i + 64] = bytes(64)
        ks[32 * i:32 * i + 32] = bytes(32)
    h = wide.upload(bytes(braw), crypto_amd.PRECOMPUTE)
    s = orc.synth_scalars(0xC24, n)
    v = wide.fr_upload(s)
    assert wide.msm_dev(h, v.device_ptr(), n) == orc.g0_mul(orc.fr_dot(s, bytes(ks)))
    # the BP++ commitment shape over two wide tables: one pipeline, two terms (go/bppp/docs_test.go:560-564)
    h2 = wide.random_bases(0xC25, n, flags=crypto_amd.PRECOMPUTE)
    s2 = orc.synth_scalars(0xC26, n - 9)
    v2 = wide.fr_upload(s2)
    got = wide.msm_fr_multi([(h, 0, 1, n, v, 0, 1), (h2, 9, 1, n - 9, v2, 0, 1)])
    k2 = orc.synth_base_scalars(0xC25, n)
    want = orc.g0_mul(orc.fr_dot(s, bytes(ks)) + orc.fr_dot(s2, k2[32 * 9:]))
    assert got == want
    for x in (h, h2, v, v2):
        x.free()


def test_default_window_choice_at_2_20_and_2_21(eng, orc):
    for log2n in (20, 21):
        n = 1 << log2n
        h = eng.random_bases(SEED_B, n, flags=crypto_amd.PRECOMPUTE)
        v = eng.fr_random(0xC27, n)
        assert eng.msm_dev(h, v.device_ptr(), n) == orc.synth_msm_closed_form(SEED_B, 0xC27, n)
        m = n - 12345
        assert eng.msm_dev(h, v.device_ptr() + 32 * 45, m, off=45) == orc.synth_msm_closed_form(SEED_B, 0xC27, m, first=45)
        h.free()
        v.free()


def test_pass_lengths_between_2_20_and_9_8ths(eng, orc):
    """A wide pass may hold up to 9/8 x 2^20 points (2^20 + 9 is the BP++ commitments' HVec, one pass);
    above that the passes are balanced.  The index field of the sort's registers holds 21 bits: a 20-bit
    field sent indices >= 2^20 to the next table level (found as wrong sums and an out-of-bounds gather)."""
    for n in ((1 << 20) + 9, (1 << 20) + 70000, (9 << 17), (9 << 17) + 1):
        h = eng.random_bases(SEED_B, n, flags=crypto_amd.PRECOMPUTE)
        s = bytearray(orc.synth_scalars(0xC29, n))
        for skew in (False, True):
            if skew:  # a quarter of the scalars in one bucket (oversized group, split sort, heavy bucket)
                s[32 * (2 << 18):32 * (3 << 18)] = _be([1] * (1 << 18))
            v = eng.fr_upload(bytes(s))
            want = orc.synth_msm_closed_form_scalars(SEED_B, bytes(s))
            assert eng.msm_dev(h, v.device_ptr(), n) == want
            assert eng.msm_dev(h, v.device_ptr(), n) == want
            v.free()
        h.free()
