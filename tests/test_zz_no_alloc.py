"""bn256_msm_dev_noalloc (include/bn256msm.h): the MSM pipeline over the library's static device
array instead of hipMalloc'd workspaces.  It has no counterpart in the reference; the bar is the
existing entry point it is a variant of: identical bytes to bn256_msm_dev on the same inputs, on two
calls in a row (the second smaller, so it runs over the first call's leftovers), plus the closed
form of the synthetic workload and the size limit."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

SEED_B, SEED_S = 0x5EED0001, 0x5EED0002


def _pair(eng, h, v, n, off=0, voff=0):
    ptr = v.device_ptr() + 32 * voff
    return eng.msm_dev_noalloc(h, ptr, n, off=off), eng.msm_dev(h, ptr, n, off=off)


def test_noalloc_matches_msm_dev_twice_second_smaller(eng, orc):
    import crypto_amd
    n = 1 << 20
    h = eng.random_bases(SEED_B, n, flags=crypto_amd.PRECOMPUTE)
    v = eng.fr_random(SEED_S, n)
    try:
        got, want = _pair(eng, h, v, n)
        assert got == want
        assert got == orc.synth_msm_closed_form(SEED_B, SEED_S, n)
        # second call, smaller, over a sub-range that starts inside a sort tile: the array still holds
        # the first call's records, offsets, bucket and partial sums
        m, off = 70001, 4099
        got2, want2 = _pair(eng, h, v, m, off=off, voff=off)
        assert got2 == want2 and got2 != got
        # and the first one again: a smaller call in between leaves nothing behind that matters
        assert eng.msm_dev_noalloc(h, v.device_ptr(), n) == want
        # the allocating entry points are unaffected by the swap of workspaces
        assert eng.msm_dev(h, v.device_ptr(), n) == want
        assert eng.msm_dev_noalloc(h, v.device_ptr(), 0) == bytes(64)
        # skewed scalars (every entry in one bucket): the call is repeated with the split-sort kernels,
        # whose workspace comes out of the same array
        ones = eng.fr_upload((1).to_bytes(32, "big") * n)
        got3, want3 = _pair(eng, h, ones, n)
        assert got3 == want3 == orc.synth_msm_closed_form_scalars(SEED_B, (1).to_bytes(32, "big") * n)
        ones.free()
        assert eng.msm_dev_noalloc(h, v.device_ptr(), n) == want
    finally:
        h.free()
        v.free()


def test_noalloc_small_and_tableless_against_oracle(eng, orc):
    n = 3001
    braw = orc.synth_bases(SEED_B, n)
    sraw = orc.synth_scalars(SEED_S, n)
    want = orc.msm(braw, sraw, mt=True)
    h = eng.upload(braw)
    v = eng.fr_upload(sraw)
    try:
        assert eng.msm_dev_noalloc(h, v.device_ptr(), n) == want
        assert eng.msm_dev_noalloc(h, v.device_ptr(), 17) == orc.msm(braw[:64 * 17], sraw[:32 * 17])
    finally:
        h.free()
        v.free()
    n = 1 << 20  # the largest table-less call the array holds (16 bucket sets of 2^15)
    h = eng.random_bases(SEED_B, n)
    v = eng.fr_random(SEED_S, n)
    try:
        got, want = _pair(eng, h, v, n)
        assert got == want == orc.synth_msm_closed_form(SEED_B, SEED_S, n)
    finally:
        h.free()
        v.free()


def test_noalloc_rejects_sizes_beyond_the_array(eng, orc):
    import crypto_amd
    from crypto_amd.engine import BN256Error
    n = 1 << 21   # table-less: runs as two passes of 2^20 points, whose workspaces fit the array
    h = eng.random_bases(SEED_B, n)
    v = eng.fr_random(SEED_S, n)
    try:
        got, want = _pair(eng, h, v, n)
        assert got == want == orc.synth_msm_closed_form(SEED_B, SEED_S, n)
        # 18-bit windows without a table: 15 sets of 2^17 buckets = 283 MiB of bucket sums + 126 MiB of entries
        eng.set_tuning(window_bits=18)
        try:
            with pytest.raises(BN256Error) as ei:
                eng.msm_dev_noalloc(h, v.device_ptr(), 1 << 20)
            assert ei.value.code == -9 and "scratch" in str(ei.value)
        finally:
            eng.set_tuning()
        # the context is as usable as before, on both entry points
        m = 1 << 16
        assert eng.msm_dev_noalloc(h, v.device_ptr(), m) == eng.msm_dev(h, v.device_ptr(), m)
    finally:
        h.free()
        v.free()
