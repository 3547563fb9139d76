"""Randomised differential test of the MSM entry points: table size, window width, chunk length, call
kind (host scalars, device pointer, strided resident vector, several segments, submit / wait), range and
scalar distribution are drawn from a seeded generator; every result is compared bit for bit with the
oracle's closed form over the synthetic bases (k_i * G0 with known k_i: sum_i a_i k_i mod n times G0).
Shapes no hand-written case covers are what this is for: the 20-bit index field that sent indices
>= 2^20 to the next table level was of that kind."""
import random

import pytest

import crypto_amd

pytestmark = pytest.mark.gpu
N = 0x8FB501E34AA387F9AA6FECB86184DC212E8D8E12F82B39241A2EF45B57AC7261


def _scalars(rng, kind, n):
    if kind == "uniform":
        return [rng.randrange(1 << 256) for _ in range(n)]            # also >= n: reduced like Mod(Order)
    if kind == "small":
        return [rng.randrange(1 << rng.choice((1, 8, 20, 40))) for _ in range(n)]
    if kind == "few":
        pool = [rng.randrange(N) for _ in range(rng.choice((1, 3, 17)))] + [0, 1, N - 1]
        return [rng.choice(pool) for _ in range(n)]
    if kind == "bits":                                               # aL / aR of go/bp/main.go:84-85
        return [rng.choice((0, 1, N - 1)) for _ in range(n)]
    edge = [0, 1, 2, N - 1, N, N + 1, (N - 1) // 2, (N + 1) // 2, (1 << 256) - 1, 1 << 255, 1 << 240, (1 << 240) - 1]
    return [rng.choice(edge) if rng.random() < 0.5 else rng.randrange(N) for _ in range(n)]


def _be(vals):
    return b"".join(int(v).to_bytes(32, "big") for v in vals)


@pytest.mark.parametrize("seed", range(16))
def test_random_calls_against_the_closed_form(orc, seed):
    rng = random.Random(0xC0FFEE + seed)
    e = crypto_amd.Engine(0)
    try:
        for case in range(14):
            table = rng.random() < 0.6
            # mostly small and medium tables; now and then one that takes the 20-bit-window path and several passes
            size = int(2 ** rng.uniform(0, 18.3 if rng.random() < 0.85 else 20.4)) + rng.randrange(3)
            seed_b = 0xAB00 + 100 * seed + case
            c = rng.choice((0, 0, 0, 4, 7, 10, 13, 16, 18) + ((20,) if table else ()))
            k = rng.choice((0, 0, 2, 8, 64))
            e.set_tuning(c, k)
            h = e.random_bases(seed_b, size, flags=crypto_amd.PRECOMPUTE if table else 0)
            ks = orc.synth_base_scalars(seed_b, size)

            def kint(i):
                return int.from_bytes(ks[32 * i:32 * i + 32], "big")

            def dot(a, off):  # sum_i a_i k_{off+i} mod n, in C for long ranges
                return orc.fr_dot(_be(x % N for x in a), ks[32 * off:32 * (off + len(a))]) if a else 0
            kind = rng.choice(("host", "dev", "fr", "multi", "submit"))
            dist = rng.choice(("uniform", "small", "few", "bits", "edge"))
            tag = f"seed {seed} case {case}: size {size} table {table} c {c} K {k} {kind} {dist}"
            if kind == "fr":
                stride, vstride = rng.choice((1, 1, 2, 3)), rng.choice((1, 1, 2))
                off = rng.randrange(size)
                n = rng.randrange(0, (size - off - 1) // stride + 2)
                n = min(n, (size - off - 1) // stride + 1, 50000)
                voff = rng.randrange(4)
                a = _scalars(rng, dist, voff + max(n, 1) * vstride)
                v = e.fr_upload(_be(x % N for x in a))               # resident vectors hold reduced scalars
                got = e.msm_fr(h, off, stride, n, v, voff, vstride)
                want = sum(a[voff + i * vstride] * kint(off + i * stride) for i in range(n)) % N
                v.free()
            elif kind == "multi":
                segs, want = [], 0
                for _ in range(rng.choice((1, 2, 3, 5))):
                    off = rng.randrange(size)
                    n = rng.randrange(0, min(size - off, 3000) + 1)
                    a = _scalars(rng, dist, n)
                    segs.append((h, off, a))
                    want += dot(a, off)
                got = e.msm_multi([(hh, off, [x % (1 << 256) for x in a]) for hh, off, a in segs])
                want %= N
            else:
                off = rng.randrange(size)
                n = rng.randrange(0, size - off + 1)
                a = _scalars(rng, dist, n)
                raw = _be(a)
                if kind == "host":
                    got = e.msm(h, raw, off=off, n=n)
                else:
                    v = e.fr_upload(_be(x % N for x in a)) if n else None
                    ptr = v.device_ptr() if n else 0
                    if kind == "dev":
                        got = e.msm_dev(h, ptr, n, off=off)
                    else:
                        t = e.msm_submit(h, ptr, n, off=off)
                        got = e.msm_wait(t)
                    if v:
                        v.free()
                want = dot(a, off)
            assert got == orc.g0_mul(want), tag
            h.free()
    finally:
        e.close()


@pytest.mark.parametrize("seed", range(3))
def test_random_calls_on_a_multi_device_context(orc, seed):
    """The same against bn256_ctx_create_multi (two or three shards on the box's one device): ranges that
    start, end or lie inside any shard, host scalars, device pointer, resident vector."""
    rng = random.Random(0xD0D0 + seed)
    e = crypto_amd.Engine(devices=[0] * rng.choice((2, 3)))
    try:
        for case in range(10):
            table = rng.random() < 0.5
            size = int(2 ** rng.uniform(0, 17)) + rng.randrange(3)
            seed_b = 0xCD00 + 100 * seed + case
            h = e.random_bases(seed_b, size, flags=crypto_amd.PRECOMPUTE if table else 0)
            ks = orc.synth_base_scalars(seed_b, size)
            off = rng.randrange(size)
            n = rng.randrange(0, size - off + 1)
            a = _scalars(rng, rng.choice(("uniform", "few", "bits", "edge")), n)
            kind = rng.choice(("host", "dev", "fr"))
            if kind == "host":
                got = e.msm(h, _be(a), off=off, n=n)
            else:
                v = e.fr_upload(_be(x % N for x in a)) if n else None
                if kind == "dev":
                    got = e.msm_dev(h, v.device_ptr() if n else 0, n, off=off)
                else:
                    got = e.msm_fr(h, off, 1, n, v) if n else bytes(64)
                if v:
                    v.free()
            want = orc.fr_dot(_be(x % N for x in a), ks[32 * off:32 * (off + n)]) if n else 0
            assert got == orc.g0_mul(want), f"seed {seed} case {case}: size {size} table {table} {kind} off {off} n {n}"
            h.free()
    finally:
        e.close()


@pytest.mark.parametrize("table", [True, False])
def test_sizes_around_the_two_pass_sort_thresholds(orc, table):
    """The two-pass LDS sort takes over from 2^15 points with a table and from 2^17 without
    (msm_enqueue_srcs, two_min): sizes on both sides of the switch, uniform and skewed scalars (the aL / aR
    distribution puts nearly every record into one group: the oversized flag and the repeat with the split
    kernels at the smallest sizes the path now sees)."""
    rng = random.Random(0x2BA5 + table)
    e = crypto_amd.Engine(0)
    try:
        edge = 1 << (15 if table else 17)
        size = edge + 4097
        seed_b = 0x2BA50 + table
        h = e.random_bases(seed_b, size, flags=crypto_amd.PRECOMPUTE if table else 0)
        ks = orc.synth_base_scalars(seed_b, size)
        for n in (edge - 1, edge, edge + 1, edge + 2049, edge + 4097):
            for dist in ("uniform", "bits", "few"):
                a = _scalars(rng, dist, n)
                v = e.fr_upload(_be(x % N for x in a))
                got = e.msm_dev(h, v.device_ptr(), n)
                v.free()
                want = orc.fr_dot(_be(x % N for x in a), ks[:32 * n])
                assert got == orc.g0_mul(want), f"table {table} n {n} {dist}"
        h.free()
    finally:
        e.close()
