"""Randomised differential test of the element-wise entry points (rows a7-a9, the generator folds of
go/bp/main.go:355-356 and go/bppp/docs_test.go:1325-1326): sizes on both sides of every kernel switch
(2^13 outputs: GLV half-chains; 24576: quad / lane per output; 2^14: per-point scalar multiplication by
quads / lanes), offsets, strides, ragged lengths and scalars are drawn from a seeded generator.  The
inputs are the synthetic bases k_i * G0, so every output has the closed form (s_lo k_a + s_hi k_b) * G0;
a random sample of the outputs (and the first and last) is checked bit for bit."""
import random

import pytest

import crypto_amd

pytestmark = pytest.mark.gpu
N = 0x8FB501E34AA387F9AA6FECB86184DC212E8D8E12F82B39241A2EF45B57AC7261
SIZES = (1, 2, 3, 17, 100, 1000, 5000, 8191, 8192, 8193, 16383, 16384, 16385, 24575, 24576, 24577, 40000)


def _scalar(rng):
    return rng.choice((0, 1, 2, N - 1, N - 2, (N - 1) // 2, rng.randrange(1 << 20), rng.randrange(1 << 129),
                       rng.randrange(N), rng.randrange(N), rng.randrange(N)))


def _sample(rng, n, cnt=12):
    return sorted({0, n - 1} | {rng.randrange(n) for _ in range(min(cnt, n))}) if n else []


@pytest.mark.parametrize("seed", range(8))
def test_random_elementwise_calls(orc, seed):
    rng = random.Random(0xE1E + seed)
    e = crypto_amd.Engine(0)
    try:
        size = 100000
        seed_b = 0xE100 + seed
        g = e.random_bases(seed_b, size)
        ks = orc.synth_base_scalars(seed_b, size)

        def k(i):
            return int.from_bytes(ks[32 * i:32 * i + 32], "big")

        def check(h, want_scalar, n, tag):
            assert len(h) == n, tag
            for i in _sample(rng, n):
                got = h.download(i, 1)[0]
                w = want_scalar(i) % N
                assert got == (orc.g0_mul(w) if w else bytes(64)), f"{tag} output {i}"
            h.free()

        for case in range(10):
            kind = rng.choice(("fold", "fold2", "fold2_submit", "on_scalar", "each", "add"))
            n = rng.choice(SIZES)
            tag = f"seed {seed} case {case}: {kind} n {n}"
            if kind == "fold":
                stride = rng.choice((1, 2))
                lo, hi = rng.randrange(100), rng.randrange(100)
                n = min(n, (size - 100 - 1) // stride)
                s_lo, s_hi = _scalar(rng), _scalar(rng)
                check(e.fold(g, lo, g, hi, n, s_lo, s_hi, stride=stride),
                      lambda i: s_lo * k(lo + i * stride) + s_hi * k(hi + i * stride), n, tag)
            elif kind in ("fold2", "fold2_submit"):
                stride = rng.choice((1, 2))
                n_lo = min(n, (size - 100 - 1) // stride)
                n_hi = max(0, n_lo - rng.choice((0, 0, 1, 2, n_lo)))
                if rng.random() < 0.3:
                    n_lo, n_hi = n_hi, n_lo
                lo, hi = rng.randrange(100), rng.randrange(100)
                s_lo, s_hi = _scalar(rng), _scalar(rng)
                if kind == "fold2":
                    h = e.fold2(g, lo, n_lo, g, hi, n_hi, s_lo, s_hi, stride=stride)
                else:
                    h, t = e.fold2_submit(g, lo, n_lo, g, hi, n_hi, s_lo, s_hi, stride=stride)
                    e.fold_wait(t)
                check(h, lambda i: (s_lo * k(lo + i * stride) if i < n_lo else 0) +
                      (s_hi * k(hi + i * stride) if i < n_hi else 0), max(n_lo, n_hi), tag + f" ({n_lo}, {n_hi})")
            elif kind == "on_scalar":
                off, s = rng.randrange(100), _scalar(rng)
                check(e.vector_point_mul_on_scalar(g, s, off=off, n=n), lambda i: s * k(off + i), n, tag)
            elif kind == "each":
                off = rng.randrange(100)
                a = [_scalar(rng) for _ in range(n)]
                check(e.scalar_mul_each(g, a, off=off), lambda i: a[i] * k(off + i), n, tag)
            else:
                cand, boff = rng.choice((1, N - 1, 5)), rng.choice((0, 7))
                a = e.vector_point_mul_on_scalar(g, 1, off=0, n=n)
                b = e.vector_point_mul_on_scalar(g, cand, off=boff, n=n)
                check(e.hadamard_point_mul(a, b), lambda i: k(i) + cand * k(boff + i), n, tag)  # P + P and P - P included
                a.free()
                b.free()
        g.free()
    finally:
        e.close()
