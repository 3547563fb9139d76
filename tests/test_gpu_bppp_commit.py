"""BASELINE.json configs[3]: the Bulletproofs++ vector commitments Co, Cl, Cr (CommitOL / CommitR,
go/bppp/docs_test.go:560-564,588-589) and Cs (:1002-1003) -- <HVec[0:9+Nv], r||l> + <GVec[0:Nm], n> --
as ONE device pipeline over two fixed-base tables with resident scalars (bn256_msm_fr_multi), at
fixture size against the oracle's term loop and at Nv = Nm = 2^20 against the closed form."""
import pytest

import crypto_amd

pytestmark = pytest.mark.gpu
SEED_H, SEED_G = 0xB9901, 0xB9902


def _closed(orc, nh, ng, seed_rl, seed_n):
    k = (orc.synth_dot(SEED_H, seed_rl, nh) + orc.synth_dot(SEED_G, seed_n, ng)) % 0x8FB501E34AA387F9AA6FECB86184DC212E8D8E12F82B39241A2EF45B57AC7261
    return orc.g0_mul(k)


def test_commit_shape_small_vs_oracle_term_loop(eng, orc):
    nv, nm = 300, 200
    hraw, graw = orc.synth_bases(SEED_H, 9 + nv), orc.synth_bases(SEED_G, nm)
    rl, nn = orc.synth_scalars(0x11, 9 + nv), orc.synth_scalars(0x12, nm)
    want = orc.msm(hraw + graw, rl + nn)
    for flags in (0, crypto_amd.PRECOMPUTE):
        H, G = eng.upload(hraw, flags), eng.upload(graw, flags)
        vrl, vn = eng.fr_upload(rl), eng.fr_upload(nn)
        assert eng.msm_fr_multi([(H, 0, 1, 9 + nv, vrl, 0, 1), (G, 0, 1, nm, vn, 0, 1)]) == want
        # host-scalar form of the same call, and Com's H[9:] segment (go/bppp/docs_test.go:514-519)
        sc = lambda raw: [int.from_bytes(raw[32 * i:32 * i + 32], "big") for i in range(len(raw) // 32)]  # noqa: E731
        assert eng.msm_multi([(H, 0, sc(rl)), (G, 0, sc(nn))]) == want
        got = eng.msm_fr_multi([(H, 9, 1, nv, vrl, 9, 1)])
        assert got == orc.msm(hraw[64 * 9:], rl[32 * 9:])
        # strided views of tables in one pipeline (reducePoints halves) and a scaled segment (gathered path)
        got = eng.msm_fr_multi([(H, 1, 2, 100, vrl, 0, 2), (G, 0, 2, 100, vn, 1, 2)])
        hs = b"".join(hraw[64 * (1 + 2 * i):64 * (2 + 2 * i)] for i in range(100)) + b"".join(graw[128 * i:128 * i + 64] for i in range(100))
        ss = b"".join(rl[64 * i:64 * i + 32] for i in range(100)) + b"".join(nn[32 * (1 + 2 * i):32 * (2 + 2 * i)] for i in range(100))
        assert got == orc.msm(hs, ss)
        for x in (H, G, vrl, vn):
            x.free()


def test_commitments_2_20_closed_form_four_in_a_row(eng, orc):
    nv = nm = 1 << 20
    H = eng.random_bases(SEED_H, 9 + nv, flags=crypto_amd.PRECOMPUTE)
    G = eng.random_bases(SEED_G, nm, flags=crypto_amd.PRECOMPUTE)
    seeds = [(0x21, 0x22), (0x23, 0x24), (0x25, 0x26), (0x27, 0x28)]     # (r||l, n) of Co, Cl, Cr, Cs
    vec = [(eng.fr_random(a, 9 + nv), eng.fr_random(b, nm)) for a, b in seeds]
    try:
        want = [_closed(orc, 9 + nv, nm, a, b) for a, b in seeds]
        segs = lambda i: [(H, 0, 1, 9 + nv, vec[i][0], 0, 1), (G, 0, 1, nm, vec[i][1], 0, 1)]  # noqa: E731
        assert eng.msm_fr_multi(segs(0)) == want[0]
        # the four commitments need nothing of each other: three lanes, the fourth behind the first wait
        t = [eng.msm_fr_multi_submit(segs(i)) for i in range(3)]
        got = [eng.msm_wait(t[0])]
        t.append(eng.msm_fr_multi_submit(segs(3)))
        got += [eng.msm_wait(x) for x in t[1:]]
        assert got == want
        # the pad / truncate rule of go/bppp/util.go:178-196 lives in the caller: a shorter n-vector is
        # a shorter segment
        short = 12345
        assert eng.msm_fr_multi([(H, 0, 1, 9 + nv, vec[0][0], 0, 1), (G, 0, 1, short, vec[0][1], 0, 1)]) == \
            orc.g0_mul(orc.synth_dot(SEED_H, seeds[0][0], 9 + nv) + orc.synth_dot(SEED_G, seeds[0][1], short))
    finally:
        for a, b in vec:
            a.free()
            b.free()
        H.free()
        G.free()
