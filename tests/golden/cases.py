"""Shared definition of the MSM parity cases (inputs only; no arithmetic).

A case is (name, n, seed_bases, seed_scalars, edits).  Inputs start from the synthetic
generators (base_i = synth_base(seed_bases, i), scalar_i = synth_scalar(seed_scalars, i)) and
are then edited to hit the edge cases the domain has: zero scalars, scalar 1,
scalar n-1, scalars >= n, duplicate bases, P and -P both present, infinity bases, and the
Bulletproofs-like 0/1 and 0/(n-1) distributions (go/bp/main.go:84-85).
Edits: ("scalar", i, int) | ("dup", i, j) base_i := base_j | ("negof", i, j) base_i := -base_j
       | ("inf", i) base_i := infinity | ("bits", lo, hi) scalars[lo:hi] := i & 1
       | ("bitsneg", lo, hi) scalars[lo:hi] := (i & 1) ? n-1 : 0
"""
N_ORDER = 0x8FB501E34AA387F9AA6FECB86184DC212E8D8E12F82B39241A2EF45B57AC7261
P_FIELD = 0x8FB501E34AA387F9AA6FECB86184DC21EE5B88D120B5B59E185CAC6C5E089667

CASES = [
    ("n1", 1, 0x5EED0001, 0x5EED0002, []),
    ("n1_zero", 1, 0x5EED0001, 0x5EED0002, [("scalar", 0, 0)]),
    ("n2_cancel", 2, 0x5EED0001, 0x5EED0002, [("negof", 1, 0), ("scalar", 0, 12345), ("scalar", 1, 12345)]),
    ("n2_double", 2, 0x5EED0001, 0x5EED0002, [("dup", 1, 0), ("scalar", 0, 7), ("scalar", 1, 7)]),
    ("n3", 3, 0x5EED0001, 0x5EED0002, [("scalar", 1, 1)]),
    ("n64_edges", 64, 0x5EED0001, 0x5EED0002,
     [("scalar", 0, 0), ("scalar", 1, 1), ("scalar", 2, N_ORDER - 1), ("scalar", 3, N_ORDER),
      ("scalar", 4, (1 << 256) - 1), ("scalar", 5, (N_ORDER - 1) // 2), ("scalar", 6, (N_ORDER + 1) // 2),
      ("dup", 8, 7), ("dup", 9, 7), ("negof", 10, 7), ("inf", 11), ("inf", 12), ("scalar", 12, 0),
      ("scalar", 13, 1 << 255), ("scalar", 14, (1 << 128) - 1), ("scalar", 15, 0x8000800080008000)]),
    ("n129", 129, 0x5EED0011, 0x5EED0012, []),
    ("n128_bulletproof", 128, 0x5EED0001, 0x5EED0002, [("bits", 0, 64), ("bitsneg", 64, 128)]),
    ("n1024", 1024, 0x5EED0001, 0x5EED0002, [("dup", 100, 99), ("negof", 200, 199), ("inf", 300)]),
    ("n1024_samebase", 1024, 0x5EED0001, 0x5EED0002, [("dup", i, 0) for i in range(1, 1024)]),
    ("n65536", 65536, 0x5EED0001, 0x5EED0002, []),
]


def neg_point_bytes(pt: bytes) -> bytes:
    if pt == bytes(64):
        return pt
    y = int.from_bytes(pt[32:], "big")
    return pt[:32] + ((P_FIELD - y) % P_FIELD).to_bytes(32, "big")


def apply_edits(bases: list, scalars: list, edits) -> None:
    """bases: list of 64-byte points, scalars: list of ints; edited in place."""
    for e in edits:
        if e[0] == "scalar":
            scalars[e[1]] = e[2]
        elif e[0] == "dup":
            bases[e[1]] = bases[e[2]]
        elif e[0] == "negof":
            bases[e[1]] = neg_point_bytes(bases[e[2]])
        elif e[0] == "inf":
            bases[e[1]] = bytes(64)
        elif e[0] == "bits":
            for i in range(e[1], e[2]):
                scalars[i] = i & 1
        elif e[0] == "bitsneg":
            for i in range(e[1], e[2]):
                scalars[i] = (N_ORDER - 1) if (i & 1) else 0
        else:
            raise ValueError(e)
