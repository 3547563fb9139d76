"""CPU restatement (big-int Python) of the reference's BN256-G1 commitment path.

TEST INFRASTRUCTURE ONLY.  Nothing under crypto_amd/ may import this module; it is
used by tests/, by tests/golden/make_golden.py (fixture generator) and by
__graft_entry__.smoke() as the checker.  It never runs on the product path.

PARITY STATUS: *unpinned for G1 points*.  The group arithmetic of the reference lives
in the un-vendored Go module github.com/cloudflare/bn256
v0.0.0-20230823183853-e5d9e5b354ab (/root/reference/go.mod:6) which is absent from
this container, and no test under /root/reference holds a literal G1 point
(SURVEY.md section 8c).  What *is* pinned by the reference: the group order n
(go/bppp/docs_test.go:365 prints n-4, n-1, (n-1)/2 ...) and scalar-helper outputs
(go/bppp/util_test.go comments, go/bp/docs_test.go:392).  The point arithmetic below
restates the published definition of the curve (y^2 = x^3 + 3 over p, generator
(1, -2), BN parameter u = 6518589491078791937) with the textbook affine chord-tangent
law, i.e. the same formulas as go/ec/main.go:63-87 plus the P = +-Q / infinity cases
that file omits.  Because normalised affine coordinates are unique, any correct
implementation of the group law is byte-identical with the reference's Marshal().

Each function cites the reference file:line it follows.
"""
from __future__ import annotations

# --------------------------------------------------------------------------------------
# Curve constants.  u from cloudflare/bn256 constants.go (published); p, n derived.
# --------------------------------------------------------------------------------------
U = 6518589491078791937
P = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
N = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1          # bn256.Order
B = 3
G0 = (1, P - 2)                                             # curveGen = (1, -2)
INF = None

assert P == 0x8FB501E34AA387F9AA6FECB86184DC21EE5B88D120B5B59E185CAC6C5E089667
assert N == 0x8FB501E34AA387F9AA6FECB86184DC212E8D8E12F82B39241A2EF45B57AC7261
# go/bppp/docs_test.go:365 prints these residues; they pin n.
assert N - 4 == 65000549695646603732796438742359905742570406053903786389881062969044166799965
assert (N - 1) // 2 == 32500274847823301866398219371179952871285203026951893194940531484522083399984


# --------------------------------------------------------------------------------------
# Group law (affine, chord-tangent).  Shape of go/ec/main.go:63-87, with edge cases.
# --------------------------------------------------------------------------------------
def is_on_curve(pt) -> bool:
    if pt is INF:
        return True
    x, y = pt
    return (y * y - (x * x * x + B)) % P == 0


def neg(pt):
    if pt is INF:
        return INF
    x, y = pt
    return (x, (-y) % P)


def add(p1, p2):
    """bn256.G1.Add: handles infinity, P == Q (doubling) and P == -Q."""
    if p1 is INF:
        return p2
    if p2 is INF:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P          # go/ec/main.go:78-87
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P             # go/ec/main.go:63-76
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def double(pt):
    return add(pt, pt)


# Jacobian helpers: only to make the restatement fast enough for 2^10..2^16 fixtures.
def _jac_double(X, Y, Z):
    if Z == 0 or Y == 0:
        return (1, 1, 0)
    A = X * X % P
    Bq = Y * Y % P
    C = Bq * Bq % P
    D = 2 * ((X + Bq) * (X + Bq) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_add_affine(X1, Y1, Z1, x2, y2):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % P
    U2 = x2 * Z1Z1 % P
    S2 = y2 * Z1 * Z1Z1 % P
    H = (U2 - X1) % P
    R = (S2 - Y1) % P
    if H == 0:
        if R == 0:
            return _jac_double(X1, Y1, Z1)
        return (1, 1, 0)
    HH = H * H % P
    HHH = H * HH % P
    V = X1 * HH % P
    X3 = (R * R - HHH - 2 * V) % P
    Y3 = (R * (V - X3) - Y1 * HHH) % P
    Z3 = Z1 * H % P
    return (X3, Y3, Z3)


def _jac_to_affine(X, Y, Z):
    if Z == 0:
        return INF
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def mul(k: int, pt):
    """bn256.G1.ScalarMult(pt, k): k.pt by MSB-first double-and-add.  k = 0 -> infinity."""
    if k < 0:
        raise ValueError("negative scalar")
    if pt is INF or k == 0:
        return INF
    x, y = pt
    acc = (1, 1, 0)
    for bit in bin(k)[2:]:
        acc = _jac_double(*acc)
        if bit == "1":
            acc = _jac_add_affine(*acc, x, y)
    return _jac_to_affine(*acc)


def mul_affine_slow(k: int, pt):
    """Same as mul() using only the affine law above (used to cross-check mul())."""
    acc = INF
    for bit in bin(k)[2:] if k else "":
        acc = double(acc)
        if bit == "1":
            acc = add(acc, pt)
    return acc


def base_mul(k: int):
    """bn256.G1.ScalarBaseMult(k)."""
    return mul(k % N if k >= N else k, G0)


# --------------------------------------------------------------------------------------
# Wire format: G1.Marshal()/Unmarshal() -- 64 bytes, x||y big-endian, infinity = zeros.
# Consumed by hash() at go/bp/util.go:228-230 and compared at go/bp/main.go:220,278.
# --------------------------------------------------------------------------------------
def marshal(pt) -> bytes:
    if pt is INF:
        return bytes(64)
    return pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def unmarshal(b: bytes):
    if len(b) != 64:
        raise ValueError("bn256: not enough data")
    x = int.from_bytes(b[:32], "big")
    y = int.from_bytes(b[32:], "big")
    if x >= P or y >= P:
        raise ValueError("bn256: coordinate exceeds modulus")
    if x == 0 and y == 0:
        return INF
    if not is_on_curve((x, y)):
        raise ValueError("bn256: malformed point")
    return (x, y)


def scalar_to_32(s: int) -> bytes:
    """scalarTo32Byte / to32Bytes, go/bp/util.go:209-220."""
    raw = s.to_bytes((s.bit_length() + 7) // 8, "big")
    if len(raw) >= 32:
        return raw[:32]
    return bytes(32 - len(raw)) + raw


# --------------------------------------------------------------------------------------
# The hot path: vectorPointScalarMul and its callers.
# --------------------------------------------------------------------------------------
def vector_point_scalar_mul(g, a):
    """go/bp/util.go:165-175 -- serial ScalarMult + Add loop; panics on length mismatch."""
    if len(g) != len(a):
        raise ValueError("invalid length for scalar mul")
    res = mul(a[0], g[0])
    for i in range(1, len(g)):
        res = add(res, mul(a[i], g[i]))
    return res


def vector_point_scalar_mul_bppp(g, a):
    """go/bppp/util.go:178-196 -- empty g -> identity; short a zero-padded; long a truncated."""
    if len(g) == 0:
        return INF
    a = list(a) + [0] * max(0, len(g) - len(a))
    res = mul(a[0], g[0])
    for i in range(1, len(g)):
        res = add(res, mul(a[i], g[i]))
    return res


def msm_pippenger(g, a, c: int = 8):
    """Algorithm-independent cross-check of vector_point_scalar_mul (bucket method).

    Not a restatement of anything in the reference; it exists so that large fixtures can
    be generated quickly and so that the reference-loop result has a second witness."""
    if len(g) != len(a):
        raise ValueError("invalid length for scalar mul")
    nwin = (256 + c - 1) // c
    total = (1, 1, 0)
    for w in reversed(range(nwin)):
        for _ in range(c):
            total = _jac_double(*total)
        buckets = [None] * (1 << c)
        for pt, k in zip(g, a):
            if pt is INF:
                continue
            d = (k >> (w * c)) & ((1 << c) - 1)
            if d:
                cur = buckets[d]
                buckets[d] = (pt[0], pt[1], 1) if cur is None else _jac_add_affine(*cur, *pt)
        run = INF
        acc = INF
        for d in range((1 << c) - 1, 0, -1):
            if buckets[d] is not None:
                run = add(run, _jac_to_affine(*buckets[d]))
            acc = add(acc, run)
        if acc is not INF:
            tot_aff = _jac_to_affine(*total)
            s = add(tot_aff, acc)
            total = (1, 1, 0) if s is INF else (s[0], s[1], 1)
    return _jac_to_affine(*total)


def com(g, h, a, b):
    """go/bp/util.go:92-94."""
    return add(mul(a, g), mul(b, h))


def vec_com(g, h, a, b):
    """go/bp/util.go:96-98."""
    return add(vector_point_scalar_mul(g, a), vector_point_scalar_mul(h, b))


def product_com(g, h, u, a, b):
    """go/bp/util.go:100-104."""
    p = vec_com(g, h, a, b)
    return add(p, mul(vector_mul(a, b), u))


def inner_product_h(n, g, h, u, a, a1, b, b1, c):
    """go/bp/util.go:15-27."""
    if n % 2 != 0:
        raise ValueError("invalid n")
    n1 = n // 2
    res = vector_point_scalar_mul(g[:n1], a)
    res = add(res, vector_point_scalar_mul(g[n1:], a1))
    res = add(res, vector_point_scalar_mul(h[:n1], b))
    res = add(res, vector_point_scalar_mul(h[n1:], b1))
    return add(res, mul(c, u))


def vector_point_mul_on_scalar(g, a):
    """go/bp/util.go:177-183."""
    return [mul(a, gi) for gi in g]


def hadamard_point_mul(a, b):
    """go/bp/util.go:185-195."""
    if len(a) != len(b):
        raise ValueError("invalid length for scalar mul")
    return [add(x, y) for x, y in zip(a, b)]


def vector_points_add_bppp(a, b):
    """go/bppp/util.go:279-293 -- pads the shorter side with infinity."""
    a = list(a) + [INF] * max(0, len(b) - len(a))
    b = list(b) + [INF] * max(0, len(a) - len(b))
    return [add(x, y) for x, y in zip(a, b)]


# --------------------------------------------------------------------------------------
# Scalar (Fr) helpers, go/bp/util.go:29-49,106-163,197-207.
# --------------------------------------------------------------------------------------
def fadd(x, y):
    return (x + y) % N


def fsub(x, y):
    return (x - y) % N


def fmul(x, y):
    return (x * y) % N


def finv(x):
    return pow(x, -1, N)


def ntharr(val, n):
    res = [1] * n
    for i in range(1, n):
        res[i] = fmul(res[i - 1], val)
    return res


def invntharr(val, n):
    return ntharr(finv(val), n)


def vector_add(a, b):
    if len(a) != len(b):
        raise ValueError("invalid length")
    return [fadd(x, y) for x, y in zip(a, b)]


def vector_sub(a, b):
    if len(a) != len(b):
        raise ValueError("invalid length")
    return [fsub(x, y) for x, y in zip(a, b)]


def vector_mul_on_scalar(a, c):
    return [fmul(x, c) for x in a]


def vector_mul(a, b):
    if len(a) != len(b):
        raise ValueError("invalid length")
    res = 0
    for x, y in zip(a, b):
        res = fadd(res, fmul(x, y))
    return res


def hadamard_mul(a, b):
    if len(a) != len(b):
        raise ValueError("invalid length")
    return [fmul(x, y) for x, y in zip(a, b)]


def to_bits(v, n):
    """go/bp/util.go:235-251 (little-endian bits, padded/truncated to n)."""
    res = [(v >> i) & 1 for i in range(max(v.bit_length(), 1))]
    res += [0] * max(0, n - len(res))
    return res[:n]


# BP++ variants (go/bppp/util.go): pad-tolerant vector ops.
def _pad2(a, b):
    a = list(a) + [0] * max(0, len(b) - len(a))
    b = list(b) + [0] * max(0, len(a) - len(b))
    return a, b


def vector_mul_bppp(a, b):
    """go/bppp/util.go:143-157."""
    a, b = _pad2(a, b)
    return vector_mul(a, b)


def weight_vector_mul(a, b, mu):
    """go/bppp/util.go:159-176."""
    a, b = _pad2(a, b)
    res, e = 0, mu % N
    for x, y in zip(a, b):
        res = fadd(res, fmul(fmul(x, y), e))
        e = fmul(e, mu)
    return res


def vector_add_bppp(a, b):
    a, b = _pad2(a, b)
    return vector_add(a, b)


def vector_sub_bppp(a, b):
    a, b = _pad2(a, b)
    return vector_sub(a, b)


def vector_tensor_mul(a, b):
    """go/bppp/util.go:134-141."""
    res = []
    for y in b:
        res += vector_mul_on_scalar(a, y)
    return res


def powvector(v, n):
    """go/bppp/util.go:54-63."""
    return ntharr(v, n) if n else []


def reduce_vector(v):
    """go/bppp/util.go:198-211 -- even / odd index split."""
    return list(v[0::2]), list(v[1::2])


# --------------------------------------------------------------------------------------
# Keccak-256 (legacy 0x01 padding) -- iden3 keccak256.Hash, used at go/bp/util.go:232.
# Published algorithm (Keccak-f[1600], rate 1088); self-tested against the well-known
# digest of the empty string in tests.
# --------------------------------------------------------------------------------------
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [
    [0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61],
    [28, 55, 25, 21, 56], [27, 20, 39, 8, 14],
]
_M64 = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def _keccak_f(A):
    for rc in _RC:
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        Bm = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                Bm[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[Bm[x][y] ^ ((~Bm[(x + 1) % 5][y]) & Bm[(x + 2) % 5][y]) for y in range(5)]
             for x in range(5)]
        A[0][0] ^= rc
    return A


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        blk = msg[off:off + rate]
        for i in range(rate // 8):
            A[i % 5][i // 5] ^= int.from_bytes(blk[8 * i:8 * i + 8], "little")
        A = _keccak_f(A)
    out = b""
    for i in range(4):
        out += A[i % 5][i // 5].to_bytes(8, "little")
    return out


def fs_hash(scalars, points) -> int:
    """hash(), go/bp/util.go:222-233."""
    data = b"".join(scalar_to_32(s) for s in scalars) + b"".join(marshal(p) for p in points)
    return int.from_bytes(keccak256(data), "big") % N


# --------------------------------------------------------------------------------------
# crypto/rand stand-ins: rand.Int rejection loop and RandomG1.
# reader(n) -> n bytes.
# --------------------------------------------------------------------------------------
class ByteStream:
    """Deterministic rand.Reader stand-in: SplitMix64 stream, 8 bytes big-endian per step."""

    def __init__(self, seed: int):
        self.s = seed & _M64

    def next64(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def read(self, n: int) -> bytes:
        out = b""
        while len(out) < n:
            out += self.next64().to_bytes(8, "big")
        return out[:n]


def rand_int(reader: ByteStream, mx: int = N) -> int:
    """crypto/rand.Int(reader, max) for a 256-bit max: 32 bytes BE, reject if >= max."""
    assert mx.bit_length() == 256
    while True:
        k = int.from_bytes(reader.read(32), "big")
        if k < mx:
            return k


def random_g1(reader: ByteStream):
    """bn256.RandomG1: k <- rand.Int (retry on 0), return k.G0."""
    while True:
        k = rand_int(reader)
        if k:
            return k, base_mul(k)


def values(reader, n):
    """go/bp/util.go:51-63."""
    return [rand_int(reader) for _ in range(n)]


def points(reader, n):
    """go/bp/util.go:81-90."""
    return [random_g1(reader)[1] for _ in range(n)]


# --------------------------------------------------------------------------------------
# Synthetic workload generators shared with the device library (bn256_random_bases /
# bn256_random_scalars).  Not in the reference: replaces points(n)/values(n) at scale
# (SURVEY.md 8a row a16, 8d).  Definition: word j of element i =
# splitmix64_mix(seed + (4*i + j + 1) * GOLDEN); 256-bit value = w0 | w1<<64 | w2<<128 |
# w3<<192, reduced mod n; bases use k = value (0 -> 1) and base_i = k.G0.
# --------------------------------------------------------------------------------------
def _mix64(z: int) -> int:
    z &= _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def synth_scalar(seed: int, i: int) -> int:
    v = 0
    for j in range(4):
        v |= _mix64(seed + (4 * i + j + 1) * 0x9E3779B97F4A7C15) << (64 * j)
    return v % N


def synth_scalars(seed: int, n: int):
    return [synth_scalar(seed, i) for i in range(n)]


def synth_base_scalar(seed: int, i: int) -> int:
    return synth_scalar(seed, i) or 1


def synth_bases(seed: int, n: int):
    return [base_mul(synth_base_scalar(seed, i)) for i in range(n)]


# --------------------------------------------------------------------------------------
# Bulletproofs range proof + inner-product argument, go/bp/main.go.
# --------------------------------------------------------------------------------------
class InnerArgumentPublic:
    """go/bp/main.go:38-55."""

    def __init__(self, n, G, H, U):
        self.N, self.G, self.H, self.U = n, G, H, U

    @classmethod
    def new(cls, reader, n):
        G = points(reader, n)
        H = points(reader, n)
        U = points(reader, 1)[0]
        return cls(n, G, H, U)


class BulletProofPublic:
    """go/bp/main.go:17-34."""

    def __init__(self, inner, n, G, H):
        self.inner, self.N, self.G, self.H = inner, n, G, H

    @classmethod
    def new(cls, reader, n):
        inner = InnerArgumentPublic.new(reader, n)
        G = points(reader, 1)[0]
        H = points(reader, 1)[0]
        return cls(inner, n, G, H)


def inner_arg_proof(pub: InnerArgumentPublic, a, b):
    """go/bp/main.go:321-377.  Returns (L[], R[], a, b, P_top)."""
    if pub.N == 1:
        return [], [], a[0], b[0], None
    if pub.N % 2 != 0:
        raise ValueError("invalid n: should be 2^x")
    n1 = pub.N // 2
    cl = vector_mul(a[:n1], b[n1:])
    cr = vector_mul(a[n1:], b[:n1])
    L = add(vec_com(pub.G[n1:], pub.H[:n1], a[:n1], b[n1:]), mul(cl, pub.U))
    R = add(vec_com(pub.G[:n1], pub.H[n1:], a[n1:], b[:n1]), mul(cr, pub.U))
    Pm = product_com(pub.G, pub.H, pub.U, a, b)
    x = fs_hash([pub.N], [Pm, L, R])
    xinv = finv(x)
    g1 = hadamard_point_mul(vector_point_mul_on_scalar(pub.G[:n1], xinv),
                            vector_point_mul_on_scalar(pub.G[n1:], x))
    h1 = hadamard_point_mul(vector_point_mul_on_scalar(pub.H[:n1], x),
                            vector_point_mul_on_scalar(pub.H[n1:], xinv))
    a1 = vector_add(vector_mul_on_scalar(a[:n1], x), vector_mul_on_scalar(a[n1:], xinv))
    b1 = vector_add(vector_mul_on_scalar(b[n1:], x), vector_mul_on_scalar(b[:n1], xinv))
    Ls, Rs, fa, fb, _ = inner_arg_proof(InnerArgumentPublic(n1, g1, h1, pub.U), a1, b1)
    return [L] + Ls, [R] + Rs, fa, fb, Pm


def inner_arg_verify(pub: InnerArgumentPublic, Pm, L, R, a, b, deep=0):
    """go/bp/main.go:273-319.  Returns None on success, error string otherwise."""
    if pub.N == 1:
        p = add(com(pub.G[0], pub.H[0], a, b), mul(fmul(a, b), pub.U))
        if marshal(Pm) != marshal(p):
            return "failed to verify inner argument proof"
        return None
    if pub.N % 2 != 0:
        return "invalid n: should be 2^x"
    n1 = pub.N // 2
    x = fs_hash([pub.N], [Pm, L[deep], R[deep]])
    xinv = finv(x)
    x2 = fmul(x, x)
    x2inv = finv(x2)
    g1 = hadamard_point_mul(vector_point_mul_on_scalar(pub.G[:n1], xinv),
                            vector_point_mul_on_scalar(pub.G[n1:], x))
    h1 = hadamard_point_mul(vector_point_mul_on_scalar(pub.H[:n1], x),
                            vector_point_mul_on_scalar(pub.H[n1:], xinv))
    p = add(mul(x2, L[deep]), Pm)
    p = add(p, mul(x2inv, R[deep]))
    return inner_arg_verify(InnerArgumentPublic(n1, g1, h1, pub.U), p, L, R, a, b, deep + 1)


def bp_prove(pub: BulletProofPublic, v: int, prv: int, reader: ByteStream):
    """go/bp/main.go:81-178.  Randomness order: alpha, sl, sr, ro, tau1, tau2."""
    n = pub.N
    IG, IH = pub.inner.G, pub.inner.H
    onen = [1] * n
    al = to_bits(v, n)
    ar = vector_sub(al, onen)
    alpha = values(reader, 1)[0]
    A = add(vec_com(IG, IH, al, ar), mul(alpha, pub.H))
    sl = values(reader, n)
    sr = values(reader, n)
    ro = values(reader, 1)[0]
    S = add(vec_com(IG, IH, sl, sr), mul(ro, pub.H))
    V = com(pub.G, pub.H, v, prv)
    y = fs_hash([n], [A, S, V])
    z = fs_hash([y], [A, S])
    yn = ntharr(y, n)
    z2 = fmul(z, z)
    twon = ntharr(2, n)
    t1 = fadd(
        vector_mul(hadamard_mul(yn, sr), vector_sub(al, vector_mul_on_scalar(onen, z))),
        vector_mul(sl, vector_add(vector_mul_on_scalar(twon, z2),
                                  hadamard_mul(yn, vector_add(ar, vector_mul_on_scalar(onen, z))))),
    )
    t2 = vector_mul(hadamard_mul(yn, sr), sl)
    tau1 = values(reader, 1)[0]
    tau2 = values(reader, 1)[0]
    T1 = com(pub.G, pub.H, t1, tau1)
    T2 = com(pub.G, pub.H, t2, tau2)
    x = fs_hash([y, z], [T1, T2])
    x2 = fmul(x, x)
    l = vector_add(vector_sub(al, vector_mul_on_scalar(onen, z)), vector_mul_on_scalar(sl, x))
    r = vector_add(
        hadamard_mul(yn, vector_add(ar, vector_add(vector_mul_on_scalar(onen, z),
                                                   vector_mul_on_scalar(sr, x)))),
        vector_mul_on_scalar(twon, z2),
    )
    tx = vector_mul(l, r)
    taux = fadd(fmul(tau2, x2), fadd(fmul(tau1, x), fmul(z2, prv)))
    nu = fadd(alpha, fmul(ro, x))
    yinvn = invntharr(y, n)
    h1 = [mul(yinvn[i], IH[i]) for i in range(n)]
    Ls, Rs, fa, fb, _ = inner_arg_proof(InnerArgumentPublic(n, IG, h1, pub.inner.U), l, r)
    return dict(V=V, A=A, S=S, T1=T1, T2=T2, tx=tx, taux=taux, nu=nu, L=Ls, R=Rs, a=fa, b=fb,
                y=y, z=z, x=x)


def bp_verify(pub: BulletProofPublic, pr) -> str | None:
    """go/bp/main.go:181-249.  Returns None on success or the reference's error string."""
    n = pub.N
    IG, IH = pub.inner.G, pub.inner.H
    y = fs_hash([n], [pr["A"], pr["S"], pr["V"]])
    z = fs_hash([y], [pr["A"], pr["S"]])
    yn = ntharr(y, n)
    z2 = fmul(z, z)
    z3 = fmul(z2, z)
    onen = [1] * n
    twon = ntharr(2, n)
    x = fs_hash([y, z], [pr["T1"], pr["T2"]])
    x2 = fmul(x, x)
    yinvn = invntharr(y, n)
    h1 = [mul(yinvn[i], IH[i]) for i in range(n)]
    deltayz = fsub(fmul(fsub(z, z2), vector_mul(onen, yn)), fmul(z3, vector_mul(onen, twon)))
    c1 = com(pub.G, pub.H, pr["tx"], pr["taux"])
    c2 = mul(z2, pr["V"])
    c2 = add(c2, mul(deltayz, pub.G))
    c2 = add(c2, mul(x, pr["T1"]))
    c2 = add(c2, mul(x2, pr["T2"]))
    if marshal(c1) != marshal(c2):
        return "failed: tx ?= t0 + t1*x +t2*x^2"
    Pm = add(pr["A"], mul(x, pr["S"]))
    Pm = add(Pm, vector_point_scalar_mul(IG, vector_mul_on_scalar(onen, fsub(0, z))))
    Pm = add(Pm, vector_point_scalar_mul(
        h1, vector_add(vector_mul_on_scalar(yn, z), vector_mul_on_scalar(twon, z2))))
    Pm = add(Pm, mul(fsub(0, pr["nu"]), pub.H))
    Pm = add(Pm, mul(pr["tx"], pub.inner.U))
    return inner_arg_verify(InnerArgumentPublic(n, IG, h1, pub.inner.U), Pm,
                            pr["L"], pr["R"], pr["a"], pr["b"])


# --------------------------------------------------------------------------------------
# Bulletproofs++ weighted-norm-linear argument, go/bppp/docs_test.go:1242-1338.
# --------------------------------------------------------------------------------------
def values_bppp(reader: ByteStream, n: int):
    """go/bppp/util.go:11-25: rand.Int is drawn and discarded, the value is the constant 2."""
    out = []
    for _ in range(n):
        rand_int(reader)
        out.append(2)
    return out


def reduce_points(v):
    """go/bppp/util.go:213-226."""
    return list(v[0::2]), list(v[1::2])


def vector_point_mul_on_scalar_bppp(g, a):
    return [mul(a % N, p) for p in g]


def wnla(g, G, H, c, C, ro, mu, l, n, reader, rounds=None):
    """go/bppp/docs_test.go:1269-1338.  Returns (verified, rounds) with rounds = [(X, R), ...];
    raises ValueError("Failed to verify!") where the reference panics."""
    rounds = [] if rounds is None else rounds
    roinv = finv(ro)
    if len(l) + len(n) < 6:
        v = fadd(vector_mul_bppp(c, l), weight_vector_mul(n, n, mu))
        Cc = mul(v, g)
        Cc = add(Cc, vector_point_scalar_mul_bppp(H, l))
        Cc = add(Cc, vector_point_scalar_mul_bppp(G, n))
        if marshal(Cc) != marshal(C):
            raise ValueError("Failed to verify!")
        return True, rounds, l, n
    y = values_bppp(reader, 1)[0]
    c0, c1 = reduce_vector(c)
    l0, l1 = reduce_vector(l)
    n0, n1 = reduce_vector(n)
    G0, G1 = reduce_points(G)
    H0, H1 = reduce_points(H)
    l_ = vector_add_bppp(l0, vector_mul_on_scalar(l1, y))
    n_ = vector_add_bppp(vector_mul_on_scalar(n0, roinv), vector_mul_on_scalar(n1, y))
    mu2 = fmul(mu, mu)
    vx = fadd(fmul(weight_vector_mul(n0, n1, mu2), fmul(2, roinv)),
              fadd(vector_mul_bppp(c0, l1), vector_mul_bppp(c1, l0)))
    vr = fadd(weight_vector_mul(n1, n1, mu2), vector_mul_bppp(c1, l1))
    X = mul(vx, g)
    X = add(X, vector_point_scalar_mul_bppp(H0, l1))
    X = add(X, vector_point_scalar_mul_bppp(H1, l0))
    X = add(X, vector_point_scalar_mul_bppp(G0, vector_mul_on_scalar(n1, ro)))
    X = add(X, vector_point_scalar_mul_bppp(G1, vector_mul_on_scalar(n0, roinv)))
    R = mul(vr, g)
    R = add(R, vector_point_scalar_mul_bppp(H1, l1))
    R = add(R, vector_point_scalar_mul_bppp(G1, n1))
    rounds.append((X, R))
    H_ = vector_points_add_bppp(H0, vector_point_mul_on_scalar_bppp(H1, y))
    G_ = vector_points_add_bppp(vector_point_mul_on_scalar_bppp(G0, ro), vector_point_mul_on_scalar_bppp(G1, y))
    c_ = vector_add_bppp(c0, vector_mul_on_scalar(c1, y))
    C_ = add(C, mul(y, X))
    C_ = add(C_, mul(fsub(fmul(y, y), 1), R))
    return wnla(g, G_, H_, c_, C_, mu, mu2, l_, n_, reader, rounds)


def wnla_setup_and_run(seed: int, l, n):
    """TestWNLA, go/bppp/docs_test.go:1242-1267, with the reader injected."""
    rd = ByteStream(seed)
    g = points(rd, 1)[0]
    G = points(rd, len(n))
    H = points(rd, len(l))
    c = values_bppp(rd, len(l))
    ro = values_bppp(rd, 1)[0]
    mu = fmul(ro, ro)
    v = fadd(vector_mul_bppp(c, l), weight_vector_mul(n, n, mu))
    C = mul(v, g)
    C = add(C, vector_point_scalar_mul_bppp(H, l))
    C = add(C, vector_point_scalar_mul_bppp(G, n))
    ok, rounds, lf, nf = wnla(g, G, H, c, C, ro, mu, list(l), list(n), rd)
    return dict(C=C, rounds=rounds, l=lf, n=nf, ok=ok)


# --------------------------------------------------------------------------------------
# Bulletproofs++ arithmetic-circuit protocol, go/bppp/docs_test.go:496-1240: the commitment
# path (Com, CommitOL, CommitR, V_, Cs, PT, CT, the three bytes.Equal checks, then wnla) with
# the scalar algebra that feeds it.  The reference's debug prints (Wl*w, Eq. 34, the
# polynomial cross-checks) produce no protocol value and are left out.
# --------------------------------------------------------------------------------------
def fpow(x, y):
    """go/bppp/util.go:110-116."""
    return pow(finv(x), -y, N) if y < 0 else pow(x % N, y, N)


def vector_mul_on_matrix(a, m):
    """go/bppp/util.go:295-309: a * M (vector of column inner products)."""
    return [vector_mul_bppp(a, [row[j] for row in m]) for j in range(len(m[0]))]


def matrix_mul_on_vector(a, m):
    """go/bppp/util.go:311-319."""
    return [vector_mul_bppp(a, row) for row in m]


def diag_inv(x, n):
    """go/bppp/util.go:321-340."""
    inv = finv(x)
    res, val = [[0] * n for _ in range(n)], inv
    for i in range(n):
        res[i][i] = val
        val = fmul(val, inv)
    return res


def com_bppp(v, s, G, H):
    """Com, go/bppp/docs_test.go:514-519."""
    res = mul(v[0] % N, G)
    res = add(res, mul(s % N, H[0]))
    return add(res, vector_point_scalar_mul_bppp(H[9:], v[1:]))


def _partition(fmap, typ, n, wo):
    """lo / ll / lr / no of CommitOL / CommitR (docs_test.go:533-558, 579-586): fmap[typ][j] is the
    index into wo that the PartitionF returns for (typ, j), or None."""
    out = []
    for j in range(n):
        i = fmap[typ][j]
        out.append(0 if i is None else wo[i] % N)
    return out


def commit_ol(pub, wo, wl, fmap, reader):
    """CommitOL, go/bppp/docs_test.go:522-567."""
    ro_ = values_bppp(reader, 7)
    rl_ = values_bppp(reader, 6)
    ro = [ro_[0], ro_[1], ro_[2], ro_[3], 0, ro_[4], ro_[5], ro_[6], 0]
    rl = [rl_[0], rl_[1], rl_[2], 0, rl_[3], rl_[4], rl_[5], 0, 0]
    nl = list(wl)
    no = _partition(fmap, 4, pub["Nm"], wo)
    lo = _partition(fmap, 1, pub["Nv"], wo)
    ll = _partition(fmap, 2, pub["Nv"], wo)
    Co = add(vector_point_scalar_mul_bppp(pub["HVec"], ro + lo), vector_point_scalar_mul_bppp(pub["GVec"], no))
    Cl = add(vector_point_scalar_mul_bppp(pub["HVec"], rl + ll), vector_point_scalar_mul_bppp(pub["GVec"], nl))
    return ro, rl, no, nl, lo, ll, Co, Cl


def commit_r(pub, wo, wr, fmap, reader):
    """CommitR, go/bppp/docs_test.go:569-591."""
    rr_ = values_bppp(reader, 5)
    rr = [rr_[0], rr_[1], 0, rr_[2], rr_[3], rr_[4], 0, 0, 0]
    nr = list(wr)
    lr = _partition(fmap, 3, pub["Nv"], wo)
    Cr = add(vector_point_scalar_mul_bppp(pub["HVec"], rr + lr), vector_point_scalar_mul_bppp(pub["GVec"], nr))
    return rr, nr, lr, Cr


def ac_protocol(pub, prv, reader):
    """ArithmeticCircuitProtocol + InnerArithmeticCircuitProtocol2, go/bppp/docs_test.go:496-1240.
    pub: Nm, Nl, Nv, Nw, No, K, G, GVec, HVec, Wm, Wl, Am, Al, Fl, Fm.  prv: v (K x Nv), sv (K), wl,
    wr, wo, f = {typ: [index or None per j]} for typ 1..4 (the PartitionF tabulated).
    Returns the transcript as a dict (points as affine tuples / None)."""
    Nm, Nl, Nv, K = pub["Nm"], pub["Nl"], pub["Nv"], pub["K"]
    fmap = prv["f"]
    V = [com_bppp(prv["v"][i], prv["sv"][i], pub["G"], pub["HVec"]) for i in range(K)]
    ro, rl, no, nl, lo, ll, Co, Cl = commit_ol(pub, prv["wo"], prv["wl"], fmap, reader)
    rr, nr, lr, Cr = commit_r(pub, prv["wo"], prv["wr"], fmap, reader)

    ch_ro = values_bppp(reader, 1)[0]
    ch_lambda = values_bppp(reader, 1)[0]
    ch_beta = values_bppp(reader, 1)[0]
    ch_delta = values_bppp(reader, 1)[0]

    Wl, Wm = pub["Wl"], pub["Wm"]
    MlnL = [row[:Nm] for row in Wl]
    MmnL = [row[:Nm] for row in Wm]
    MlnR = [row[Nm:2 * Nm] for row in Wl]
    MmnR = [row[Nm:2 * Nm] for row in Wm]
    WlO = [row[2 * Nm:] for row in Wl]
    WmO = [row[2 * Nm:] for row in Wm]

    def mapped(W, typ, n):  # docs_test.go:652-759
        return [[0 if fmap[typ][j] is None else row[fmap[typ][j]] % N for j in range(n)] for row in W]

    MlnO, MmnO = mapped(WlO, 4, Nm), mapped(WmO, 4, Nm)
    MllL, MmlL = mapped(WlO, 2, Nv), mapped(WmO, 2, Nv)
    MllR, MmlR = mapped(WlO, 3, Nv), mapped(WmO, 3, Nv)
    MllO, MmlO = mapped(WlO, 1, Nv), mapped(WmO, 1, Nv)

    ch_mu = fmul(ch_ro, ch_ro)
    Fl, Fm = int(bool(pub["Fl"])), int(bool(pub["Fm"]))

    def lcomb(i):  # :778-783
        return fadd(fmul(Fl, fpow(ch_lambda, Nv * i)), fmul(Fm, fpow(ch_mu, Nv * i + 1)))

    V_ = None  # :786-797
    for i in range(K):
        V_ = add(V_, mul(lcomb(i), V[i]))
    V_ = mul(2, V_)

    lam = vector_add_bppp(  # :800-806
        vector_tensor_mul(vector_mul_on_scalar(powvector(ch_lambda, Nv), ch_mu), powvector(fpow(ch_mu, Nv), K)),
        vector_tensor_mul(powvector(ch_mu, Nv), powvector(fpow(ch_lambda, Nv), K)))
    lam = vector_mul_on_scalar(lam, Fl & Fm)
    lam = vector_sub_bppp(powvector(ch_lambda, Nl), lam)
    mu = vector_mul_on_scalar(powvector(ch_mu, Nm), ch_mu)  # :809

    muDiagInv = diag_inv(ch_mu, Nm)  # :824-833
    cnL = vector_mul_on_matrix(vector_sub_bppp(vector_mul_on_matrix(lam, MlnL), vector_mul_on_matrix(mu, MmnL)), muDiagInv)
    cnR = vector_mul_on_matrix(vector_sub_bppp(vector_mul_on_matrix(lam, MlnR), vector_mul_on_matrix(mu, MmnR)), muDiagInv)
    cnO = vector_mul_on_matrix(vector_sub_bppp(vector_mul_on_matrix(lam, MlnO), vector_mul_on_matrix(mu, MmnO)), muDiagInv)
    clL = vector_sub_bppp(vector_mul_on_matrix(lam, MllL), vector_mul_on_matrix(mu, MmlL))
    clR = vector_sub_bppp(vector_mul_on_matrix(lam, MllR), vector_mul_on_matrix(mu, MmlR))
    clO = vector_sub_bppp(vector_mul_on_matrix(lam, MllO), vector_mul_on_matrix(mu, MmlO))

    ls = values_bppp(reader, Nv)  # :836-837
    ns = values_bppp(reader, Nm)

    v_ = 0  # :840-851
    for i in range(K):
        v_ = fadd(v_, fmul(prv["v"][i][0], lcomb(i)))
    v_ = fmul(v_, 2)
    rv = [0] * 9  # :853-865
    rv1 = 0
    for i in range(K):
        rv1 = fadd(rv1, fmul(prv["sv"][i], lcomb(i)))
    rv[0] = fmul(rv1, 2)
    v_1 = [0]  # :868-879
    for i in range(K):
        v_1 = vector_add_bppp(v_1, vector_mul_on_scalar(prv["v"][i][1:], lcomb(i)))
    v_1 = vector_mul_on_scalar(v_1, 2)

    check = add(mul(v_, pub["G"]), vector_point_scalar_mul_bppp(pub["HVec"], rv + v_1))  # :882-886
    check_V = marshal(V_) == marshal(check)

    cl0 = vector_sub_bppp(  # :888-891
        vector_mul_on_scalar(powvector(ch_lambda, Nv)[1:], Fl),
        vector_mul_on_scalar(vector_mul_on_scalar(powvector(ch_mu, Nv)[1:], ch_mu), Fm))

    wv, vm, d, dinv = weight_vector_mul, vector_mul_bppp, ch_delta, finv(ch_delta)
    two = 2
    f_ = {k: 0 for k in range(-2, 7)}  # :894-946
    f_[-2] = fsub(f_[-2], wv(ns, ns, ch_mu))
    f_[-1] = fadd(f_[-1], vm(cl0, ls))
    f_[-1] = fadd(f_[-1], fmul(fmul(two, d), wv(ns, no, ch_mu)))
    f_[0] = fsub(f_[0], fmul(two, vm(clR, ls)))
    f_[0] = fsub(f_[0], fmul(d, vm(cl0, lo)))
    f_[0] = fsub(f_[0], fmul(wv(ns, vector_add_bppp(nl, cnR), ch_mu), two))
    f_[0] = fsub(f_[0], fmul(fmul(d, d), wv(no, no, ch_mu)))
    f_[1] = fadd(f_[1], fmul(two, vm(clL, ls)))
    f_[1] = fadd(f_[1], fmul(two, fmul(d, vm(clR, lo))))
    f_[1] = fadd(f_[1], vm(cl0, ll))
    f_[1] = fadd(f_[1], fmul(wv(ns, vector_add_bppp(nr, cnL), ch_mu), two))
    f_[1] = fadd(f_[1], fmul(wv(no, vector_add_bppp(nl, cnR), ch_mu), fmul(two, d)))
    f_[2] = fadd(f_[2], wv(cnR, cnR, ch_mu))
    f_[2] = fsub(f_[2], fmul(two, fmul(dinv, vm(clO, ls))))
    f_[2] = fsub(f_[2], fmul(two, fmul(d, vm(clL, lo))))
    f_[2] = fsub(f_[2], fmul(two, vm(clR, ll)))
    f_[2] = fsub(f_[2], vm(cl0, lr))
    f_[2] = fsub(f_[2], fmul(fmul(two, dinv), wv(ns, cnO, ch_mu)))
    f_[2] = fsub(f_[2], fmul(fmul(two, d), wv(no, vector_add_bppp(nr, cnL), ch_mu)))
    f_[2] = fsub(f_[2], wv(vector_add_bppp(nl, cnR), vector_add_bppp(nl, cnR), ch_mu))
    f_[4] = fadd(f_[4], fmul(fmul(two, dinv), wv(cnO, cnR, ch_mu)))
    f_[4] = fadd(f_[4], wv(cnL, cnL, ch_mu))
    f_[4] = fsub(f_[4], fmul(fmul(two, dinv), vm(clO, ll)))
    f_[4] = fsub(f_[4], fmul(two, vm(clL, lr)))
    f_[4] = fsub(f_[4], fmul(two, vm(clR, v_1)))
    f_[4] = fsub(f_[4], fmul(fmul(two, dinv), wv(vector_add_bppp(nl, cnR), cnO, ch_mu)))
    f_[4] = fsub(f_[4], wv(vector_add_bppp(nr, cnL), vector_add_bppp(nr, cnL), ch_mu))
    f_[5] = fsub(f_[5], fmul(fmul(two, dinv), wv(cnO, cnL, ch_mu)))
    f_[5] = fadd(f_[5], fmul(fmul(two, dinv), vm(clO, lr)))
    f_[5] = fadd(f_[5], fmul(two, vm(clL, v_1)))
    f_[5] = fadd(f_[5], fmul(fmul(two, dinv), wv(vector_add_bppp(nr, cnL), cnO, ch_mu)))
    f_[6] = fsub(f_[6], fmul(fmul(two, dinv), vm(clO, v_1)))
    # f_[3] (:938-946) is printed only; it enters no commitment

    binv = finv(ch_beta)  # :984-998
    rs = [
        fadd(f_[-1], fmul(ch_beta, fmul(d, ro[1]))),
        fmul(f_[-2], binv),
        fmul(fsub(fadd(f_[0], fmul(d, ro[0])), fmul(ch_beta, rl[1])), binv),
        fadd(fmul(fsub(f_[1], rl[0]), binv), fadd(rr[1], fmul(d, ro[2]))),
        fadd(fmul(fadd(f_[2], rr[0]), binv), fsub(fmul(d, ro[3]), rl[2])),
        fsub(0, fmul(rv[0], binv)),
        fadd(fmul(f_[4], binv), fadd(fmul(d, ro[5]), fsub(rr[3], rl[4]))),
        fadd(fmul(f_[5], binv), fsub(fadd(rr[4], fmul(d, ro[6])), rl[5])),
        fadd(fmul(f_[6], binv), fadd(fsub(fmul(d, ro[7]), rl[6]), rr[5])),
    ]
    Cs = add(vector_point_scalar_mul_bppp(pub["HVec"], rs + ls), vector_point_scalar_mul_bppp(pub["GVec"], ns))  # :1002-1003

    t = values_bppp(reader, 1)[0]  # :1009-1012
    tinv, t2 = finv(t), fmul(t, t)
    t3 = fmul(t2, t)

    lT = vector_mul_on_scalar(rs + ls, tinv)  # :1038-1042
    lT = vector_sub_bppp(lT, vector_mul_on_scalar(ro + lo, d))
    lT = vector_add_bppp(lT, vector_mul_on_scalar(rl + ll, t))
    lT = vector_sub_bppp(lT, vector_mul_on_scalar(rr + lr, t2))
    lT = vector_add_bppp(lT, vector_mul_on_scalar(rv + v_1, t3))

    pnT = vector_mul_on_scalar(cnO, fmul(dinv, t3))  # :1044-1050
    pnT = vector_sub_bppp(pnT, vector_mul_on_scalar(cnL, t2))
    pnT = vector_add_bppp(pnT, vector_mul_on_scalar(cnR, t))
    psT = wv(pnT, pnT, ch_mu)
    psT = fadd(psT, fmul(two, fmul(vm(lam, pub["Al"]), t3)))
    psT = fsub(psT, fmul(two, fmul(vm(mu, pub["Am"]), t3)))

    n_T = vector_mul_on_scalar(ns, tinv)  # :1069-1074
    n_T = vector_sub_bppp(n_T, vector_mul_on_scalar(no, d))
    n_T = vector_add_bppp(n_T, vector_mul_on_scalar(nl, t))
    n_T = vector_sub_bppp(n_T, vector_mul_on_scalar(nr, t2))
    nT = vector_add_bppp(pnT, n_T)

    PT = add(mul(psT, pub["G"]), vector_point_scalar_mul_bppp(pub["GVec"], pnT))  # :1077-1078

    cr_T = [1, fmul(ch_beta, tinv), fmul(ch_beta, t), fmul(ch_beta, t2), fmul(ch_beta, t3),  # :1080-1096
            fmul(ch_beta, fmul(t, t3)), fmul(ch_beta, fmul(t2, t3)), fmul(ch_beta, fmul(t3, t3)),
            fmul(ch_beta, fmul(fmul(t3, t), t3))]
    cl_T = vector_mul_on_scalar(clO, fmul(t3, dinv))
    cl_T = vector_sub_bppp(cl_T, vector_mul_on_scalar(clL, t2))
    cl_T = vector_add_bppp(cl_T, vector_mul_on_scalar(clR, t))
    cl_T = vector_mul_on_scalar(cl_T, 2)
    cl_T = vector_sub_bppp(cl_T, cl0)
    cT = cr_T + cl_T  # :1171

    CT = add(PT, mul(tinv, Cs))  # :1173-1177
    CT = add(CT, mul(fsub(0, d), Co))
    CT = add(CT, mul(t, Cl))
    CT = add(CT, mul(fsub(0, t2), Cr))
    CT = add(CT, mul(t3, V_))
    vT = fadd(psT, fmul(v_, t3))  # :1179

    CTPrv = add(mul(vT, pub["G"]), vector_point_scalar_mul_bppp(pub["HVec"], lT))  # :1183-1189
    CTPrv = add(CTPrv, vector_point_scalar_mul_bppp(pub["GVec"], nT))
    check_CT = marshal(CT) == marshal(CTPrv)

    CLeft = mul(tinv, Cs)  # :1192-1203
    CLeft = add(CLeft, mul(fsub(0, d), Co))
    CLeft = add(CLeft, mul(t, Cl))
    CLeft = add(CLeft, mul(fsub(0, t2), Cr))
    CLeft = add(CLeft, mul(t3, V_))
    CTRight = add(mul(fmul(v_, t3), pub["G"]), vector_point_scalar_mul_bppp(pub["HVec"], lT))
    CTRight = add(CTRight, vector_point_scalar_mul_bppp(pub["GVec"], n_T))
    check_45 = marshal(CLeft) == marshal(CTRight)

    try:  # :1239
        ok, rounds, lf, nf = wnla(pub["G"], list(pub["GVec"]), list(pub["HVec"]), cT, CT, ch_ro, ch_mu, lT, nT, reader)
    except ValueError:
        ok, rounds, lf, nf = False, None, None, None
    return dict(V=V, Co=Co, Cl=Cl, Cr=Cr, V_=V_, Cs=Cs, PT=PT, CT=CT, check_V=check_V, check_CT=check_CT,
                check_45=check_45, vT=vT, rs=rs, lT=lT, nT=nT, cT=cT, wnla_ok=ok, rounds=rounds, l=lf, n=nf)


# --------------------------------------------------------------------------------------
# go/schnorr-bn256.  Hash = go-ethereum crypto.Keccak256 = legacy Keccak-256
# over the concatenation (go/schnorr-bn256/types.go:21).
# --------------------------------------------------------------------------------------
def big_bytes(x: int) -> bytes:
    """big.Int.Bytes(): minimal big-endian, empty for 0."""
    return x.to_bytes((x.bit_length() + 7) // 8, "big")


def schnorr_r(reader, G):
    """R(G), go/schnorr-bn256/schnorr.go:17-24: r <- rand.Int, (r, r*G)."""
    r = rand_int(reader)
    return r, mul(r, G)


def schnorr_msg(*chunks: bytes) -> int:
    """Msg, go/schnorr-bn256/schnorr.go:26-28."""
    return int.from_bytes(keccak256(b"".join(chunks)), "big") % N


def schnorr_multisig(prv: int, r: int, pub_common, r_common, m: int):
    """MultiSigSchnorr, go/schnorr-bn256/schnorr.go:32-40.  s = r + hash*prv is NOT reduced there."""
    h = schnorr_msg(big_bytes(m), marshal(pub_common), marshal(r_common))
    return r_common, r + h * prv


def schnorr_sign(prv: int, pub, G, m: int, reader):
    """SignSchnorr, go/schnorr-bn256/schnorr.go:44-59."""
    r, R = schnorr_r(reader, G)
    return schnorr_multisig(prv, r, pub, R, m)


def schnorr_verify(sig, pub, G, m: int) -> bool:
    """VerifySchnorr, go/schnorr-bn256/schnorr.go:64-79."""
    R, s = sig
    h = schnorr_msg(big_bytes(m), marshal(pub), marshal(R))
    p1 = add(mul(h, pub), R)
    p2 = mul(s % N, G)          # k*G = (k mod n)*G
    return marshal(p1) == marshal(p2)


# --------------------------------------------------------------------------------------
# go/ve-ca: verifiable encryption by cut-and-choose, k = 128 (go/ve-ca/main.go:8).  F = big.Int mod
# Order, G = bn256.G1 (go/ve-ca/types.go); Hash = Keccak256Hash(...) mod Order (types.go:93-96);
# FBytes = big.Int.Bytes() (minimal); E / D = XOR of the 32-byte paddings (go/ve-ca/enc.go).
# --------------------------------------------------------------------------------------
VECA_K = 128


def veca_hash(*chunks: bytes) -> int:
    return int.from_bytes(keccak256(b"".join(chunks)), "big") % N


def veca_e(key: int, val: int) -> int:
    """E (and D: the same XOR), go/ve-ca/enc.go:7-25.  The result may exceed Order."""
    return int.from_bytes(bytes(a ^ b for a, b in zip(key.to_bytes(32, "big"), val.to_bytes(32, "big"))), "big")


def _veca_rho(GenG, GenH, C, X):
    pre = marshal(GenG) + marshal(GenH) + marshal(C) + b"".join(marshal(x) for x in X)
    return veca_hash(pre, b"\x00"), veca_hash(pre, b"\x01")


def _veca_c(rho0, rho1, e):
    return veca_hash(big_bytes(rho0), big_bytes(rho1), *[b"".join(big_bytes(x) for x in row) for row in e])


def veca_encrypt(v, r, s_r, g, GenG, GenH, reader, k=VECA_K):
    """Encrypt, go/ve-ca/main.go:18-92.  Draw order: p[0..3][k], then x[0..1][k] (MustSampleRandomF)."""
    p = [[rand_int(reader) for _ in range(k)] for _ in range(4)]
    x = [[rand_int(reader) for _ in range(k)] for _ in range(2)]
    s = [[pow(s_r, p[i][j], N) for j in range(k)] for i in range(4)]
    X = [add(mul(x[0][i], GenH), mul(x[1][i], GenG)) for i in range(k)]
    C = add(mul(v, GenH), mul(r, GenG))
    rho0, rho1 = _veca_rho(GenG, GenH, C, X)
    alpha = [[0] * k for _ in range(4)]
    e = [[0] * k for _ in range(4)]
    for i in range(k):
        alpha[0][i] = (x[0][i] - rho0 * v) % N
        alpha[1][i] = (x[0][i] - rho1 * v) % N
        alpha[2][i] = (x[1][i] - rho0 * r) % N
        alpha[3][i] = (x[1][i] - rho1 * r) % N
        for t in range(4):
            e[t][i] = veca_e(s[t][i], alpha[t][i])
    c = _veca_c(rho0, rho1, e)
    proof = {"E": e, "X": X, "Alpha": [[0] * k, [0] * k], "P": [[0] * k, [0] * k], "Share": [[0] * k, [0] * k]}
    for i in range(k):
        b = (c >> i) & 1
        proof["Alpha"][0][i] = alpha[b][i]
        proof["Alpha"][1][i] = alpha[b + 2][i]
        proof["P"][0][i] = p[b][i]
        proof["P"][1][i] = p[b + 2][i]
        d = 1 - b
        proof["Share"][0][i] = pow(g, p[d][i], N)
        proof["Share"][1][i] = pow(g, p[d + 2][i], N)
    return proof, C


def veca_verify(proof, s_r, C, GenG, GenH, k=VECA_K):
    """Verify, go/ve-ca/main.go:94-133.  Returns None, or the message the reference prints."""
    rho = _veca_rho(GenG, GenH, C, proof["X"])
    c = _veca_c(rho[0], rho[1], proof["E"])
    for i in range(k):
        b = (c >> i) & 1
        X = add(add(mul(proof["Alpha"][0][i], GenH), mul(proof["Alpha"][1][i], GenG)), mul(rho[b], C))
        if marshal(X) != marshal(proof["X"][i]):
            return "invalid sigma answer"
        e0 = veca_e(pow(s_r, proof["P"][0][i], N), proof["Alpha"][0][i])
        e1 = veca_e(pow(s_r, proof["P"][1][i], N), proof["Alpha"][1][i])
        if big_bytes(e0) != big_bytes(proof["E"][b][i]):
            return "invalid e0 answer"
        if big_bytes(e1) != big_bytes(proof["E"][b + 2][i]):
            return "invalid e1 answer"
    return None


def veca_decrypt(u, proof, C, GenG, GenH, k=VECA_K):
    """Decrypt, go/ve-ca/main.go:135-174."""
    rho = _veca_rho(GenG, GenH, C, proof["X"])
    c = _veca_c(rho[0], rho[1], proof["E"])
    for i in range(k):
        b = (c >> i) & 1
        d = 1 - b
        key_v = pow(proof["Share"][0][i], u, N)
        key_r = pow(proof["Share"][1][i], u, N)
        alpha0 = veca_e(key_v, proof["E"][d][i])
        alpha1 = veca_e(key_r, proof["E"][d + 2][i])
        den = pow((rho[b] - rho[d]) % N, -1, N)
        v = (alpha0 - proof["Alpha"][0][i]) % N * den % N
        r = (alpha1 - proof["Alpha"][1][i]) % N * den % N
        if marshal(add(mul(v, GenH), mul(r, GenG))) == marshal(C):
            return v, r
    raise ValueError("Failed to recover")


def veca_proof_bytes(proof, k=VECA_K) -> bytes:
    """Wire format used by libbpamd (the reference defines none): E[4][k] | Alpha[2][k] | P[2][k] |
    Share[2][k] as 32-byte big-endian each, then X[k] as Marshal() bytes."""
    out = b""
    for name, rows in (("E", 4), ("Alpha", 2), ("P", 2), ("Share", 2)):
        for t in range(rows):
            out += b"".join(x.to_bytes(32, "big") for x in proof[name][t])
    return out + b"".join(marshal(x) for x in proof["X"])
