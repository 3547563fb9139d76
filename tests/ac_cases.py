"""The circuits of the reference's BP++ arithmetic-circuit tests, go/bppp/docs_test.go:43-494, as the
dictionaries oracle/bn256_ref.py:ac_protocol and crypto_amd.bp.ArithmeticCircuit.run take (without the
generators, which both sides draw from the seeded reader).  values() of go/bppp/util.go:11-25 always
returns 2, so sv = e = 2.  Each case: (name, pre_draws, post_draws, pub, prv)."""
import bn256_ref as r

N = r.N


def bint(v):
    return v % N


def frac(a, b):
    return r.fmul(bint(a), r.finv(bint(b)))


def test_ac_protocol():
    """TestACProtocol, docs_test.go:43-146: knowledge of p, q with p*q = r."""
    p, q = 3, 5
    m = r.vector_mul_on_scalar([bint(-3), bint(-5)], r.finv(bint(30 + 500)))
    al = r.vector_mul_on_scalar(m, bint(-15 * 1000))
    wlw = r.vector_mul_on_scalar(m, bint(15530))
    winv = [frac(3, 259), frac(5, 259), frac(15, 259)]
    Wl = [[r.fmul(wlw[i], winv[j]) for j in range(3)] for i in range(2)]
    pub = dict(Nm=1, Nl=2, Nv=2, Nw=3, No=1, K=1, Wm=[[0, 0, 1]], Wl=Wl, Am=[0], Al=al, Fl=True, Fm=False)
    f = {1: [None] * 2, 2: [None] * 2, 3: [None] * 2, 4: [0]}           # typ == 4 -> index
    prv = dict(v=[[p, q]], sv=[2], wl=[p], wr=[q], wo=[p * q], f=f)
    return "TestACProtocol", 0, 1, pub, prv


def test_arithmetic_circuit2():
    """TestArithmeticCircuit2, docs_test.go:148-225: x + y = r, x * y = z."""
    x, y, rr, z = 3, 5, 8, 15
    pub = dict(Nm=1, Nl=2, Nv=2, Nw=4, No=2, K=1, Wm=[[0, 0, 1, 0]], Wl=[[0, 1, 0, 0], [1, 0, 0, bint(-1)]],
               Am=[0], Al=[bint(-rr), 0], Fl=True, Fm=False)
    f = {1: [None] * 2, 2: [0, 1], 3: [None] * 2, 4: [None]}            # typ == 2 -> index
    prv = dict(v=[[x, y]], sv=[2], wl=[x], wr=[y], wo=[z, rr], f=f)
    return "TestArithmeticCircuit2", 0, 1, pub, prv


def lookup_circuit(d, m, name):
    """TestAC3 / TestAC4, docs_test.go:227-494: digits d (base len(m)) with pole multiplicities m;
    reciprocal-sum lookup argument with challenge e = values(1)[0] = 2."""
    Nd, Np = len(d), len(m)
    e = 2
    Nm, No = Nd, Np
    Nv = Nd + Nd + Np
    Nl, Nw = Nv, Nd + Nd + Np
    rinv = [r.finv(r.fadd(x, e)) for x in d]
    v_ = [bint(x) for x in d] + rinv + [bint(x) for x in m]
    Wm = [[0] * Nw for _ in range(Nm)]
    for i in range(Nm):
        Wm[i][i + Nm] = bint(-e)
    Wl = [[0] * Nw for _ in range(Nl)]
    for i in range(Nm):
        Wl[i][i] = bint(-1)
    for i in range(Nm):
        for j in range(Nm):
            Wl[i + Nm][j + Nm] = 1
    for i in range(Nm):
        Wl[i + Nm][i + Nm] = 0
    for i in range(Nm):
        for j in range(No):
            Wl[i + Nm][j + 2 * Nm] = r.fsub(0, r.finv(r.fadd(e, bint(j))))
    for i in range(No):
        Wl[i + 2 * Nm][i + 2 * Nm] = bint(-1)
    pub = dict(Nm=Nm, Nl=Nl, Nv=Nv, Nw=Nw, No=No, K=1, Wm=Wm, Wl=Wl, Am=[1] * Nm, Al=[0] * Nl, Fl=True, Fm=False)
    f = {1: [None] * Nv, 2: [j if j < No else None for j in range(Nv)], 3: [None] * Nv, 4: [None] * Nm}
    prv = dict(v=[v_], sv=[2], wl=[bint(x) for x in d], wr=rinv, wo=[bint(x) for x in m], f=f)
    return name, 2, 0, pub, prv


def test_ac3():
    d = [0, 4, 5, 0, 15, 4, 11, 10]                                     # 0x0450f4ba, docs_test.go:228-249
    m = [2, 0, 0, 0, 2, 1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 1]
    return lookup_circuit(d, m, "TestAC3")


def test_ac4():
    return lookup_circuit([1], [0, 1], "TestAC4")                       # docs_test.go:369-494


def wide_lookup(nd=96, np_=16):
    """The TestAC3 construction at a size where the commitments are real MSMs (Nv = 2*nd + np_)."""
    d = [(7 * i + 3) % np_ for i in range(nd)]
    m = [d.count(j) for j in range(np_)]
    return lookup_circuit(d, m, f"lookup_{nd}x{np_}")


def with_generators(case, seed):
    """Adds G, GVec, HVec from the seeded reader in the reference's order; returns (pub, prv, reader)."""
    name, pre, post, pub, prv = case
    rd = r.ByteStream(seed)
    r.values_bppp(rd, pre)
    pub = dict(pub)
    pub["G"] = r.points(rd, 1)[0]
    pub["GVec"] = r.points(rd, pub["Nm"])
    pub["HVec"] = r.points(rd, 9 + pub["Nv"])
    r.values_bppp(rd, post)
    return pub, prv, rd


CASES = [test_ac_protocol, test_arithmetic_circuit2, test_ac3, test_ac4]
