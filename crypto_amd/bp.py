"""ctypes binding of libbpamd.so (include/bp_amd.h): the C++ host-side mirror of go/bp."""
from __future__ import annotations

import ctypes as C
import os

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BPAMD_LIB") or os.path.join(_HERE, "libbpamd.so")  # override: A/B builds
EXPORTS = ["bp_schnorr_msg", "bp_schnorr_multisig", "bp_schnorr_sign_det", "bp_schnorr_verify_batch",
           "bp_veca_encrypt_det", "bp_veca_verify", "bp_veca_decrypt", "bp_setup", "bp_prove", "bp_verify", "bp_prove_det", "bp_verify_det", "bp_inner_product_roundtrip", "bp_keccak256", "bp_last_error",
           "bp_wnla_det", "bp_wnla_synthetic", "bp_ac_det", "bp_ac_wlw"]
_bp = None


class AcCircuit(C.Structure):
    """bp_ac_circuit of include/bp_amd.h."""
    _fields_ = [(k, C.c_int) for k in ("Nm", "Nl", "Nv", "Nw", "No", "K", "Fl", "Fm")] + \
               [(k, C.c_char_p) for k in ("Wm", "Wl", "Am", "Al", "v", "sv", "wl", "wr", "wo")] + \
               [(k, C.POINTER(C.c_int32)) for k in ("f_lo", "f_ll", "f_lr", "f_no")]


def load() -> C.CDLL:
    global _bp
    if _bp is not None:
        return _bp
    _lib.load()  # libbn256msm first (and torch's HIP runtime before it)
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m crypto_amd.build`")
    lib = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    lib.bp_setup.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_char_p, C.c_char_p, C.c_char_p]
    lib.bp_prove.argtypes = [vp, C.c_int, vp, vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, sz,
                             C.POINTER(sz)]
    lib.bp_verify.argtypes = [vp, C.c_int, vp, vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, sz]
    lib.bp_prove_det.argtypes = [vp, C.c_int, C.c_char_p, C.c_uint64, C.c_char_p, sz, C.POINTER(sz)]
    lib.bp_verify_det.argtypes = [vp, C.c_int, C.c_uint64, C.c_char_p, sz]
    lib.bp_inner_product_roundtrip.argtypes = [vp, C.c_int, C.c_uint64, C.c_char_p, C.c_char_p, C.c_int]
    lib.bp_wnla_det.argtypes = [vp, C.c_uint64, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, sz, C.POINTER(sz)]
    lib.bp_wnla_synthetic.argtypes = [vp, C.c_uint64, sz, sz, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_int)]
    lib.bp_ac_det.argtypes = [vp, C.c_uint64, C.c_int, C.c_int, C.POINTER(AcCircuit), C.c_char_p, sz, C.POINTER(sz)]
    if hasattr(lib, "bp_ac_wlw"):  # absent from an older build loaded through BPAMD_LIB (A/B runs)
        lib.bp_ac_wlw.argtypes = [C.POINTER(AcCircuit), C.c_char_p]
    cp = C.c_char_p
    lib.bp_schnorr_msg.argtypes = [cp, sz, cp]
    lib.bp_schnorr_multisig.argtypes = [cp, cp, cp, cp, cp, cp]
    lib.bp_schnorr_sign_det.argtypes = [vp, C.c_uint64, cp, cp, cp, cp, cp, cp]
    lib.bp_schnorr_verify_batch.argtypes = [vp, sz, cp, cp, cp, cp, cp, cp]
    lib.bp_veca_encrypt_det.argtypes = [vp, C.c_uint64, cp, cp, cp, cp, cp, cp, cp, cp]
    lib.bp_veca_verify.argtypes = [vp, cp, cp, cp, cp, cp]
    lib.bp_veca_decrypt.argtypes = [cp, cp, cp, cp, cp, cp, cp]
    lib.bp_keccak256.argtypes = [C.c_char_p, sz, C.c_char_p]
    lib.bp_last_error.restype = C.c_char_p
    _bp = lib
    return lib


class BulletProofPublic:
    """NewBulletProofPublic / Prove / Verify (go/bp/main.go:27-34, :81-249), the product form: generators
    resident on the device, U, g, h as Marshal() bytes, all randomness from the OS CSPRNG (bp_setup /
    bp_prove / bp_verify).  Proofs are randomised; see BulletProofs below for the seeded test harness."""

    def __init__(self, engine, n: int, G=None, H=None, U: bytes | None = None, g: bytes | None = None,
                 h: bytes | None = None):
        from .engine import Bases
        self.eng, self.lib, self.n = engine, load(), n
        if G is None:
            hg, hh = C.c_void_p(), C.c_void_p()
            bu, bg, bh = (C.create_string_buffer(64) for _ in range(3))
            rc = self.lib.bp_setup(engine.ctx, n, C.byref(hg), C.byref(hh), bu, bg, bh)
            if rc:
                raise RuntimeError(f"bp_setup rc={rc}: {self.lib.bp_last_error().decode()}")
            self.G, self.H, self.U, self.g, self.h = Bases(engine, hg), Bases(engine, hh), bu.raw, bg.raw, bh.raw
        else:
            self.G, self.H, self.U, self.g, self.h = G, H, U, g, h

    def prove(self, v: int, prv: int) -> bytes:
        cap = 64 * 5 + 32 * 5 + 4 + 128 * 64
        out = C.create_string_buffer(cap)
        ln = C.c_size_t(0)
        rc = self.lib.bp_prove(self.eng.ctx, self.n, self.G.h, self.H.h, self.U, self.g, self.h,
                               int(v).to_bytes(32, "big"), int(prv).to_bytes(32, "big"), out, cap, C.byref(ln))
        if rc:
            raise RuntimeError(f"bp_prove rc={rc}: {self.lib.bp_last_error().decode()}")
        return out.raw[:ln.value]

    def verify(self, proof: bytes) -> str | None:
        """None when accepted, else the reference's error string."""
        rc = self.lib.bp_verify(self.eng.ctx, self.n, self.G.h, self.H.h, self.U, self.g, self.h, proof, len(proof))
        return None if rc == 0 else self.lib.bp_last_error().decode()


class BulletProofs:
    """TEST HARNESS: NewBulletProofPublic / Prove / Verify (go/bp/main.go) with an injected, seeded reader
    (bp_prove_det / bp_verify_det).  The seed fixes generators (known discrete logs) and blinding alike and
    is shared with the verifier: reproducible transcripts for parity tests, neither sound nor hiding."""

    def __init__(self, engine):
        self.eng = engine
        self.lib = load()

    def prove(self, n: int, v: int, seed: int) -> bytes:
        cap = 64 * 5 + 32 * 5 + 4 + 128 * 64
        out = C.create_string_buffer(cap)
        ln = C.c_size_t(0)
        rc = self.lib.bp_prove_det(self.eng.ctx, n, int(v).to_bytes(32, "big"), seed, out, cap, C.byref(ln))
        if rc:
            raise RuntimeError(f"bp_prove_det rc={rc}: {self.lib.bp_last_error().decode()}")
        return out.raw[:ln.value]

    def verify(self, n: int, seed: int, proof: bytes) -> str | None:
        """None when accepted, else the reference's error string."""
        rc = self.lib.bp_verify_det(self.eng.ctx, n, seed, proof, len(proof))
        if rc == 0:
            return None
        return self.lib.bp_last_error().decode()

    def inner_product_roundtrip(self, n: int, seed: int, a, b, tamper: bool = False) -> str | None:
        rc = self.lib.bp_inner_product_roundtrip(
            self.eng.ctx, n, seed, b"".join(int(x).to_bytes(32, "big") for x in a),
            b"".join(int(x).to_bytes(32, "big") for x in b), int(tamper))
        if rc == 0:
            return None
        return self.lib.bp_last_error().decode()


class WNLA:
    """wnla / TestWNLA of go/bppp/docs_test.go:1242-1338 over the device library."""

    def __init__(self, engine):
        self.eng = engine
        self.lib = load()

    def run(self, seed: int, l, n) -> dict:
        cap = 1 << 16
        out = C.create_string_buffer(cap)
        ln = C.c_size_t(0)
        rc = self.lib.bp_wnla_det(self.eng.ctx, seed, len(l), len(n),
                                  b"".join(int(x).to_bytes(32, "big") for x in l),
                                  b"".join(int(x).to_bytes(32, "big") for x in n), out, cap, C.byref(ln))
        if rc not in (0, -21):
            raise RuntimeError(f"bp_wnla_det rc={rc}: {self.lib.bp_last_error().decode()}")
        raw = out.raw[:ln.value]
        pos = 64
        res = {"C": raw[:64].hex(), "ok": rc == 0, "rounds": []}
        k = int.from_bytes(raw[pos:pos + 4], "big")
        pos += 4
        for _ in range(k):
            res["rounds"].append((raw[pos:pos + 64].hex(), raw[pos + 64:pos + 128].hex()))
            pos += 128
        for key in ("l", "n"):
            k = int.from_bytes(raw[pos:pos + 4], "big")
            pos += 4
            res[key] = [int.from_bytes(raw[pos + 32 * i:pos + 32 * i + 32], "big") for i in range(k)]
            pos += 32 * k
        assert pos == len(raw)
        return res

    def synthetic(self, seed: int, nl: int, nn: int) -> dict:
        ms0, ms1, rounds = C.c_double(0), C.c_double(0), C.c_int(0)
        rc = self.lib.bp_wnla_synthetic(self.eng.ctx, seed, nl, nn, C.byref(ms0), C.byref(ms1), C.byref(rounds))
        if rc not in (0, -21):
            raise RuntimeError(f"bp_wnla_synthetic rc={rc}: {self.lib.bp_last_error().decode()}")
        return {"ok": rc == 0, "ms_setup": ms0.value, "ms_run": ms1.value, "rounds": rounds.value}


class ArithmeticCircuit:
    """ArithmeticCircuitProtocol of go/bppp/docs_test.go:496-1240 over the device library
    (bp_ac_det).  `pub` / `prv` are the dictionaries the oracle's ac_protocol takes, without the
    generators (they come from the seeded reader): Nm, Nl, Nv, Nw, No, K, Wm, Wl, Am, Al, Fl, Fm and
    v, sv, wl, wr, wo, f = {typ: [index or None]}."""

    def __init__(self, engine):
        self.eng = engine
        self.lib = load()

    def run(self, seed: int, pre_draws: int, post_draws: int, pub: dict, prv: dict) -> dict:
        n_order = 0x8fb501e34aa387f9aa6fecb86184dc212e8d8e12f82b39241a2ef45b57ac7261

        def sc(xs):
            return b"".join((int(x) % n_order).to_bytes(32, "big") for x in xs)

        def flat(m):
            return sc([x for row in m for x in row])

        def fmap(typ, n):
            return (C.c_int32 * n)(*[-1 if i is None else int(i) for i in prv["f"][typ][:n]])

        keep = [fmap(1, pub["Nv"]), fmap(2, pub["Nv"]), fmap(3, pub["Nv"]), fmap(4, pub["Nm"])]
        c = AcCircuit(pub["Nm"], pub["Nl"], pub["Nv"], pub["Nw"], pub["No"], pub["K"], int(bool(pub["Fl"])),
                      int(bool(pub["Fm"])), flat(pub["Wm"]), flat(pub["Wl"]), sc(pub["Am"]), sc(pub["Al"]),
                      flat(prv["v"]), sc(prv["sv"]), sc(prv["wl"]), sc(prv["wr"]), sc(prv["wo"]),
                      C.cast(keep[0], C.POINTER(C.c_int32)), C.cast(keep[1], C.POINTER(C.c_int32)),
                      C.cast(keep[2], C.POINTER(C.c_int32)), C.cast(keep[3], C.POINTER(C.c_int32)))
        cap = 1 << 20
        out = C.create_string_buffer(cap)
        ln = C.c_size_t(0)
        rc = self.lib.bp_ac_det(self.eng.ctx, seed, pre_draws, post_draws, C.byref(c), out, cap, C.byref(ln))
        if rc not in (0, -21):
            raise RuntimeError(f"bp_ac_det rc={rc}: {self.lib.bp_last_error().decode()}")
        raw, pos = out.raw[:ln.value], 0

        def pt():
            nonlocal pos
            pos += 64
            return raw[pos - 64:pos].hex()

        def u32():
            nonlocal pos
            pos += 4
            return int.from_bytes(raw[pos - 4:pos], "big")

        def frs():
            nonlocal pos
            k = u32()
            v = [int.from_bytes(raw[pos + 32 * i:pos + 32 * i + 32], "big") for i in range(k)]
            pos += 32 * k
            return v

        res = {"V": [pt() for _ in range(pub["K"])]}
        for k in ("Co", "Cl", "Cr", "V_", "Cs", "PT", "CT"):
            res[k] = pt()
        res["check_V"], res["check_CT"], res["check_45"] = (bool(b) for b in raw[pos:pos + 3])
        pos += 3
        res["vT"] = int.from_bytes(raw[pos:pos + 32], "big")
        pos += 32
        res["lT"], res["nT"], res["cT"] = frs(), frs(), frs()
        k = u32()
        res["rounds"] = [(pt(), pt()) for _ in range(k)]
        res["l"], res["n"] = frs(), frs()
        res["wnla_ok"] = rc == 0
        assert pos == len(raw)
        return res


def wl_times_w(pub: dict, prv: dict) -> list:
    """matrixMulOnVector(w, Wl), w = wl || wr || wo (go/bppp/util.go:311-319): the "Wl*w" the reference's circuit
    tests print (go/bppp/docs_test.go:326).  The C++ mirror's host Fr arithmetic (bp_ac_wlw); no GPU needed."""
    lib = load()
    n_order = 0x8fb501e34aa387f9aa6fecb86184dc212e8d8e12f82b39241a2ef45b57ac7261

    def sc(xs):
        return b"".join((int(x) % n_order).to_bytes(32, "big") for x in xs)

    c = AcCircuit()
    c.Nm, c.Nl, c.Nv, c.Nw, c.No, c.K = pub["Nm"], pub["Nl"], pub["Nv"], pub["Nw"], pub["No"], pub["K"]
    c.Wl, c.wl, c.wr, c.wo = sc([x for row in pub["Wl"] for x in row]), sc(prv["wl"]), sc(prv["wr"]), sc(prv["wo"])
    out = C.create_string_buffer(32 * pub["Nl"])
    rc = lib.bp_ac_wlw(C.byref(c), out)
    if rc:
        raise RuntimeError(f"bp_ac_wlw rc={rc}: {lib.bp_last_error().decode()}")
    return [int.from_bytes(out.raw[32 * i:32 * i + 32], "big") for i in range(pub["Nl"])]


def parse_proof(raw: bytes) -> dict:
    """Splits the wire format of include/bp_amd.h into named fields (hex for points, ints for scalars)."""
    pos = 0
    out = {}
    for k in ("V", "A", "S", "T1", "T2"):
        out[k] = raw[pos:pos + 64].hex()
        pos += 64
    for k in ("tx", "taux", "nu"):
        out[k] = int.from_bytes(raw[pos:pos + 32], "big")
        pos += 32
    cnt = int.from_bytes(raw[pos:pos + 4], "big")
    pos += 4
    out["L"] = [raw[pos + 64 * i:pos + 64 * i + 64].hex() for i in range(cnt)]
    pos += 64 * cnt
    out["R"] = [raw[pos + 64 * i:pos + 64 * i + 64].hex() for i in range(cnt)]
    pos += 64 * cnt
    out["a"] = int.from_bytes(raw[pos:pos + 32], "big")
    out["b"] = int.from_bytes(raw[pos + 32:pos + 64], "big")
    assert pos + 64 == len(raw)
    return out


def _b32(x: int) -> bytes:
    return int(x).to_bytes(32, "big")


class Schnorr:
    """go/schnorr-bn256/schnorr.go over libbpamd: Msg, MultiSigSchnorr, SignSchnorr (seeded r),
    VerifySchnorr batched on the device.  Points are Marshal() bytes, S the reference's unreduced int."""

    def __init__(self, engine):
        self.eng, self.lib = engine, load()

    def _chk(self, rc, what):
        if rc:
            raise RuntimeError(f"{what} rc={rc}: {self.lib.bp_last_error().decode()}")

    def msg(self, *chunks: bytes) -> int:
        data = b"".join(chunks)
        out = C.create_string_buffer(32)
        self._chk(self.lib.bp_schnorr_msg(data, len(data), out), "bp_schnorr_msg")
        return int.from_bytes(out.raw, "big")

    def multisig(self, prv: int, r: int, pub_common: bytes, r_common: bytes, m: int) -> tuple:
        out = C.create_string_buffer(64)
        self._chk(self.lib.bp_schnorr_multisig(_b32(prv), _b32(r), pub_common, r_common, _b32(m), out), "bp_schnorr_multisig")
        return r_common, int.from_bytes(out.raw, "big")

    def sign(self, prv: int, pub: bytes, G: bytes, m: int, seed: int) -> tuple:
        R, S = C.create_string_buffer(64), C.create_string_buffer(64)
        self._chk(self.lib.bp_schnorr_sign_det(self.eng.ctx, seed, _b32(prv), pub, G, _b32(m), R, S), "bp_schnorr_sign_det")
        return R.raw, int.from_bytes(S.raw, "big")

    def verify_batch(self, sigs, pubs, G: bytes, ms) -> list:
        n = len(sigs)
        ok = C.create_string_buffer(max(n, 1))
        self._chk(self.lib.bp_schnorr_verify_batch(
            self.eng.ctx, n, b"".join(s[0] for s in sigs), b"".join(int(s[1]).to_bytes(64, "big") for s in sigs),
            b"".join(pubs), G, b"".join(_b32(m) for m in ms), ok), "bp_schnorr_verify_batch")
        return [bool(ok.raw[i]) for i in range(n)]

    def verify(self, sig, pub: bytes, G: bytes, m: int) -> bool:
        return self.verify_batch([sig], [pub], G, [m])[0]


class VeCa:
    """go/ve-ca/main.go over libbpamd: Encrypt (seeded reader), Verify, Decrypt; proofs as the byte string
    described in include/bp_amd.h."""
    K = 128
    PROOF_BYTES = 10 * 128 * 32 + 128 * 64

    def __init__(self, engine):
        self.eng, self.lib = engine, load()

    def encrypt(self, v: int, r: int, s_r: int, g: int, GenG: bytes, GenH: bytes, seed: int) -> tuple:
        proof, Cc = C.create_string_buffer(self.PROOF_BYTES), C.create_string_buffer(64)
        rc = self.lib.bp_veca_encrypt_det(self.eng.ctx, seed, _b32(v), _b32(r), _b32(s_r), _b32(g), GenG, GenH, proof, Cc)
        if rc:
            raise RuntimeError(f"bp_veca_encrypt_det rc={rc}: {self.lib.bp_last_error().decode()}")
        return proof.raw, Cc.raw

    def verify(self, proof: bytes, s_r: int, Cc: bytes, GenG: bytes, GenH: bytes) -> str | None:
        """None when Verify returns true, else the message the reference prints."""
        if len(proof) != self.PROOF_BYTES:
            raise ValueError("proof length")
        rc = self.lib.bp_veca_verify(self.eng.ctx, proof, _b32(s_r), Cc, GenG, GenH)
        return None if rc == 0 else self.lib.bp_last_error().decode()

    def decrypt(self, u: int, proof: bytes, Cc: bytes, GenG: bytes, GenH: bytes) -> tuple:
        v, r = C.create_string_buffer(32), C.create_string_buffer(32)
        rc = self.lib.bp_veca_decrypt(_b32(u), proof, Cc, GenG, GenH, v, r)
        if rc:
            raise RuntimeError(self.lib.bp_last_error().decode())
        return int.from_bytes(v.raw, "big"), int.from_bytes(r.raw, "big")
