"""Generates tests/golden/f4.json from the Python restatement (oracle/bn256_ref.py): the
go/schnorr-bn256 test scenario (TestSchnorrSignature, schnorr_test.go:16-80) and one go/ve-ca
Encrypt / Verify / Decrypt run (TestEverythingWorks, main_test.go:10-37) from fixed reader seeds.
Run once in the build container:  python tests/golden/make_f4_golden.py
The reference cannot be executed here (no Go toolchain): these vectors pin the C++ mirror and the
device library to the Python restatement -- parity unpinned with respect to the Go code."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import bn256_ref as r  # noqa: E402

SEED_SETUP, SEED_SIGN, SEED_ENC = 0xF400, 0xF401, 0xF402


def main():
    rd = r.ByteStream(SEED_SETUP)
    out = {"seeds": {"setup": SEED_SETUP, "sign": SEED_SIGN, "encrypt": SEED_ENC}}
    # ---- schnorr: G, alice / bob keys and nonces in the test's draw order
    _, G = r.random_g1(rd)
    a_prv, a_pub = r.schnorr_r(rd, G)
    b_prv, b_pub = r.schnorr_r(rd, G)
    a_r, a_R = r.schnorr_r(rd, G)
    b_r, b_R = r.schnorr_r(rd, G)
    Rc, Pub = r.add(a_R, b_R), r.add(a_pub, b_pub)
    m = r.schnorr_msg(b"Hello world")
    sa, sb = r.schnorr_multisig(a_prv, a_r, Pub, Rc, m), r.schnorr_multisig(b_prv, b_r, Pub, Rc, m)
    sig = (Rc, (sa[1] + sb[1]) % r.N)
    assert r.schnorr_verify(sig, Pub, G, m)
    single = r.schnorr_sign(a_prv, a_pub, G, m, r.ByteStream(SEED_SIGN))
    assert r.schnorr_verify(single, a_pub, G, m) and not r.schnorr_verify(single, b_pub, G, m)
    hx = lambda p: r.marshal(p).hex()  # noqa: E731
    out["schnorr"] = {
        "G": hx(G), "alice": {"prv": hex(a_prv), "pub": hx(a_pub), "r": hex(a_r), "R": hx(a_R)},
        "bob": {"prv": hex(b_prv), "pub": hx(b_pub), "r": hex(b_r), "R": hx(b_R)},
        "msg_hello_world": hex(m), "R_common": hx(Rc), "pub_common": hx(Pub),
        "alice_S": hex(sa[1]), "bob_S": hex(sb[1]), "combined_S": hex(sig[1]),
        "single": {"R": hx(single[0]), "S": hex(single[1])},
    }
    # ---- ve-ca
    g = r.rand_int(rd)
    _, GenG = r.random_g1(rd)
    _, GenH = r.random_g1(rd)
    u = r.rand_int(rd)
    share = pow(g, u, r.N)
    v, rr = r.rand_int(rd), r.rand_int(rd)
    proof, C = r.veca_encrypt(v, rr, share, g, GenG, GenH, r.ByteStream(SEED_ENC))
    assert r.veca_verify(proof, share, C, GenG, GenH) is None
    assert r.veca_decrypt(u, proof, C, GenG, GenH) == (v, rr)
    raw = r.veca_proof_bytes(proof)
    out["veca"] = {
        "g": hex(g), "GenG": hx(GenG), "GenH": hx(GenH), "receiver_prv": hex(u), "receiver_share": hex(share),
        "v": hex(v), "r": hex(rr), "C": hx(C), "proof_bytes": len(raw), "proof_sha256": hashlib.sha256(raw).hexdigest(),
        "X_first": [hx(x) for x in proof["X"][:4]], "X_sha256": hashlib.sha256(b"".join(r.marshal(x) for x in proof["X"])).hexdigest(),
        "E00": hex(proof["E"][0][0]), "Alpha00": hex(proof["Alpha"][0][0]), "Share10": hex(proof["Share"][1][0]),
    }
    with open(os.path.join(HERE, "f4.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote f4.json")


if __name__ == "__main__":
    main()
