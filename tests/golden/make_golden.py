"""Generates tests/golden/bn256_g1.json from the big-int Python restatement
(oracle/bn256_ref.py).  Run once in the build container:  python tests/golden/make_golden.py

The reference itself cannot be executed here (no Go toolchain, cloudflare/bn256 not
vendored), so these vectors pin the C restatement and the device library to the Python
restatement -- "parity unpinned" with respect to the Go library, see DESIGN.md."""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, HERE)
import bn256_ref as r  # noqa: E402
from cases import CASES, apply_edits  # noqa: E402


def main():
    out = {"curve": {"p": hex(r.P), "n": hex(r.N), "b": 3, "g0": r.marshal(r.G0).hex()}}
    ks = [0, 1, 2, 3, 4, 5, 0xFFFF, 0x10000, 1 << 128, r.N - 2, r.N - 1, r.N, r.N + 1, (1 << 256) - 1,
          r.synth_scalar(0xA5A5, 0), r.synth_scalar(0xA5A5, 1)]
    out["kg"] = [{"k": hex(k), "point": r.marshal(r.mul(k, r.G0)).hex()} for k in ks]
    for e in out["kg"][:6]:
        k = int(e["k"], 16)
        assert r.marshal(r.mul_affine_slow(k, r.G0)).hex() == e["point"]
    out["keccak256"] = [
        {"msg": m.hex(), "digest": r.keccak256(m).hex()}
        for m in [b"", b"abc", bytes(135), bytes(136), bytes(range(200)), b"\xff" * 272]
    ]
    assert out["keccak256"][0]["digest"] == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert out["keccak256"][1]["digest"] == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"

    # synthetic generators: first elements + digests
    t0 = time.time()
    sb = r.synth_bases(0x5EED0001, 1024)
    ss = r.synth_scalars(0x5EED0002, 1024)
    out["synth"] = {
        "seed_bases": hex(0x5EED0001), "seed_scalars": hex(0x5EED0002),
        "bases_head": [r.marshal(p).hex() for p in sb[:4]],
        "scalars_head": [hex(s) for s in ss[:4]],
        "bases_sha256_1024": hashlib.sha256(b"".join(r.marshal(p) for p in sb)).hexdigest(),
        "scalars_sha256_1024": hashlib.sha256(b"".join(s.to_bytes(32, "big") for s in ss)).hexdigest(),
    }
    print("synth", time.time() - t0, flush=True)

    msm = []
    cache = {}
    for name, n, sbs, sss, edits in CASES:
        t0 = time.time()
        key = (sbs, n)
        if key not in cache:
            cache[key] = [r.marshal(p) for p in r.synth_bases(sbs, n)]
        bases = list(cache[key])
        scalars = r.synth_scalars(sss, n)
        apply_edits(bases, scalars, edits)
        pts = [r.unmarshal(b) for b in bases]
        if n <= 1024:
            res = r.vector_point_scalar_mul(pts, scalars)           # the reference loop
            assert res == r.msm_pippenger(pts, scalars)
        else:
            res = r.msm_pippenger(pts, scalars, c=12)
            assert res == r.msm_pippenger(pts, scalars, c=9)
        msm.append({"name": name, "n": n, "result": r.marshal(res).hex(),
                    "inputs_sha256": hashlib.sha256(b"".join(bases) + b"".join(s.to_bytes(32, "big") for s in scalars)).hexdigest()})
        print(name, time.time() - t0, flush=True)
    out["msm"] = msm

    # element-wise helpers on 8 points
    g = sb[:8]
    a = ss[:8]
    x = ss[8]
    out["elementwise"] = {
        "mul_on_scalar": [r.marshal(p).hex() for p in r.vector_point_mul_on_scalar(g, x)],
        "mul_each": [r.marshal(r.mul(k, p)).hex() for k, p in zip(a, g)],
        "hadamard_0_4": [r.marshal(p).hex() for p in r.hadamard_point_mul(g[:4], g[4:])],
        "fold_x_xinv": [r.marshal(p).hex() for p in r.hadamard_point_mul(
            r.vector_point_mul_on_scalar(g[:4], r.finv(x)), r.vector_point_mul_on_scalar(g[4:], x))],
        "x": hex(x),
    }

    # Bulletproofs transcripts with injected randomness (SURVEY 3.4 order)
    trs = []
    for n, v, seed in [(4, 11, 0xB0B0), (8, 200, 0xB0B1), (64, 11, 0xB0B2)]:
        t0 = time.time()
        rd = r.ByteStream(seed)
        pub = r.BulletProofPublic.new(rd, n)
        prv = r.rand_int(rd)
        pr = r.bp_prove(pub, v, prv, rd)
        assert r.bp_verify(pub, pr) is None
        trs.append({
            "n": n, "v": v, "seed": hex(seed), "prv": hex(prv),
            "G": [r.marshal(p).hex() for p in pub.inner.G], "H": [r.marshal(p).hex() for p in pub.inner.H],
            "U": r.marshal(pub.inner.U).hex(), "g": r.marshal(pub.G).hex(), "h": r.marshal(pub.H).hex(),
            "proof": {k: (r.marshal(pr[k]).hex() if k in ("V", "A", "S", "T1", "T2") else
                          [r.marshal(p).hex() for p in pr[k]] if k in ("L", "R") else hex(pr[k]))
                      for k in ("V", "A", "S", "T1", "T2", "tx", "taux", "nu", "L", "R", "a", "b", "y", "z", "x")},
        })
        print("bp", n, time.time() - t0, flush=True)
    out["bulletproofs"] = trs

    with open(os.path.join(HERE, "bn256_g1.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("written")


if __name__ == "__main__":
    main()
