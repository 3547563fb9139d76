#!/usr/bin/env python3
"""Generates tests/golden/bppp_ac.json: transcripts of the BP++ arithmetic-circuit protocol
(go/bppp/docs_test.go:496-1240) on the reference's own test circuits (TestACProtocol,
TestArithmeticCircuit2, TestAC3, TestAC4: tests/ac_cases.py), computed by the Python restatement
oracle/bn256_ref.py:ac_protocol with a seeded byte stream standing in for crypto/rand.

The reference holds no literal point for these (parity unpinned); the fixture
pins the oracle against regressions and gives the device path a committed target.

  python tests/golden/make_ac_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))
import ac_cases  # noqa: E402
import bn256_ref as r  # noqa: E402

SEED = 0xAC00


def transcript_json(tr):
    h = lambda p: r.marshal(p).hex()
    return {
        "V": [h(p) for p in tr["V"]],
        **{k: h(tr[k]) for k in ("Co", "Cl", "Cr", "V_", "Cs", "PT", "CT")},
        "check_V": tr["check_V"], "check_CT": tr["check_CT"], "check_45": tr["check_45"],
        "vT": hex(tr["vT"]), "lT": [hex(x) for x in tr["lT"]], "nT": [hex(x) for x in tr["nT"]],
        "cT": [hex(x) for x in tr["cT"]],
        "wnla_ok": tr["wnla_ok"],
        "rounds": [[h(x), h(y)] for x, y in tr["rounds"]],
        "l": [hex(x) for x in tr["l"]], "n": [hex(x) for x in tr["n"]],
    }


def main():
    out = {"seed": SEED, "cases": {}}
    for mk in ac_cases.CASES:
        case = mk()
        pub, prv, rd = ac_cases.with_generators(case, SEED)
        out["cases"][case[0]] = transcript_json(r.ac_protocol(pub, prv, rd))
    with open(os.path.join(HERE, "bppp_ac.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote bppp_ac.json:", list(out["cases"]))


if __name__ == "__main__":
    main()
