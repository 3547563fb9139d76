"""Extracts the one multi-entry known answer the reference's own tests hold for this path: the 32 Fr values of
Wl*w that TestAC3 printed and its author pasted as a comment, go/bppp/docs_test.go:365 (challenge e = 2, the
constant values() returns, go/bppp/util.go:21).  Output: tests/golden/ref_wlw_ac3.json -- data only (decimal
integers), no reference source text.  Run once in the build container (the reference is not on the GPU box):
    python tests/golden/make_ref_wlw.py
Lines :366 and :492-493 are stale prints and are NOT fixtures: :366 comes from an older revision of the circuit
(entries 9, 10, 12-15 and the last six differ from any e under today's matrices: entry 8 gives
e = -1/x = 9771...4764, but entry 9 = -1/(4 + e) does not hold), and :492-493 are the TestAC3 comment pasted under
TestAC4, whose Wl*w has 4 entries."""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/go/bppp/docs_test.go"


def main():
    lines = open(SRC).read().split("\n")
    line = lines[365 - 1]
    assert line.lstrip().startswith("//Wl*w = ["), line[:40]
    vals = re.findall(r"\d+", line.split("[", 1)[1])
    assert len(vals) == 32
    out = {"source": "go/bppp/docs_test.go:365 (TestAC3, e = 2)", "wl_times_w": vals}
    with open(os.path.join(HERE, "ref_wlw_ac3.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote ref_wlw_ac3.json:", len(vals), "values")


if __name__ == "__main__":
    main()
