#!/usr/bin/env python3
"""profiles/r03_same_box_vs_r02.md from what tools/ab_rounds.sh left under gpurun_out/ (round 2's tree and this tree,
each running its own default bench.py and flow tools, alternating on ONE box)."""
import json
import os
import re
import statistics

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
O = os.path.join(ROOT, "gpurun_out")


def lines(r):
    return [json.loads(x) for x in open(os.path.join(O, f"ab_rounds_{r}.jsonl")) if x.strip()]


def rng(vals, fmt):
    vals = list(vals)
    return " / ".join(fmt.format(v) for v in vals)


r2, r3 = lines("r2"), lines("r3")
rows = [
    ("`value` (four calls in flight), Mpts/s", lambda j: j["value"] / 1e6, "{:.0f}"),
    ("`single_call` median, ms", lambda j: j["single_call"]["median_ms"], "{:.3f}"),
    ("`phases_ms.accumulate` = `roofline.kernel_ms`, ms", lambda j: j["phases_ms"]["accumulate"], "{:.3f}"),
    ("`phases_ms.device_total`, ms", lambda j: j["phases_ms"]["device_total"], "{:.3f}"),
    ("`phases_ms.reduce`, ms", lambda j: j["phases_ms"]["reduce"], "{:.3f}"),
    ("`pcie_inclusive` pinned, ms", lambda j: j["pcie_inclusive"]["pinned"]["median_ms"], "{:.3f}"),
    ("`pcie_inclusive` pageable, ms", lambda j: j["pcie_inclusive"]["pageable"]["median_ms"], "{:.3f}"),
    ("`n_2p24` in flight, ms per step", lambda j: j["n_2p24"]["ms_per_step"], "{:.2f}"),
    ("`n_2p24` one blocking call, ms", lambda j: j["n_2p24"]["single_call_median_ms"], "{:.2f}"),
    ("`bulletproofs_like`, ms", lambda j: j["bulletproofs_like"]["median_ms"], "{:.3f}"),
]
L = ["# Round 2's tree against round 3's on ONE box", "",
     "`tools/ab_rounds.sh`: `git archive 9c7408f` (the head VERDICT r2 judged) built with its own `crypto_amd/build.py`",
     "under `ab/r2tree/`, and this tree; each runs its own `python3 bench.py --steps 50 --warmup 5` and its own",
     "`tools/wnla_bench.py`, `tools/bp_bench.py`, `tools/perf_elementwise.py`, alternating, two rounds.  The boxes of the",
     "pool differ by several per cent (DESIGN.md 6), so this is the comparison between rounds that does not depend on",
     "which box a run drew.  Values of the two rounds of the script, in order.", "",
     "| bench line | round 2 | round 3 |", "|---|---|---|"]
for name, f, fmt in rows:
    L.append(f"| {name} | {rng(map(f, r2), fmt)} | {rng(map(f, r3), fmt)} |")
flows = open(os.path.join(O, "ab_rounds_flows.txt")).read().splitlines()


def pick(tag, key):
    out = []
    for x in flows:
        if x.startswith(tag + " ") and "{" in x:
            out.append(json.loads(x[x.index("{"):])[key])
    return out


def ew(tag, pat):
    return [float(m.group(1)) for x in flows if x.startswith(tag + " ew") for m in [re.search(pat, x)] if m]


PAT_SM, PAT_FOLD = r"scalar_mul_each\s+([0-9.]+)", r"fold to 2\^19\s+([0-9.]+)"
sm2 = rng(ew("r2", PAT_SM), "{:.2f}") or "(round 2's tool prints nothing for this size)"
sm3 = rng(ew("r3", PAT_SM), "{:.2f}")
fo2 = rng(ew("r2", PAT_FOLD), "{:.2f}") or "(the same)"
fo3 = rng(ew("r3", PAT_FOLD), "{:.2f}")
L += [f"| WNLA at \\|l\\| = 2^20 (`tools/wnla_bench.py`), ms | {rng(pick('r2 wnla', 'ms_run'), '{:.2f}')} | {rng(pick('r3 wnla', 'ms_run'), '{:.2f}')} |",
      f"| go/bp N = 64 prove, best of 5, set-up included (`tools/bp_bench.py`), ms | {rng(pick('r2 bp', 'prove_ms_min'), '{:.2f}')} | {rng(pick('r3 bp', 'prove_ms_min'), '{:.2f}')} |",
      f"| go/bp N = 64 verify, ms | {rng(pick('r2 bp', 'verify_ms_min'), '{:.2f}')} | {rng(pick('r3 bp', 'verify_ms_min'), '{:.2f}')} |",
      f"| h1 loop, 2^20 points (`tools/perf_elementwise.py 20`), ms | {sm2} | {sm3} |",
      f"| generator fold, 2^19 outputs, ms | {fo2} | {fo3} |",
      ""]
m = lambda f, js: statistics.mean(map(f, js))
L.append(f"Means: `value` ×{m(rows[0][1], r3) / m(rows[0][1], r2):.3f}, one blocking call ×{m(rows[1][1], r3) / m(rows[1][1], r2):.3f}, "
         f"`k_accumulate` ×{m(rows[2][1], r3) / m(rows[2][1], r2):.3f}, 2^24 in flight ×{m(rows[7][1], r3) / m(rows[7][1], r2):.3f}.")
open(os.path.join(ROOT, "profiles", "r03_same_box_vs_r02.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L))
