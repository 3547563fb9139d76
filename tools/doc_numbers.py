#!/usr/bin/env python3
"""Regenerates the measured-number blocks of DESIGN.md and README.md from the committed profile files, so
the prose can never drift from the records:

  python tools/doc_numbers.py [rNN]        (default: the newest profiles/rNN_bench_2p20.json)

Everything between `<!-- numbers:begin -->` and `<!-- numbers:end -->` in the two files is replaced."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else sorted(
    os.path.basename(f)[:3] for f in glob.glob(os.path.join(P, "r*_bench_2p20.json")))[-1]
bench = json.load(open(os.path.join(P, f"{tag}_bench_2p20.json")))
pmc = json.load(open(os.path.join(P, f"{tag}_pmc_2p20.json")))


def stats(name):
    out = {}
    with open(os.path.join(P, name), newline="") as f:
        for r in csv.DictReader(f):
            k = re.sub(r"^void ", "", re.sub(r"\(.*", "", r["Name"]))
            out[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return out


d1 = stats(f"{tag}_rocprofv3_kernel_stats_bench_2p20_depth1.csv")
d3 = stats(f"{tag}_rocprofv3_kernel_stats_bench_2p20.csv")
n = bench["config"]["points_per_gpu"]
ph, sc, rf = bench["phases_ms"], bench["single_call"], bench["roofline"]
acc = pmc.get("k_accumulate<true>") or pmc["k_accumulate"]
adds = acc["SQ_INSTS_VALU"] and 13 * n  # one mixed addition per (point, window), minus zero digits (2^-20 of them)
L = []
L.append(f"Source: `profiles/{tag}_bench_2p20.json` (`python3 bench.py --gpus 1 --steps {bench['steps']} --warmup {bench['warmup']}`, the driver's command, on one MI355X), "
         f"`profiles/{tag}_pmc_2p20.json`, `profiles/{tag}_rocprofv3_kernel_stats_bench_2p20*.csv`; regenerate with "
         f"`tools/refresh_profiles.sh` on the GPU box, then `tools/pmc_summary.py` and `tools/doc_numbers.py` here.")
L.append("")
L.append("| quantity (2²⁰ points, fixed-base table, default tuning: 13 windows of 20 bits) | value |")
L.append("|---|---|")
L.append(f"| `value`: {bench['config'].get('calls_in_flight', 3)} calls in flight ({bench['config'].get('hw_queues', 4)} hardware queues), scalars resident | **{bench['value'] / 1e6:.0f} Mpts/s** ({bench['ms_per_step']:.3f} ms per step, whole-MSM fraction {rf['whole_msm_frac']:.3f}) |")
L.append(f"| `single_call`: one blocking `bn256_msm_dev`, median of {sc['calls']} | {sc['median_ms']:.3f} ms = {sc['points_per_s'] / 1e6:.0f} Mpts/s, whole-MSM fraction **{sc['whole_msm_frac']:.3f}** |")
L.append(f"| phases of a blocking call (HIP events, ms) | pass A {ph['sort_pass_a']:.3f}, pass B {ph['sort_pass_b']:.3f}, accumulate {ph['accumulate']:.3f}, fix {ph['fix']:.3f}, reduce {ph['reduce']:.3f}, device total {ph['device_total']:.3f} |")
L.append(f"| `roofline` (`k_accumulate`, 2²⁰ × 23 120 IMAD per launch) | {rf['kernel_ms']:.3f} ms → {rf['achieved']:.3e} IMAD/s = **{rf['frac']:.3f}** of {rf['peak']:.2e}" +
         (f" ({rf['frac_vs_r01_peak']:.3f} of the 3.47e13 that rounds 1 and 2 priced against)" if 'frac_vs_r01_peak' in rf else "") + " |")
if "k_accumulate<true>" in d1:
    L.append(f"| rocprofv3 kernel-trace average of `k_accumulate`, `--depth 1` / default command | {d1['k_accumulate<true>'][1] / 1e3:.3f} ms over {d1['k_accumulate<true>'][0]} launches / {d3['k_accumulate<true>'][1] / 1e3:.3f} ms over {d3['k_accumulate<true>'][0]} |")
L.append(f"| `k_accumulate` PMC per launch | `SQ_INSTS_VALU` {acc['SQ_INSTS_VALU'] / 1e6:.1f} M = {acc['SQ_INSTS_VALU'] * 64 / adds:.0f} per addition, {acc['SQ_INSTS_VALU'] * 64 / n:.0f} lane-ops per point (canonical 23 120: ×{acc['SQ_INSTS_VALU'] * 64 / n / 23120:.2f}); FETCH {acc['FETCH_SIZE_KB'] * 1024 / 1e9:.2f} GB + WRITE {acc['WRITE_SIZE_KB'] * 1024 / 1e9:.2f} GB = `traffic` {rf['traffic'] / 1e9:.2f} GB ({rf['traffic'] / (rf['kernel_ms'] * 1e-3) / 1e12:.2f} TB/s) |")
if "sort_hbm" in bench:
    sh = bench["sort_hbm"]
    L.append(f"| sort passes, algorithmic bytes over phase time | pass A {sh['pass_a']['GB/s'] / 1e3:.2f} TB/s, pass B {sh['pass_b']['GB/s'] / 1e3:.2f} TB/s |")
if "pcie_inclusive" in bench:
    pc = bench["pcie_inclusive"]
    L.append(f"| `pcie_inclusive`: `bn256_msm`, scalars in host memory | pageable {pc['pageable']['median_ms']:.2f} ms = {pc['pageable']['points_per_s'] / 1e6:.0f} Mpts/s, pinned {pc['pinned']['median_ms']:.2f} ms = {pc['pinned']['points_per_s'] / 1e6:.0f} Mpts/s |")
if "bulletproofs_like" in bench:
    bp = bench["bulletproofs_like"]
    L.append(f"| `bulletproofs_like` (aL / aR scalars) | {bp['median_ms']:.3f} ms = {bp['points_per_s'] / 1e6:.0f} Mpts/s |")
if "n_2p24" in bench:
    b = bench["n_2p24"]
    L.append(f"| `n_2p24`: 2²⁴ points on this one GPU ({b.get('passes_per_call', '?')} passes, one reduction) | {b['ms_per_step']:.2f} ms per step = **{b['points_per_s'] / 1e6:.0f} Mpts/s** {b.get('calls_in_flight', 3)} in flight; blocking {b['single_call_median_ms']:.2f} ms = {b['single_call_points_per_s'] / 1e6:.0f} Mpts/s; whole-MSM fraction {b['whole_msm_frac']:.3f}; closed form {'matches' if b.get('matches_oracle_closed_form') else 'NOT CHECKED'} |")
if "multi_device_ctx" in bench and "median_ms" in bench["multi_device_ctx"]:
    m = bench["multi_device_ctx"]
    L.append(f"| `multi_device_ctx`: `bn256_ctx_create_multi`, devices {m['devices']}, 2²⁴ points | {m['median_ms']:.2f} ms = {m['points_per_s'] / 1e6:.0f} Mpts/s ({m.get('note', 'one shard per device')}); same bytes as the rank path: {m.get('same_result_as_rank_path')} |")
if "no_table_2p20" in bench:
    b = bench["no_table_2p20"]
    L.append(f"| `no_table_2p20`: one blocking table-less call (the reference's loop has no precomputation) | {b['median_ms']:.3f} ms = {b['points_per_s'] / 1e6:.0f} Mpts/s, whole-MSM fraction {b['whole_msm_frac']:.3f}; same bytes as the fixed-base call: {b.get('same_result_as_fixed_base')} |")
if "h1_2p20" in bench and "fold_2p19" in bench:
    h, f = bench["h1_2p20"], bench["fold_2p19"]
    L.append(f"| `h1_2p20` (a9: 2²⁰ per-point scalar multiplications, scalars uploaded inside the call) / `fold_2p19` (a7: 2¹⁹ fold outputs) | {h['median_ms']:.2f} ms = {h['points_per_s'] / 1e6:.0f} M points/s, issue fraction {h['imad_issue_frac']:.2f} / {f['median_ms']:.2f} ms = {f['outputs_per_s'] / 1e6:.0f} M outputs/s, issue fraction {f['imad_issue_frac']:.2f}; sampled outputs match the closed form: {h.get('sampled_outputs_match_closed_form')}, {f.get('sampled_outputs_match_closed_form')} |")
if "bp_n64" in bench:
    b = bench["bp_n64"]
    L.append(f"| `bp_n64` (configs[0]: go/bp range proof, N = 64) | prove {b['prove_median_ms']:.2f} ms, verify {b['verify_median_ms']:.2f} ms (medians of {b['runs']}); verify accepts: {b['verify_accepts']}, tampered proof rejected: {b['tampered_rejected']} |")
if "bppp_2p20" in bench:
    b = bench["bppp_2p20"]
    w = b.get("wnla_2p20", {})
    L.append(f"| `bppp_2p20` (configs[3]: BP++ commitments of {b['terms_per_commitment']} terms; WNLA at \|l\| = 2²⁰) | {b['blocking']['ms_per_commitment']:.2f} ms per commitment blocking ({b['blocking']['points_per_s'] / 1e6:.0f} Mpts/s), {b['on_lanes']['ms_per_commitment']:.2f} ms on lanes; closed form: {b.get('matches_oracle_closed_form')}; WNLA {w.get('median_ms', float('nan')):.2f} ms, {w.get('rounds')} rounds, final check passes: {w.get('final_check_passes')} |")
if "cpu_baseline" in bench:
    c = bench["cpu_baseline"]
    ac = c.get("all_cores")
    extra = f"; all {ac['cores']} host cores (not the reference, which is single-goroutine): {ac['value']:.0f} points/s" if ac else ""
    L.append(f"| `cpu_baseline` (C restatement of the reference loop, {c['cores']} thread; host: {c.get('host_cpu', {}).get('model_name', '?')}, {c.get('host_cpu', {}).get('os_cpu_count', '?')} CPUs) | {c['value']:.0f} points/s{extra} |")
L.append("")
L.append("| kernel (rocprofv3 kernel trace, `--depth 1`: one call on the chip) | launches per call | average µs |")
L.append("|---|---|---|")
calls = d1.get("k_accumulate<true>", (1, 0))[0]
for k in ("k_part<20, true>", "k_group_sort_tl", "k_accumulate<true>", "k_fix_boundary", "k_heavy", "k_fold_lane2", "k_fold_lane", "k_fold", "k_pack_results"):
    if k in d1:
        L.append(f"| `{k}` | {d1[k][0] / calls:.0f} | {d1[k][1]:.1f} |")
block = "\n".join(L)
for name in ("DESIGN.md", "README.md"):
    path = os.path.join(ROOT, name)
    s = open(path).read()
    if "<!-- numbers:begin -->" not in s:
        print(name, ": no marker, skipped")
        continue
    s = re.sub(r"<!-- numbers:begin -->.*?<!-- numbers:end -->",
               lambda _m: "<!-- numbers:begin -->\n" + block + "\n<!-- numbers:end -->", s, flags=re.S)
    open(path, "w").write(s)
    print(name, ": block written from", tag)
