#!/usr/bin/env python3
"""Headline benchmark: BN256 G1 multi-scalar multiplication (the reference's
vectorPointScalarMul, go/bp/util.go:165-175) on MI355X.

  python bench.py [--gpus N --steps K --warmup W] [--log2n 20]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one complete MSM: every rank runs the device pipeline on its contiguous slice of
2^log2n points (bases resident in HBM with their fixed-base table, scalars resident in HBM
when the timed region starts), the 64-byte partial sums are exchanged with one RCCL
all_gather, and every rank adds them on the host.  value = total points / second.
Steps are issued through bn256_msm_submit / bn256_msm_wait with --depth (default 3) calls in
flight, the way a prover issues its independent commitments (A and S, L and R): step i+1 is
enqueued before step i's result is collected, and all K results are collected inside the timed
region.  --depth 1 is the strictly blocking sequence; its rate is also reported
("blocking_call", measured after the timed region).
Weak scaling: the per-GPU slice is fixed (2^20 by default = BASELINE.json configs[2]); with
--log2n 21 --gpus 8 the job is exactly configs[4] (2^24 points over 8 GPUs).

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement" for the roofline accounting.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IMAD_PER_POINT = 23120          # 17 windows x 10 modmul x 136 IMAD
IMAD_PEAK = 3.47e13             # measured v_mad_u64_u32 lane-ops/s, profiles/r01_ubench_valu.txt
SEED_BASES, SEED_SCALARS = 0x5EED0001, 0x5EED0002


def cpu_baseline(sample_terms: int):
    """Reference algorithm (per-term double-and-add, then add), C restatement, 1 thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_c import OracleC
    orc = OracleC()
    bases = orc.synth_bases(SEED_BASES, sample_terms)
    scalars = orc.synth_scalars(SEED_SCALARS, sample_terms)
    t0 = time.perf_counter()
    out = orc.msm(bases, scalars, mt=False)
    dt = time.perf_counter() - t0
    return sample_terms / dt, out, orc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log2n", type=int, default=20, help="points per GPU = 2^log2n")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-precompute", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1 << 17)
    ap.add_argument("--depth", type=int, default=3, choices=[1, 2, 3], help="MSM calls in flight per GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # BENCH_FORCE_DIST=1 runs the multi-GPU code path (process group, all_gather of partials, barrier)
    # with a single rank, to rehearse it on a one-GPU box
    use_dist = world > 1 or bool(os.environ.get("BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import crypto_amd
    from crypto_amd import sharding
    eng = crypto_amd.Engine(local_rank)
    eng.set_tuning(args.window_bits, args.chunk)
    n = 1 << args.log2n
    first = rank * n
    flags = 0 if args.no_precompute else crypto_amd.PRECOMPUTE
    t0 = time.perf_counter()
    bases = eng.random_bases(SEED_BASES, n, first=first, flags=flags)
    setup_s = time.perf_counter() - t0
    scalars = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    eng.random_scalars_into(SEED_SCALARS, n, scalars.data_ptr(), first=first)

    ex = sharding.AsyncExchange(world, dev) if use_dist else None
    host_us = [0.0] * 3  # host time in submit / wait / exchange (start, finish of an older one, sum)

    def run(steps, depth, phase=None):
        def wait(ticket):
            p = eng.msm_wait(ticket)
            if phase is not None:
                phase[:] = [a + b for a, b in zip(phase, eng.last_timing())]
            return p

        return sharding.run_pipelined(steps, depth, lambda: eng.msm_submit(bases, scalars.data_ptr(), n), wait, ex, host_us)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.depth, args.depth)  # set-up: every lane allocates its workspaces
    result = run(args.warmup, args.depth)
    sync()
    host_us[:] = [0.0] * 3
    t0 = time.perf_counter()
    result = run(args.steps, args.depth)
    sync()
    dt = time.perf_counter() - t0
    host_timed = [x / max(args.steps, 1) for x in host_us]
    # The same K steps as strictly blocking calls (not part of `value`).  The per-phase HIP events and
    # the roofline's kernel time are taken here: with several calls in flight an event pair also spans
    # the time a kernel waits behind the other call's kernels, which is not kernel time.
    phase = [0.0] * 7
    t0 = time.perf_counter()
    result_blocking = run(args.steps, 1, phase)
    sync()
    dt_blocking = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        steps = max(args.steps, 1)
        traffic = executed = None
        try:  # HBM bytes per k_accumulate launch from the committed rocprofv3 PMC passes (2^20, c=16 only)
            if args.log2n == 20 and not args.no_precompute and not args.window_bits:
                with open(os.path.join(ROOT, "profiles", "r01_pmc_2p20.json")) as f:
                    pmc = json.load(f)["k_accumulate"]
                traffic = (pmc["FETCH_SIZE_KB"] + pmc["WRITE_SIZE_KB"]) * 1024.0
                # executed work next to the canonical 23 120 IMAD/point ("report executed
                # IMADs so waste is visible"): VALU lane-operations of all kinds per point
                executed = pmc["SQ_INSTS_VALU"] * 64.0 / n
        except Exception:
            traffic = executed = None
        phase = [p / steps for p in phase]
        acc_s = phase[3] * 1e-3
        achieved = n * IMAD_PER_POINT / acc_s if acc_s > 0 else 0.0
        line = {
            "metric": "MSM points/sec (BN256 G1, bit-exact with the reference's normalised affine output)",
            "value": n * world * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32x9 (29-bit digits, v_mad_i64_i32)",
            "data": "synthetic (seeded k_i*G0 bases, uniform scalars mod n)",
            "config": {
                "workload": f"single BN256 G1 MSM, 2^{args.log2n} points per GPU x {world} GPU(s) "
                            f"(BASELINE.json configs[2] at log2n=20; configs[4] at log2n=21, 8 GPUs)",
                "points_per_gpu": n,
                "fixed_base_table": not args.no_precompute,
                "parallelism": f"point-slices x{world}, RCCL all_gather of 64-byte partials",
                "calls_in_flight": args.depth,
                "setup_seconds_bases": setup_s,
                "result_xy": result.hex(),
            },
            # per call, HIP events on the call's stream, over the K blocking calls that follow the timed region
            "phases_ms": {"sort_pass_a": phase[0], "scan": phase[1], "sort_pass_b": phase[2], "accumulate": phase[3],
                          "fix": phase[4], "reduce": phase[5], "device_total": phase[6]},
            "host_us_per_step": {"submit": host_timed[0], "wait": host_timed[1], "exchange": host_timed[2]},
            "blocking_call": {"ms_per_call": dt_blocking / steps * 1e3, "points_per_s": n * world * args.steps / dt_blocking,
                              "same_result": bool(result_blocking == result)},
            "roofline": {
                "bound": "valu_int",
                "kernel": "k_accumulate",
                "kernel_ms": phase[3],
                "measured_over": f"{args.steps} blocking calls after the timed region (one call on the chip at a time)",
                "achieved": achieved,
                "peak": IMAD_PEAK,
                "unit": "IMAD/s",
                "frac": achieved / IMAD_PEAK,
                "traffic": traffic,
                "executed_valu_lane_ops_per_point": executed,
                "whole_msm_frac": (n * IMAD_PER_POINT / (dt / steps)) / IMAD_PEAK,
            },
        }
        if not args.no_precompute and n >= (1 << 18):
            # HBM rate of the sort / scatter phases (two-pass sort, 16 windows): ALGORITHMIC bytes -- 32-byte
            # scalars in, one 4-byte record per (point, window) out, then records in, sorted entries out --
            # over the live phase times; the PMC counters are in profiles/ and DESIGN.md 6
            ent = 16 * n * 4
            line["sort_hbm"] = {
                "pass_a": {"bytes": 32 * n + ent, "ms": phase[0], "GB/s": (32 * n + ent) / (phase[0] * 1e-3) / 1e9},
                "pass_b": {"bytes": 2 * ent, "ms": phase[2], "GB/s": 2 * ent / (phase[2] * 1e-3) / 1e9},
                "peak_GB/s": 8000.0,
            }
        if world == 1 and not args.no_cpu_baseline:
            v, _, orc = cpu_baseline(args.cpu_sample)
            # the oracle as checker: exact answer of the whole timed workload from its closed form
            # (bases are k_i*G0 with known k_i), bit-for-bit against the device result
            exact = orc.synth_msm_closed_form(SEED_BASES, SEED_SCALARS, n)
            line["cpu_baseline"] = {
                "value": v, "unit": "points/s", "cores": 1, "kind": "port",
                "device_result_matches_oracle_closed_form": bool(exact == result),
                "sample": f"first {args.cpu_sample} terms of the same seeded workload, reference algorithm "
                          f"(per-term double-and-add + add, go/bp/util.go:165-175) restated in C, 1 thread",
            }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
