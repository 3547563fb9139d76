#!/usr/bin/env python3
"""Headline benchmark: BN256 G1 multi-scalar multiplication (the reference's
vectorPointScalarMul, go/bp/util.go:165-175) on MI355X.

  python bench.py [--gpus N --steps K --warmup W] [--log2n 20 | --total-log2n 24]

`python bench.py --gpus N` with N > 1 works as a bare command: when WORLD_SIZE is unset the process
that was started is only a launcher -- it makes no GPU call, starts N fresh ranks of this script
(crypto_amd/launch.py: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), waits and
relays rank 0's single JSON line.  Under `python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N` the ranks are the launcher's and this script is a rank at once.

A "step" is one complete MSM: every rank runs the device pipeline on its contiguous slice of the
points (bases resident in HBM with their fixed-base table, scalars resident in HBM when the timed
region starts), the 64-byte partial sums are exchanged with an RCCL all_gather (one per --depth finished
steps: their partial sums travel together), and every rank adds them on the host.  value = total points / second.  Steps are issued through bn256_msm_submit /
bn256_msm_wait with --depth (default 4) calls in flight, the way a prover issues its independent
commitments (A and S, L and R): step i+1 is enqueued before step i's result is collected, and all K
results are collected inside the timed region.  Every call in flight runs on a HIP stream of its own; the
process asks for 8 hardware queues (GPU_MAX_HW_QUEUES, raised below if the environment's value is lower) so
that no two of them share one -- with HIP's default of 4, two of four lanes are served in order by one queue
and their calls do not overlap (tools/probe/stream_probe.hip).

  default            weak scaling: 2^log2n points per GPU (2^20 = BASELINE.json configs[2] per GPU)
  --total-log2n T    strong scaling: 2^T points in all, rank r takes the r-th contiguous slice
                     (--gpus 8 --total-log2n 24 is configs[4]; --gpus 1 --total-log2n 24 the same job
                     on one GPU, in eight passes)

Besides `value` the line carries, measured in the same run: `single_call` (SURVEY 8d's wording of
the metric: one blocking call, median; this leg and the phase-timed calls behind the roofline run BEFORE the
warm-up steps, behind `--settle-ms` of discarded calls -- `config.order`), `pcie_inclusive`
(scalars in host memory, never `value`),
`bulletproofs_like` (the aL / aR scalar distribution), `n_2p24` (2^24 points on this one GPU) at
N = 1, and `strong_2p24` (2^24 points over all ranks, with the same job on rank 0's GPU alone beside it:
`efficiency_vs_n1`) at N > 1.  At N = 1 also the other BASELINE.json configs and the element-wise rows, each
a median of >= 5 runs after a warm one, each with its exactness flag: `no_table_2p20` (the reference's loop
has no precomputation), `bp_n64` (configs[0]: go/bp prove / verify), `bppp_2p20` (configs[3]: four BP++
commitments + the WNLA at |l| = 2^20), `h1_2p20`, `fold_2p19` (go/bp/main.go:156-159, :355-356).
At N > 1 the line proves what took part: `ranks` holds world size, every rank's device (PCI bus id, UUID),
and each rank's k_accumulate time.

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement" for the roofline accounting.
"""
import argparse
import ctypes as C
import glob
import json
import os
import statistics
import sys
import time

# before anything loads the HIP runtime (see above); BENCH_HW_QUEUES=0 leaves the environment's value alone.
# A rank of a multi-GPU run has more streams (the exchange's side stream, RCCL's own): 16 queues there --
# rehearsed with one rank and the exchange forced, 794-810 Mpts/s on 8 queues, 841-847 on 16.
_queues = os.environ.get("BENCH_HW_QUEUES") or (
    "16" if int(os.environ.get("WORLD_SIZE") or 1) > 1 or os.environ.get("BENCH_FORCE_DIST") else "8")
if _queues != "0":
    os.environ["GPU_MAX_HW_QUEUES"] = str(max(int(os.environ.get("GPU_MAX_HW_QUEUES") or 4), int(_queues)))
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IMAD_PER_POINT = 23120          # 17 windows x 10 modmul x 136 IMAD
# Measured multiply-add issue rate of the chip, lane-ops/s: one asm block of dependent v_mad_i64_i32, four waves per
# SIMD, 1.714 ns per wave-instruction (tools/probe/vgpr_bank_probe.hip, profiles/r03_ubench_issue.txt) = 15.6 of the
# nominal 16 lanes/clk/SIMD at 2.4 GHz.  Rounds 1 and 2 priced against 3.47e13 (profiles/r01_ubench_valu.txt: one
# instruction per asm statement, which hipcc pads; 4.6 clk per instruction) -- kept as `frac_vs_r01_peak`.
IMAD_PEAK = 3.82e13
IMAD_PEAK_R01 = 3.47e13
SEED_BASES, SEED_SCALARS = 0x5EED0001, 0x5EED0002
N_ORDER = 0x8FB501E34AA387F9AA6FECB86184DC212E8D8E12F82B39241A2EF45B57AC7261


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-ms", type=int, default=150,
                    help="blocking calls (discarded) for this long before the blocking-call legs; the W warm-up steps and "
                         "the K timed steps follow those legs")
    ap.add_argument("--log2n", type=int, default=20, help="weak scaling: points per GPU = 2^log2n")
    ap.add_argument("--total-log2n", type=int, default=0, help="strong scaling: 2^T points over all GPUs")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-precompute", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip single_call / pcie / 2^24 / bulletproofs-like legs")
    ap.add_argument("--cpu-sample", type=int, default=1 << 17)
    ap.add_argument("--depth", type=int, default=4, choices=[1, 2, 3, 4, 5, 6, 7, 8], help="MSM calls in flight per GPU (at most BN256_ASYNC_LANES)")
    return ap.parse_args(argv)


def cpu_baseline(sample_terms: int):
    """Reference algorithm (per-term double-and-add, then add), C restatement, 1 thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_c import OracleC
    orc = OracleC()
    bases = orc.synth_bases(SEED_BASES, sample_terms)
    scalars = orc.synth_scalars(SEED_SCALARS, sample_terms)
    t0 = time.perf_counter()
    orc.msm(bases, scalars, mt=False)
    dt = time.perf_counter() - t0
    return sample_terms / dt, orc


def cpu_model() -> dict:
    """What the box says about its host CPU (a KVM guest here: the model string may be masked to "AMD EPYC",
    so family / model / stepping are reported beside it) and how many cores this process may use."""
    info = {}
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                k, _, v = ln.partition(":")
                k = k.strip()
                if k in ("model name", "cpu family", "model", "stepping", "cpu MHz") and k not in info:
                    info[k] = v.strip()
                if not ln.strip():
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = None
    return {"model_name": info.get("model name", "unknown"), "cpu_family": info.get("cpu family"),
            "model": info.get("model"), "stepping": info.get("stepping"), "cpu_mhz": info.get("cpu MHz"),
            "os_cpu_count": os.cpu_count(), "usable_cpus": usable, "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS")}


def cpu_baseline_all_cores(orc, sample_terms: int):
    """the same per-term loop split over the host's cores (OpenMP in the C restatement; the
    chunks' partial sums are added at the end).  Not the reference (which is one goroutine)."""
    try:
        threads = C.CDLL("libgomp.so.1").omp_get_max_threads()
    except OSError:
        threads = None
    bases = orc.synth_bases(SEED_BASES, sample_terms)
    scalars = orc.synth_scalars(SEED_SCALARS, sample_terms)
    t0 = time.perf_counter()
    r = orc.msm(bases, scalars, mt=True)
    dt = time.perf_counter() - t0
    return {"value": sample_terms / dt, "unit": "points/s", "cores": threads, "kind": "port",
            "label": "same algorithm on all host cores (OpenMP, 64 chunks) -- NOT the reference, which is single-goroutine",
            "sample": f"first {sample_terms} terms of the same seeded workload, {dt:.1f} s",
            "matches_closed_form": bool(r == orc.synth_msm_closed_form(SEED_BASES, SEED_SCALARS, sample_terms))}


def launcher(args) -> int:
    """The bare `python bench.py --gpus N` (N > 1, no WORLD_SIZE): no GPU call here, N fresh ranks."""
    from crypto_amd import launch
    script = os.environ.get("BENCH_CHILD_SCRIPT") or os.path.abspath(__file__)  # override: CPU test of this path
    return launch.launch_ranks(script, sys.argv[1:], args.gpus)


def worker(args) -> int:
    # stdout carries the ONE JSON line and nothing else: libraries that print there (RCCL's version
    # banner at communicator set-up, gloo's connection notice) are sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    flags = 0 if args.no_precompute else crypto_amd.PRECOMPUTE
    # one all_gather per `depth` finished steps (sharding.AsyncExchange: an exchange costs the loop ~80 us of host
    # time; rehearsed with one rank, 64 steps: 790 Mpts/s with one exchange per step, 865 with one per four, 887 without)
    ex = sharding.AsyncExchange(world, dev, batch=args.depth) if use_dist else None
    host_us = [0.0] * 3  # host time in submit / wait / exchange (start, finish of an older one, sum)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    class Workload:
        """This rank's slice [first, first + n) of a job of n_total synthetic points, resident in HBM."""

        def __init__(self, n_total, strong):
            self.n_total, self.strong = n_total, strong
            self.first, self.n = sharding.split_even(n_total, world, rank) if strong else (rank * (n_total // world), n_total // world)
            t0 = time.perf_counter()
            self.bases = eng.random_bases(SEED_BASES, self.n, first=self.first, flags=flags)
            self.setup_s = time.perf_counter() - t0
            self.scalars = torch.empty(max(self.n, 1) * 32, dtype=torch.uint8, device=dev)
            eng.random_scalars_into(SEED_SCALARS, self.n, self.scalars.data_ptr(), first=self.first)

        def run(self, steps, depth, phase=None):
            def submit():
                return eng.msm_submit(self.bases, self.scalars.data_ptr(), self.n)

            def wait(ticket):
                p = eng.msm_wait(ticket)
                if phase is not None:
                    phase[:] = [a + b for a, b in zip(phase, eng.last_timing())]
                return p

            return sharding.run_pipelined(steps, depth, submit, wait, ex, host_us)

        def blocking_calls(self, calls, phases=None):
            """`calls` strictly blocking steps (MSM, exchange, host sum), each timed on the host; `phases`
            (a list) receives every call's phase times (bn256_last_timing: summed over the call's passes)."""
            ts, res = [], None
            for _ in range(calls):
                t0 = time.perf_counter()
                one = [0.0] * 7 if phases is not None else None
                if ex is None:  # one GPU: the blocking entry point itself (bn256_msm_dev)
                    res = eng.msm_dev(self.bases, self.scalars.data_ptr(), self.n)
                    if one is not None:
                        one[:] = eng.last_timing()
                else:
                    res = self.run(1, 1, one)
                ts.append(time.perf_counter() - t0)
                if phases is not None:
                    phases.append(one)
            return ts, res

        def settle(self, seconds):
            """Blocking calls, discarded, for `seconds` of wall time: the chip needs ~0.1 s of sustained load before
            its kernel times stop moving (tools/ab_warm.sh); returns the number of calls."""
            # the number of calls is agreed between the ranks (every call ends in a collective when N > 1): from
            # the slowest rank's time for two calls
            ts, _ = self.blocking_calls(2)
            t = torch.tensor([max(min(ts), 1e-5)], dtype=torch.float64, device=dev)
            if use_dist:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            k = max(0, min(2000, int(seconds / float(t.item())) - 2))
            self.blocking_calls(k)
            return k + 2

        def timed(self, steps, depth):
            """barrier + synchronize on both sides, MAX over ranks; returns (seconds, last result)."""
            sync()
            t0 = time.perf_counter()
            res = self.run(steps, depth)
            sync()
            dt = time.perf_counter() - t0
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            if use_dist:
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            return float(tmax.item()), res

        def free(self):
            self.bases.free()
            self.scalars = None

    strong = args.total_log2n > 0
    n_total = (1 << args.total_log2n) if strong else (1 << args.log2n) * world
    wl = Workload(n_total, strong)
    n = wl.n
    # set-up (untimed, before the W warm-up steps): every lane allocates its workspaces; with an exchange,
    # enough steps for RCCL to finish its lazy channel set-up (the first collectives take milliseconds)
    wl.run(args.depth + (8 if use_dist else 0), args.depth)
    # Order of the legs: the blocking-call legs first, behind a settling period of their own, then the
    # contract's W warm-up steps and its K timed steps.  The chip's kernel times keep falling for its first ~0.1 s
    # of sustained load (tools/ab_warm.sh: the same 20 timed steps give 892-908 Mpts/s behind 5 warm-up steps and
    # 930-942 behind 50); a step here is ~1 ms, so W = 5 is 6 ms.  With the timed region first, it and the blocking
    # legs behind it both fell into that period at the driver's `--steps 20`.  Every rank runs the same sequence.
    # The same K steps as strictly blocking calls (not part of `value`), twice: first as a caller makes them
    # (`single_call`: wall time per call), then with the library's per-phase HIP events switched on
    # (bn256_ctx_set_phase_timing: the events cost 1-2.5 % of such a call, so they are off by default) for
    # `phases_ms` and the roofline's kernel time.  With several calls in flight an event pair would also span
    # the time a kernel waits behind the other call's kernels, which is not kernel time.  Medians over the calls.
    nblk = max(args.steps, 50)
    settle_calls = wl.settle(args.settle_ms * 1e-3)
    sync()
    per_call, result_blocking = wl.blocking_calls(nblk)
    sync()
    dt_blocking = sum(per_call)
    eng.set_phase_timing(True)
    wl.blocking_calls(2)
    per_phase = []
    wl.blocking_calls(nblk, per_phase)
    eng.set_phase_timing(False)
    sync()
    phase = [statistics.median(c[i] for c in per_phase) for i in range(7)]  # per call, ms
    wl.run(args.warmup, args.depth)
    host_us[:] = [0.0] * 3
    dt, result = wl.timed(args.steps, args.depth)
    steps = max(args.steps, 1)
    host_timed = [x / steps for x in host_us]

    # who took part (N > 1 or the forced 1-rank rehearsal): every rank's device and its k_accumulate time,
    # all-gathered, so that the line itself shows that `world` DISTINCT GPUs ran the job
    ranks_info = None
    if use_dist:
        prop = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device_name": prop.name,
                "pci_bus_id": "%04x:%02x:%02x" % (getattr(prop, "pci_domain_id", 0), getattr(prop, "pci_bus_id", 0),
                                                  getattr(prop, "pci_device_id", 0)),
                "uuid": str(getattr(prop, "uuid", "")), "points": wl.n,
                "k_accumulate_ms": phase[3], "single_call_median_ms": statistics.median(per_call) * 1e3,
                "hostname": os.uname().nodename, "pid": os.getpid()}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        acc_ms = [g["k_accumulate_ms"] for g in gathered]
        ids = {(g["hostname"], g["pci_bus_id"], g["uuid"]) for g in gathered}
        ranks_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                      "distinct_devices": len(ids), "all_distinct": len(ids) == world,
                      "k_accumulate_ms_min": min(acc_ms), "k_accumulate_ms_max": max(acc_ms), "per_rank": gathered}

    line = None
    if rank == 0:
        traffic = executed = pmc_file = None
        try:  # HBM bytes per k_accumulate launch from the committed rocprofv3 PMC passes (2^20, default tuning only)
            if n == (1 << 20) and not args.no_precompute and not args.window_bits:
                pmc_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_2p20.json")))[-1]
                with open(pmc_file) as f:
                    pmc = json.load(f)
                    pmc = pmc.get("k_accumulate<true>") or pmc["k_accumulate"]
                traffic = (pmc["FETCH_SIZE_KB"] + pmc["WRITE_SIZE_KB"]) * 1024.0
                # executed work next to the canonical 23 120 IMAD/point ("report executed
                # IMADs so waste is visible"): VALU lane-operations of all kinds per point
                executed = pmc["SQ_INSTS_VALU"] * 64.0 / n
        except Exception:
            traffic = executed = pmc_file = None
        acc_s = phase[3] * 1e-3
        # a fixed-base call above one pass (2^20 points with the 20-bit windows chosen from 2^20 points on,
        # MSM chunk in bn256msm.hip) runs as consecutive passes, one k_accumulate launch each; a blocking
        # call's phase times are summed over its passes, so kernel_ms covers all n points
        launches = passes_of(n, args)
        n_launch = -(-n // launches)
        achieved = n * IMAD_PER_POINT / acc_s if acc_s > 0 else 0.0
        med = statistics.median(per_call) if per_call else float("nan")
        line = {
            "metric": "MSM points/sec (BN256 G1, bit-exact with the reference's normalised affine output)",
            "value": n_total * args.steps / dt,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "int32x9 (29-bit digits, v_mad_i64_i32)",
            "data": "synthetic (seeded k_i*G0 bases, uniform scalars mod n)",
            "config": {
                "workload": (f"single BN256 G1 MSM, 2^{args.total_log2n} points in all, sliced over {world} GPU(s) "
                             f"(BASELINE.json configs[4] at T=24, 8 GPUs)" if strong else
                             f"single BN256 G1 MSM, 2^{args.log2n} points per GPU x {world} GPU(s) "
                             f"(BASELINE.json configs[2] per GPU at log2n=20)"),
                "points_total": n_total,
                "points_per_gpu": n,
                "rank0_slice": [wl.first, wl.n],
                "fixed_base_table": not args.no_precompute,
                "parallelism": f"point-slices x{world}, RCCL all_gather of 64-byte partials ({args.depth} steps' worth per call), host sum",
                "calls_in_flight": args.depth, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES") or 4),
                "order": f"set-up (table, one call per lane); {settle_calls} discarded blocking calls ({args.settle_ms} ms); "
                         f"{nblk} + {nblk} blocking calls (single_call, phases_ms / roofline); {args.warmup} warm-up steps; "
                         f"{args.steps} timed steps; the other legs",
                "setup_seconds_bases": wl.setup_s,
                "result_xy": result.hex(),
            },
            # per call (medians), HIP events on the call's stream, over the blocking calls before the warm-up steps
            "phases_ms": {"sort_pass_a": phase[0], "scan": phase[1], "sort_pass_b": phase[2], "accumulate": phase[3],
                          "fix": phase[4], "reduce": phase[5], "device_total": phase[6]},
            "host_us_per_step": {"submit": host_timed[0], "wait": host_timed[1], "exchange": host_timed[2]},
            "ranks": ranks_info,
            # SURVEY 8(d)'s wording of the metric: N / wall time of ONE blocking call (result download and,
            # at N > 1, exchange included; scalars resident), median of >= 10 after 2 warm-ups
            "single_call": {"calls": nblk, "median_ms": med * 1e3, "mean_ms": dt_blocking / nblk * 1e3,
                            "points_per_s": n_total / med if med == med else None,
                            "whole_msm_frac": (n * IMAD_PER_POINT / med) / IMAD_PEAK if med == med else None,
                            "whole_msm_frac_vs_r01_peak": (n * IMAD_PER_POINT / med) / IMAD_PEAK_R01 if med == med else None,
                            "same_result": bool(result_blocking == result)},
            "roofline": {
                "bound": "valu_int",
                "kernel": "k_accumulate",
                "kernel_ms": phase[3],
                "launches_per_call": launches,
                "points_per_launch": n_launch,
                "measured_over": f"median of {nblk} blocking calls before the warm-up steps, behind {settle_calls} discarded ones (one call on the chip at a time), "
                                 f"the library's HIP events on the call's own stream (switched on for these calls only)",
                "achieved": achieved,
                "peak": IMAD_PEAK,
                "unit": "IMAD/s",
                "frac": achieved / IMAD_PEAK,
                "frac_vs_r01_peak": achieved / IMAD_PEAK_R01,
                "peak_source": "profiles/r03_ubench_issue.txt (rounds 1-2: 3.47e13, profiles/r01_ubench_valu.txt)",
                "traffic": traffic,
                "traffic_source": os.path.relpath(pmc_file, ROOT) if pmc_file else None,
                "executed_valu_lane_ops_per_point": executed,
                "whole_msm_frac": (n * IMAD_PER_POINT / (dt / steps)) / IMAD_PEAK,
                "whole_msm_frac_vs_r01_peak": (n * IMAD_PER_POINT / (dt / steps)) / IMAD_PEAK_R01,
            },
        }
        if not args.no_precompute and n >= (1 << 18) and n <= (1 << 21) and not args.window_bits:
            # HBM rate of the sort / scatter phases (two-pass sort; 13 windows of 20 bits from 2^20 points on,
            # 16 of 16 bits below): ALGORITHMIC bytes -- 32-byte scalars in, one 4-byte record per (point,
            # window) out, then records in, sorted entries out -- over the live phase times (summed over the
            # passes of a call); the PMC counters are in profiles/ and DESIGN.md 6
            windows = 13 if n >= (1 << 20) else 16
            ent = windows * n * 4
            line["sort_hbm"] = {
                "windows": windows,
                "pass_a": {"bytes": 32 * n + ent, "ms": phase[0], "GB/s": (32 * n + ent) / (phase[0] * 1e-3) / 1e9},
                "pass_b": {"bytes": 2 * ent, "ms": phase[2], "GB/s": 2 * ent / (phase[2] * 1e-3) / 1e9},
                "peak_GB/s": 8000.0,
            }

    orc = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        v, orc = cpu_baseline(args.cpu_sample)
        # the oracle as checker: exact answer of the whole timed workload from its closed form
        # (bases are k_i*G0 with known k_i), bit-for-bit against the device result
        exact = orc.synth_msm_closed_form(SEED_BASES, SEED_SCALARS, n_total)
        line["cpu_baseline"] = {
            "value": v, "unit": "points/s", "cores": 1, "kind": "port",
            "device_result_matches_oracle_closed_form": bool(exact == result),
            "host_cpu": cpu_model(),
            "all_cores": cpu_baseline_all_cores(orc, 4 * args.cpu_sample),
            # the algorithm is O(N): seconds for the full sizes from the sampled rate (extrapolated, not measured)
            "extrapolated_seconds": {"2^20": (1 << 20) / v, "2^24": (1 << 24) / v},
            "sample": f"first {args.cpu_sample} terms of the same seeded workload, reference algorithm "
                      f"(per-term double-and-add + add, go/bp/util.go:165-175) restated in C, 1 thread",
        }

    if not args.no_extras and not strong and not args.no_precompute:
        if world == 1 and not use_dist:
            line.update(extras_one_gpu(eng, wl, orc, dev, torch))
            line.update(extras_prover(eng, wl, orc, dev, torch, result))
        # 2^24 points in all (configs[4]'s job): on this GPU at N = 1, sliced over the ranks at N > 1
        wl.free()
        big = Workload(1 << 24, True)
        big.run(args.depth, args.depth)
        ksteps = 12  # 0.2 s at N = 1 (six steps moved by 5 % from run to run)
        dtb, resb = big.timed(ksteps, args.depth)
        big.run(2, 1)
        sync()
        calls, _ = big.blocking_calls(5)
        sync()
        if rank == 0:
            entry = {"points_total": 1 << 24, "points_per_gpu": big.n, "scaling": "strong",
                     "passes_per_call": passes_of(big.n, args),
                     "steps": ksteps, "calls_in_flight": args.depth,
                     "ms_per_step": dtb / ksteps * 1e3, "points_per_s": (1 << 24) * ksteps / dtb,
                     "single_call_median_ms": statistics.median(calls) * 1e3,
                     "single_call_points_per_s": (1 << 24) / statistics.median(calls),
                     "whole_msm_frac": (big.n * IMAD_PER_POINT / (dtb / ksteps)) / IMAD_PEAK,
                     "setup_seconds_bases": big.setup_s, "result_xy": resb.hex()}
            if orc is not None:
                entry["matches_oracle_closed_form"] = bool(orc.synth_msm_closed_form(SEED_BASES, SEED_SCALARS, 1 << 24) == resb)
            line["n_2p24" if world == 1 else "strong_2p24"] = entry
        big.free()
        if world > 1 or use_dist:  # (use_dist at world 1: the one-rank rehearsal, BENCH_FORCE_DIST=1)
            # the same 2^24-point job on rank 0's GPU alone, in this very run (the other ranks' GPUs idle, their
            # processes parked on the rendezvous store): the N = 1 figure the scaling efficiency is taken against
            if rank == 0:
                t0 = time.perf_counter()
                b1 = eng.random_bases(SEED_BASES, 1 << 24, flags=flags)
                setup1 = time.perf_counter() - t0
                s1 = torch.empty((1 << 24) * 32, dtype=torch.uint8, device=dev)
                eng.random_scalars_into(SEED_SCALARS, 1 << 24, s1.data_ptr())
                sub1 = lambda: eng.msm_submit(b1, s1.data_ptr(), 1 << 24)  # noqa: E731
                sharding.run_pipelined(2, args.depth, sub1, eng.msm_wait)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res1 = sharding.run_pipelined(ksteps, args.depth, sub1, eng.msm_wait)
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t0
                e = line["n_2p24" if world == 1 else "strong_2p24"]
                e["n1_same_run"] = {"ms_per_step": dt1 / ksteps * 1e3, "points_per_s": (1 << 24) * ksteps / dt1,
                                    "setup_seconds_bases": setup1, "same_result": bool(res1 == resb)}
                e["speedup_vs_n1"] = (dt1 / ksteps) / (dtb / ksteps)
                e["efficiency_vs_n1"] = e["speedup_vs_n1"] / world
                b1.free()
                s1 = None
            try:
                from datetime import timedelta
                store = dist.distributed_c10d._get_default_store()
                if rank == 0:
                    store.set("bn256_n1_ref_done", "1")
                else:
                    store.wait(["bn256_n1_ref_done"], timedelta(seconds=400))
            except Exception:  # noqa: BLE001 -- private accessor: fall back to the collective
                pass
            dist.barrier()
        # the other form of multi-GPU use: ONE context over all the devices behind the C ABI, one process
        # (tools/multi_ctx_bench.py, a child of rank 0; the ranks wait at the barrier with their memory freed)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()
        if rank == 0:
            line["multi_device_ctx"] = multi_ctx_leg(world, resb.hex())
        if use_dist:
            # the other ranks wait on the rendezvous store, on the CPU: an RCCL barrier would park a spinning
            # kernel on every GPU the child is being timed on
            try:
                from datetime import timedelta
                store = dist.distributed_c10d._get_default_store()
                if rank == 0:
                    store.set("bn256_multi_ctx_done", "1")
                else:
                    store.wait(["bn256_multi_ctx_done"], timedelta(seconds=400))
            except Exception:  # noqa: BLE001 -- private accessor: fall back to the collective
                pass
            dist.barrier()

    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    if use_dist:
        dist.destroy_process_group()
    eng.close()
    return 0


def passes_of(n: int, args) -> int:
    """Passes (k_accumulate launches) of one fixed-base call of n points, as msm_enqueue_srcs plans them:
    20-bit windows from 2^20 points on, passes of at most 9/8 x 2^20 points there, 2^21 below."""
    if args.no_precompute:
        return 1
    wide = (n >= (1 << 20) and not args.window_bits) or args.window_bits == 20
    pass_max = (1 << 20) * 9 // 8 if wide else 1 << 21
    return -(-n // pass_max)


def multi_ctx_leg(world: int, expect_hex: str) -> dict:
    """bn256_ctx_create_multi over the node's devices, 2^24 points, in a fresh child process (a fault there
    must not cost this run its line).  One GPU: devices 0,0 -- two shards on one device, a rehearsal."""
    import subprocess
    devices = ",".join(str(i) for i in range(world)) if world > 1 else "0,0"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "multi_ctx_bench.py"), "--devices", devices,
           "--total-log2n", "24", "--expect", expect_hex]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        for ln in reversed(p.stdout.splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": f"exit code {p.returncode}", "stderr_tail": p.stderr[-300:]}
    except Exception as e:  # noqa: BLE001 -- reported in the line, never fatal for it
        return {"error": repr(e)[:300]}


def extras_one_gpu(eng, wl, orc, dev, torch) -> dict:
    """N = 1 legs that are not `value`: scalars in host memory, Bulletproofs-like scalars."""
    import numpy as np
    n = wl.n
    out = {}

    def med_of(fn, reps=10, warm=2):
        for _ in range(warm):
            r = fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), r

    want = eng.msm_dev(wl.bases, wl.scalars.data_ptr(), n)
    # wall time of bn256_msm with the scalar upload and the result download included
    host = wl.scalars.cpu()
    pageable = host.numpy()
    pinned = host.pin_memory()
    t_pg, r_pg = med_of(lambda: eng.msm_host_ptr(wl.bases, pageable.ctypes.data, n))
    t_pin, r_pin = med_of(lambda: eng.msm_host_ptr(wl.bases, pinned.data_ptr(), n))
    out["pcie_inclusive"] = {
        "note": "bn256_msm: 32 B/point of scalars cross PCIe inside the call; never `value`",
        "pageable": {"median_ms": t_pg * 1e3, "points_per_s": n / t_pg, "same_result": bool(r_pg == want)},
        "pinned": {"median_ms": t_pin * 1e3, "points_per_s": n / t_pin, "same_result": bool(r_pin == want)},
    }
    # Bulletproofs-like (go/bp/main.go:84-85): aL bits in {0,1} over the first half, aR = aL - 1 in
    # {0, n-1} over the second
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2, size=n // 2, dtype=np.uint8)
    arr = np.zeros((n, 32), dtype=np.uint8)
    arr[: n // 2, 31] = bits
    arr[n // 2:][bits == 0] = np.frombuffer((N_ORDER - 1).to_bytes(32, "big"), dtype=np.uint8)
    dbp = torch.from_numpy(arr.reshape(-1)).to(dev)
    torch.cuda.synchronize()
    t_bp, r_bp = med_of(lambda: eng.msm_dev(wl.bases, dbp.data_ptr(), n))
    out["bulletproofs_like"] = {"median_ms": t_bp * 1e3, "points_per_s": n / t_bp,
                                "distribution": "first half uniform in {0,1}, second half = first - 1 mod n (aL, aR of go/bp/main.go:84-85)"}
    if orc is not None:
        out["bulletproofs_like"]["matches_oracle_closed_form"] = bool(
            orc.synth_msm_closed_form_scalars(SEED_BASES, arr.tobytes()) == r_bp)
    return out


def load_pmc(kernel_prefix: str):
    """Per-launch counter means of a kernel from the newest committed element-wise PMC pass, or None."""
    try:
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_elementwise.json")))[-1]
        with open(f) as fh:
            d = json.load(fh)
        for k, v in d.items():
            if k.startswith(kernel_prefix):
                return v, os.path.relpath(f, ROOT)
    except Exception:  # noqa: BLE001
        pass
    return None, None


# Canonical work of ONE scalar multiplication for the element-wise rows (the MSM's 23 120 has no meaning there):
# the reference's own GLV double-and-add as SURVEY 3.5 states it -- 128 doublings (6M + 3S = 9 modmul) and 96
# mixed additions (8M + 2S = 10) -- at SURVEY 8(d)'s 136 IMAD per modmul: (128 * 9 + 96 * 10) * 136.
IMAD_PER_SCALAR_MUL = (128 * 9 + 96 * 10) * 136
# ... and of one fold output s_lo * P + s_hi * Q: both scalars split by the endomorphism, the four half-length
# scalars on ONE doubling chain (Straus / Shamir), joint sparse form per pair (half the digits non-zero):
# 128 doublings + 2 * 64 mixed additions.
IMAD_PER_FOLD = (128 * 9 + 128 * 10) * 136


def extras_prover(eng, wl, orc, dev, torch, result) -> dict:
    """N = 1: the table-less MSM, configs[0] (go/bp N = 64), configs[3] (BP++ commitments + WNLA at 2^20) and
    the element-wise rows.  Median of >= 5 after one warm run; every leg carries its exactness flag."""
    import crypto_amd
    from crypto_amd import bp as bpm
    n = wl.n
    out = {}

    def med_of(fn, reps=5, warm=1):
        for _ in range(warm):
            r = fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), r

    # ---- the reference's loop has no precomputation (go/bp/util.go:165-175), and every IPA / WNLA round after
    # the first runs over folded generators: one blocking bn256_msm_dev over a handle WITHOUT a table
    plain = eng.random_bases(SEED_BASES, n, first=wl.first)
    t, r = med_of(lambda: eng.msm_dev(plain, wl.scalars.data_ptr(), n), reps=7)
    out["no_table_2p20"] = {"median_ms": t * 1e3, "points_per_s": n / t, "calls": 7,
                            "whole_msm_frac": (n * IMAD_PER_POINT / t) / IMAD_PEAK,
                            "same_result_as_fixed_base": bool(r == result),
                            "note": "blocking bn256_msm_dev, 16 windows of 16 bits, 16 bucket sets, scalars resident"}
    # ---- element-wise rows on the same 2^n points: h1[i] = yinvn[i] * H[i] (go/bp/main.go:156-159) through
    # bn256_scalar_mul_batch(broadcast = 0) with the scalars in pinned host memory (the upload is part of the
    # call), and the generator fold g' = x^-1 G_lo + x G_hi (go/bp/main.go:355-356) through bn256_fold
    if orc is not None:
        host_sc = wl.scalars.cpu().pin_memory()
        sraw = host_sc.numpy().tobytes()
        kraw = orc.synth_base_scalars(SEED_BASES, n, wl.first)
        sc = lambda i: int.from_bytes(sraw[32 * i:32 * i + 32], "big")  # noqa: E731
        kk = lambda i: int.from_bytes(kraw[32 * i:32 * i + 32], "big")  # noqa: E731

        def h1():
            h = C.c_void_p()
            eng._chk(eng.lib.bn256_scalar_mul_batch(eng.ctx, plain.h, 0, n, C.cast(C.c_void_p(host_sc.data_ptr()), C.c_char_p),
                                                    0, C.byref(h)))
            return crypto_amd.engine.Bases(eng, h)

        def timed_handles(fn, reps=5):
            fn().free()
            ts, r = [], None
            for _ in range(reps):
                if r is not None:
                    r.free()
                t0 = time.perf_counter()
                r = fn()
                ts.append(time.perf_counter() - t0)
            return statistics.median(ts), r

        t, r = timed_handles(h1)
        idx = [0, 1, n // 3, n - 1]
        ok = all(r.download(i, 1)[0] == orc.g0_mul(sc(i) * kk(i) % N_ORDER) for i in idx)
        r.free()
        pmc, src = load_pmc("k_scalar_mul")
        out["h1_2p20"] = {"median_ms": t * 1e3, "points_per_s": n / t, "calls": 5, "sampled_outputs_match_closed_form": bool(ok),
                          "canonical_imad_per_point": IMAD_PER_SCALAR_MUL,
                          "imad_issue_frac": (n * IMAD_PER_SCALAR_MUL / t) / IMAD_PEAK,
                          "executed_valu_lane_ops_per_point": (pmc["SQ_INSTS_VALU"] * 64.0 / n) if pmc else None,
                          "pmc_source": src,
                          "note": "bn256_scalar_mul_batch(broadcast=0), 32 MiB of scalars uploaded from pinned memory inside the call"}
        s_lo, s_hi = (0x1234567 << 200) | 99, N_ORDER - 0xabcdef123
        half = n // 2
        t, r = timed_handles(lambda: eng.fold(plain, 0, plain, half, half, s_lo, s_hi))
        ok = all(r.download(i, 1)[0] == orc.g0_mul((s_lo * kk(i) + s_hi * kk(i + half)) % N_ORDER) for i in (0, 7, half // 2, half - 1))
        r.free()
        pmc, src = load_pmc("k_fold_points")
        out["fold_2p19"] = {"median_ms": t * 1e3, "outputs_per_s": half / t, "calls": 5, "sampled_outputs_match_closed_form": bool(ok),
                            "canonical_imad_per_output": IMAD_PER_FOLD,
                            "imad_issue_frac": (half * IMAD_PER_FOLD / t) / IMAD_PEAK,
                            "executed_valu_lane_ops_per_output": (pmc["SQ_INSTS_VALU"] * 64.0 / half) if pmc else None,
                            "pmc_source": src,
                            "note": "bn256_fold over the two halves of a 2^20-point vector: 2^19 outputs s_lo*lo_i + s_hi*hi_i"}
        host_sc = None
    plain.free()
    # ---- configs[0]: go/bp 64-bit range proof, prove + verify (go/bp/main_test.go:16-36) through the C++ mirror,
    # product entry points (OS randomness; generators resident)
    pub = bpm.BulletProofPublic(eng, 64)
    v, prv = 987654321, 0x1234567890ABCDEF
    pub.verify(pub.prove(v, prv))
    tp, tv, acc = [], [], True
    for _ in range(5):
        t0 = time.perf_counter()
        pr = pub.prove(v, prv)
        t1 = time.perf_counter()
        err = pub.verify(pr)
        t2 = time.perf_counter()
        tp.append(t1 - t0)
        tv.append(t2 - t1)
        acc = acc and err is None
    tampered = bytearray(pr)
    tampered[64 * 5 + 31] ^= 1  # tx
    out["bp_n64"] = {"prove_median_ms": statistics.median(tp) * 1e3, "verify_median_ms": statistics.median(tv) * 1e3,
                     "runs": 5, "verify_accepts": bool(acc), "tampered_rejected": pub.verify(bytes(tampered)) is not None,
                     "proof_bytes": len(pr), "config": "BASELINE.json configs[0]: go/bp range proof N = 64 (bp_prove / bp_verify)"}
    pub.G.free()
    pub.H.free()
    # ---- configs[3]: the four BP++ vector commitments Co, Cl, Cr, Cs (go/bppp/docs_test.go:560-564,588-589,
    # 1002-1003) at Nv = Nm = 2^20, each <HVec, r||l> + <GVec, n> as one pipeline over two fixed-base tables
    nv = nm = 1 << 20
    SEED_H, SEED_G = 0xB9901, 0xB9902
    t0 = time.perf_counter()
    H = eng.random_bases(SEED_H, 9 + nv, flags=crypto_amd.PRECOMPUTE)
    G = eng.random_bases(SEED_G, nm, flags=crypto_amd.PRECOMPUTE)
    setup_ms = (time.perf_counter() - t0) * 1e3
    seeds = [(0x21, 0x22), (0x23, 0x24), (0x25, 0x26), (0x27, 0x28)]
    vec = [(eng.fr_random(a, 9 + nv), eng.fr_random(b, nm)) for a, b in seeds]
    segs = lambda i: [(H, 0, 1, 9 + nv, vec[i][0], 0, 1), (G, 0, 1, nm, vec[i][1], 0, 1)]  # noqa: E731

    def blocking():
        return [eng.msm_fr_multi(segs(i)) for i in range(4)]

    def on_lanes():
        tk = [eng.msm_fr_multi_submit(segs(i)) for i in range(4)]
        return [eng.msm_wait(x) for x in tk]

    tb, rb = med_of(blocking)
    tl, rl = med_of(on_lanes)
    terms = 9 + nv + nm
    entry = {"terms_per_commitment": terms, "setup_ms_both_tables": setup_ms,
             "blocking": {"four_commitments_ms": tb * 1e3, "ms_per_commitment": tb / 4 * 1e3, "points_per_s": 4 * terms / tb},
             "on_lanes": {"four_commitments_ms": tl * 1e3, "ms_per_commitment": tl / 4 * 1e3, "points_per_s": 4 * terms / tl},
             "same_result": bool(rb == rl)}
    if orc is not None:
        want = [orc.g0_mul((orc.synth_dot(SEED_H, a, 9 + nv) + orc.synth_dot(SEED_G, b, nm)) % N_ORDER) for a, b in seeds]
        entry["matches_oracle_closed_form"] = bool(rb == want)
    for a, b in vec:
        a.free()
        b.free()
    H.free()
    G.free()
    # the weighted-norm-linear argument over |l| = 2^20, |n| = 2^19 (go/bppp/docs_test.go:1269-1338): all rounds'
    # X / R commitments and generator folds on the device, final check l, n opened against C
    w = bpm.WNLA(eng)
    w.synthetic(0xABCC, 1 << 20, 1 << 19)
    runs = [w.synthetic(0xABCD + i, 1 << 20, 1 << 19) for i in range(5)]
    entry["wnla_2p20"] = {"median_ms": statistics.median(r["ms_run"] for r in runs), "rounds": runs[0]["rounds"],
                          "final_check_passes": all(r["ok"] for r in runs), "runs": 5,
                          "setup_ms": statistics.median(r["ms_setup"] for r in runs)}
    out["bppp_2p20"] = entry
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launcher(args))
    sys.exit(worker(args))


if __name__ == "__main__":
    main()
