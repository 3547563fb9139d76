"""The N > 1 path with the HIP library producing the partial sums (the sum being sharded
is vectorPointScalarMul, go/bp/util.go:165-175): ranks started by crypto_amd.launch, each runs
bn256_msm_submit / bn256_msm_wait on its split_even slice, sharding.AsyncExchange + combine must
give every rank the single-GPU result.  On a one-GPU box the ranks share cuda:0 and exchange over
gloo; with >= 2 GPUs visible the same script runs one rank per GPU over RCCL, and bench.py's bare
`--gpus 2` command is exercised end to end."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "_rank_gpu.py")


def _launch(argv, nproc):
    """crypto_amd.launch from a child process, so that this pytest process stays the only holder of its
    own GPU state and the helper's stdout is captured."""
    code = ("import sys; sys.path.insert(0, %r); from crypto_amd import launch; "
            "sys.exit(launch.launch_ranks(%r, %r, %d, timeout_s=240))" % (ROOT, HELPER, list(argv), nproc))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def _check(out, eng, orc, n_total, world):
    h = eng.random_bases(0x5EED0001, n_total)
    v = eng.fr_random(0x5EED0002, n_total)
    try:
        want = eng.msm_dev(h, v.device_ptr(), n_total)
        assert want == orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, n_total)
        assert out["world"] == world and len(out["results_per_rank"]) == world
        for per_rank in out["results_per_rank"]:       # pipelined, blocking, one-shot all_gather: all the same point
            assert per_rank == [want.hex()] * 3
        # every partial is the MSM over exactly that rank's slice
        for (off, n), part in zip(out["slices"], out["partials"]):
            assert bytes.fromhex(part) == eng.msm_dev(h, v.device_ptr() + 32 * off, n, off=off)
    finally:
        h.free()
        v.free()


def test_two_ranks_hip_partials_exchanged_over_gloo(eng, orc):
    n_total = (1 << 17) + 1  # ragged slices: 65537 + 65536
    out = _launch(["--backend", "gloo", "--same-device", "--n-total", str(n_total)], 2)
    assert out["slices"] == [[0, 65537], [65537, 65536]]
    _check(out, eng, orc, n_total, 2)


def test_three_ranks_tableless_small(eng, orc):
    n_total = 1000
    out = _launch(["--backend", "gloo", "--same-device", "--n-total", str(n_total), "--precompute", "0", "--depth", "2"], 3)
    _check(out, eng, orc, n_total, 3)


def _ngpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif("_ngpus() < 2", reason="needs two GPUs (one rank per GPU over RCCL)")
def test_two_ranks_rccl_one_gpu_each(eng, orc):
    n_total = (1 << 20) + 3
    out = _launch(["--backend", "nccl", "--n-total", str(n_total)], 2)
    _check(out, eng, orc, n_total, 2)


@pytest.mark.skipif("_ngpus() < 2", reason="needs two GPUs")
def test_bench_bare_command_two_gpus(orc):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--total-log2n", "21", "--no-cpu-baseline"], env=env, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["points_per_gpu"] == 1 << 20
    assert bytes.fromhex(out["config"]["result_xy"]) == orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, 1 << 21)


def test_bench_rank_path_with_a_one_rank_rccl_group(orc):
    """bench.py's N > 1 code on one GPU: BENCH_FORCE_DIST=1 makes the single rank open an RCCL process group and
    run every step's partial sum through sharding.AsyncExchange (side stream, batched all_gather, finish_batch)
    and the barriers -- the code the driver's 2/4/8-GPU runs execute, which a one-GPU box otherwise never
    touches.  The line must carry the exact sum (closed form), a non-zero exchange time and the `ranks` block
    that an N > 1 line proves itself with (world size, devices seen by RCCL, per-rank kernel time)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577"})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=280)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    n = 1 << 20
    assert out["n_gpus"] == 1 and out["config"]["points_total"] == n
    assert bytes.fromhex(out["config"]["result_xy"]) == orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, n)
    assert out["single_call"]["same_result"] is True
    assert out["host_us_per_step"]["exchange"] > 0
    rk = out["ranks"]
    assert rk["world_size"] == 1 and rk["backend"] == "nccl" and rk["distinct_devices"] == 1 and rk["all_distinct"]
    assert len(rk["per_rank"]) == 1 and rk["per_rank"][0]["points"] == n
    assert 0 < rk["k_accumulate_ms_min"] <= rk["k_accumulate_ms_max"] < 5
    assert rk["per_rank"][0]["pci_bus_id"] and rk["per_rank"][0]["device_name"]
    # the legs only an N > 1 run has: 2^24 points over the ranks, the same job on rank 0's GPU alone in the same run
    # (the figure `efficiency_vs_n1` is taken against), the one-context-over-all-devices form
    big = out["n_2p24"]
    assert bytes.fromhex(big["result_xy"]) == orc.synth_msm_closed_form(0x5EED0001, 0x5EED0002, 1 << 24)
    assert big["n1_same_run"]["same_result"] is True and 0.7 < big["efficiency_vs_n1"] < 1.4
    assert out["multi_device_ctx"]["same_result_as_rank_path"] is True
