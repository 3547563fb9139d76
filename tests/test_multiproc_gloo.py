"""world_size-2 CPU test (gloo) of the N>1 path: slice assignment, all_gather of the 64-byte
partial points and the host-side combination (bn256_g1_sum).  The per-rank partial MSM is
produced by the oracle here (no GPU in this tier); on GPUs bench.py runs the same
sharding.all_gather_points / sharding.combine around bn256_msm_dev."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from crypto_amd import sharding
    from oracle_c import OracleC
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = OracleC()
        off, n = sharding.split_even(n_total, world, rank)
        bases = orc.synth_bases(0x5EED0001, n, first=off)
        scal = orc.synth_scalars(0x5EED0002, n, first=off)
        part = orc.msm(bases, scal)
        parts = sharding.all_gather_points(part)
        total = sharding.combine(parts)
        q.put((rank, off, n, total.hex(), [p.hex() for p in parts]))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_msm_matches_single():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_c import OracleC
    from crypto_amd import sharding
    n_total, world = 301, 2                      # odd on purpose: ragged slices
    assert [sharding.split_even(n_total, world, r) for r in range(world)] == [(0, 151), (151, 150)]
    assert sharding.split_even(5, 8, 7) == (5, 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    orc = OracleC()
    want = orc.msm(orc.synth_bases(0x5EED0001, n_total), orc.synth_scalars(0x5EED0002, n_total))
    assert res[0][3] == res[1][3] == want.hex()
    assert res[0][4] == res[1][4] and len(res[0][4]) == world
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 151, 151, 150)
